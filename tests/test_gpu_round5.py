"""Round-5 GPU tests:
  * config 5's start-up: the three strategies (broadcast / every engine builds its own / 1-of-N slices + all-gather) give byte-identical tables and the same
    hit lists -- in the library (two engines on one GPU, peer copies), through the C++ host (`-d 0,0`) and through `bench.py --gpus 2 --same-device`;
  * the fabric on its own: RCCL loaded, initialised and called next to the engine (a one-rank communicator is what a one-GPU lease can hold), peer copies
    between two engines, RCCL refusing one GPU listed twice;
  * extended tables with a bucket count that is not a power of two at tile level."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def _centres(w, maxnonce):
    from pybsgs import ecpy
    ms = [5 * 2 * w + 77, -(maxnonce * 2 * w) + 12345, w // 3, 17 * 2 * w - w, (maxnonce + 5) * 2 * w + 99, 0x5EED5EED5EED5EED5EED]
    return ms, [ecpy.mul(m % N) for m in ms]


@pytest.mark.parametrize("w,htsz,layout", [(1 << 20, 14, 4), (1 << 20, 13, 5), ((1 << 20) + 777, 12288, 5), (3 << 19, 15, 4), (1 << 20, 98306, 4)])
def test_three_startup_strategies_give_byte_identical_tables_on_two_engines(w, htsz, layout):
    """bsgs_startup_ext_tables with two engines of one process (both on GPU 0: peer copies; RCCL refuses one GPU listed twice): BROADCAST, LOCAL and
    ALLGATHER must leave BOTH engines with the same table -- equal position-dependent checksums across engines AND across strategies (the direct builder
    closes its lines sorted: a table is a function of (w, buckets) alone) --, a clean census, and the hit list of a reference build."""
    import pybsgs
    from pybsgs import ecpy
    from test_gpu_fullsize import analytic_hits
    t, b, p = 64, 8, 32
    maxnonce = t * b * p
    ms, centres = _centres(w, maxnonce)
    A = ecpy.addpubg(w)
    ref = pybsgs.Device(0)
    ref.generate_g2(A[0], A[1], t, b, p)
    ref.build_baby_table_ext(w, htsz, layout)
    want, nw, _ = ref.run(centres, 65536)
    want_sums = ref.table_checksum()
    assert ref.table_census()["total"] == w
    for k, m in enumerate(ms):
        assert set(analytic_hits(m, w, maxnonce)) <= {(c, i) for tile, c, i in want if tile == k}
    ref.close()
    seen = {}
    for strategy in (pybsgs.STARTUP_BROADCAST, pybsgs.STARTUP_LOCAL, pybsgs.STARTUP_ALLGATHER):
        devs = [pybsgs.Device(0), pybsgs.Device(0)]
        for d in devs:
            d.generate_g2(A[0], A[1], t, b, p)
        rep = pybsgs.startup_ext_tables(devs, w, htsz, layout, strategy, pybsgs.TRANSPORT_PEER)
        assert [r["strategy"] for r in rep] == [strategy] * 2 and all(r["total_s"] > 0 for r in rep)
        if strategy == pybsgs.STARTUP_LOCAL:
            assert all(r["bytes_received"] == 0 and r["build_s"] > 0 for r in rep)
        else:
            assert rep[1]["bytes_received"] > 0 and rep[1]["transport"] == pybsgs.TRANSPORT_PEER
        sums = [d.table_checksum() for d in devs]
        assert sums[0] == sums[1] == want_sums, (strategy, sums, want_sums)
        for d in devs:
            c = d.table_census()
            assert c["total"] == w and c["malformed_lines"] == 0 and c["unsorted_lines"] == 0, (strategy, c)
            got, ng, _ = d.run(centres, 65536)
            assert (ng, got) == (nw, want), strategy
            if layout == 4 and htsz > 31:                               # 64-byte lines, a bucket count that is no power of two: the kernels of their own
                assert d.last_kernel() == "giant_pair2_kernel<4, false, true>"
            assert d.table_owned()
        seen[strategy] = sums[0]
        for d in devs:
            d.close()
    assert len({tuple(v) for v in seen.values()}) == 1


@pytest.mark.parametrize("layout", ["image", 4, 5])
def test_twin_engine_shares_the_owners_table_in_place(layout):
    """bsgs_share_tables (the host's lanes: two public keys searched side by side on one GPU): the twin probes the owner's buffers -- same checksums, same hits, nothing of
    the table owned by it --, keeps working next to the owner, and its going leaves the owner's table untouched.  Engines on different GPUs are refused (same GPU here:
    the refusal of an engine as its own twin is what a one-GPU lease can show)."""
    import pybsgs
    from pybsgs import ecpy
    w, t, b, p = 1 << 18, 64, 8, 16
    ms, centres = _centres(w, t * b * p)
    A = ecpy.addpubg(w)
    own, twin = pybsgs.Device(0), pybsgs.Device(0)
    own.generate_g2(A[0], A[1], t, b, p)
    if layout == "image":
        own.build_baby_tables(w, 14, install_layout=pybsgs.TABLE_LINES64)
    else:
        own.build_baby_table_ext(w, 13, layout)
    want, nw, _ = own.run(centres, 65536)
    free_before = own.meminfo()[0]
    pybsgs.share_tables(own, twin)
    assert own.meminfo()[0] > free_before - (64 << 20)             # the giants (512 KiB here) and nothing like a table
    assert twin.table_checksum() == own.table_checksum() and twin.table_info() == own.table_info()
    assert not twin.table_owned() and own.table_owned()
    for d in (twin, own, twin):
        got, ng, _ = d.run(centres, 65536)
        assert (ng, got) == (nw, want)
    with pytest.raises(pybsgs.BsgsError, match="two different engines"):
        pybsgs.share_tables(own, own)
    sums = own.table_checksum()
    twin.close()
    assert own.table_checksum() == sums
    got, ng, _ = own.run(centres, 65536)
    assert (ng, got) == (nw, want)
    own.close()


def test_allgather_falls_back_when_the_buckets_do_not_divide():
    import pybsgs
    devs = [pybsgs.Device(0) for _ in range(3)]
    rep = pybsgs.startup_ext_tables(devs, 1 << 18, 12, pybsgs.TABLE_LINES64_LIST, pybsgs.STARTUP_ALLGATHER, pybsgs.TRANSPORT_PEER)      # 4096 buckets, 3 engines
    assert [r["strategy"] for r in rep] == [pybsgs.STARTUP_BROADCAST] * 3
    sums = [d.table_checksum() for d in devs]
    assert sums[0] == sums[1] == sums[2]
    for d in devs:
        d.close()


def test_fabric_rccl_one_rank_peer_two_engines_and_rccl_refuses_a_gpu_listed_twice():
    """the transports on their own (bsgs_debug_fabric_selftest): a broadcast and an in-place all-gather over buffers from the bucket lines' allocator, checked
    on the device.  RCCL with ONE engine = librccl dlopen'ed, ncclCommInitAll, ncclBroadcast, ncclAllGather, ncclCommDestroy next to the engine in one process
    (what a one-GPU lease can hold of north_star's "RCCL over xGMI"); peer copies between two engines on GPU 0; RCCL over `0,0` must refuse."""
    import pybsgs
    d0, d1 = pybsgs.Device(0), pybsgs.Device(0)
    assert pybsgs.fabric_selftest([d0], pybsgs.TRANSPORT_RCCL, 256 << 20) == (0, 0, pybsgs.TRANSPORT_RCCL)
    assert pybsgs.fabric_selftest([d0, d1], pybsgs.TRANSPORT_PEER, 256 << 20) == (0, 0, pybsgs.TRANSPORT_PEER)
    assert pybsgs.fabric_selftest([d0, d1], pybsgs.TRANSPORT_AUTO, 64 << 20) == (0, 0, pybsgs.TRANSPORT_PEER)      # one GPU listed twice: auto = peer copies
    with pytest.raises(pybsgs.BsgsError, match="distinct GPUs"):
        pybsgs.fabric_selftest([d0, d1], pybsgs.TRANSPORT_RCCL, 64 << 20)
    # buffers above 40 GiB are composed of mapped 4 GiB chunks (placement.hip): RCCL and peer copies must work on those too
    import torch
    if torch.cuda.mem_get_info(0)[0] > 150 * 2**30:
        assert pybsgs.fabric_selftest([d0], pybsgs.TRANSPORT_RCCL, 44 << 30) == (0, 0, pybsgs.TRANSPORT_RCCL)
        assert pybsgs.fabric_selftest([d0, d1], pybsgs.TRANSPORT_PEER, 44 << 30) == (0, 0, pybsgs.TRANSPORT_PEER)
    d0.close(); d1.close()


def test_broadcast_tables_ex_reports_its_transport_and_replicates_giants_and_table_apart():
    import pybsgs
    from pybsgs import ecpy
    w, htsz, t, b, p = 1 << 18, 14, 64, 8, 16
    A = ecpy.addpubg(w)
    d0, d1 = pybsgs.Device(0), pybsgs.Device(0)
    d0.generate_g2(A[0], A[1], t, b, p)
    d0.build_baby_tables(w, htsz, install_layout=pybsgs.TABLE_LINES64)
    used, secs = pybsgs.broadcast_tables_ex([d0, d1], pybsgs.TRANSPORT_AUTO, 1)
    assert used == pybsgs.TRANSPORT_PEER and secs > 0 and d1.table_checksum()[3] == d0.table_checksum()[3] and d1.table_checksum()[0] == 0
    used, _ = pybsgs.broadcast_tables_ex([d0, d1], pybsgs.TRANSPORT_PEER, 2)
    assert d1.table_checksum() == d0.table_checksum() and d1.table_info() == d0.table_info()
    d0.close(); d1.close()


def _host(args, tmp_path, env=None, timeout=900):
    return subprocess.run([HOST, "-dir", str(tmp_path)] + args, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))


def test_host_startup_strategies_two_engines(tmp_path):
    """bsgs_mi355x -d 0,0 with an extended table: `-startup broadcast | local | allgather` find the key, print every engine's start-up stages, and the replica
    verification reports the SAME checksums for all three; `-transport rccl` with one GPU listed twice is refused; file tables: broadcast (default) and local."""
    from pybsgs import ecpy
    key = 0xABCDE
    x, y = ecpy.mul(key)
    geo = ["-t", "64", "-b", "8", "-p", "16", "-pb", "%02x%064x" % (2 + (y & 1), x), "-pk", "1", "-pke", "ffffff", "-d", "0,0"]
    sums = {}
    for st in ("broadcast", "local", "allgather"):
        r = _host(geo + ["-ext", "-w", "16", "-htsz", "11", "-startup", st], tmp_path)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert "KEY[1]: 0x" + "%064x" % key in r.stdout and r.stdout.count("strategy %s" % st) == 2, r.stdout[-3000:]
        assert "tables on every engine (%s)" % st in r.stdout
        m = re.search(r"Replica verification: 2 engines hold identical tables \((.*?)\), probe tile", r.stdout)
        assert m, r.stdout[-2000:]
        sums[st] = m.group(1)
        if st != "local":
            assert "peer copies" in r.stdout
    assert len(set(sums.values())) == 1, sums
    bad = _host(geo + ["-ext", "-w", "16", "-htsz", "11", "-startup", "broadcast", "-transport", "rccl"], tmp_path)
    assert bad.returncode != 0 and "distinct GPUs" in bad.stderr
    # file tables (the reference's formats): device-to-device replicas (default) and the reference's own way, one upload per engine
    for st, words in (("broadcast", "Tables replicated to 1 more GPU engine(s) by peer copies"), ("local", "Tables uploaded to every GPU engine from the host")):
        r = _host(geo + ["-w", "16", "-htsz", "12", "-startup", st], tmp_path)
        assert r.returncode == 0 and words in r.stdout and "KEY[1]: 0x" + "%064x" % key in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
        assert "Replica verification: 2 engines hold identical tables" in r.stdout
    # one engine under a forced RCCL transport: nothing to replicate, nothing loaded, the search runs
    one = _host(geo[:-2] + ["-d", "0", "-w", "16", "-htsz", "12", "-transport", "rccl"], tmp_path)
    assert one.returncode == 0 and "KEY[1]: 0x" + "%064x" % key in one.stdout


def test_host_extended_table_with_any_number_of_buckets(tmp_path):
    """`-buckets N` / a fractional `-htsz`: an extended table whose bucket count is not a power of two (128-byte lines, the bucket from 48 bits of the key)
    through the host: key found at the far end of the range; a power of two given as a count is the plain -htsz"""
    from pybsgs import ecpy
    key = 0xF0F0F1
    x, y = ecpy.mul(key)
    geo = ["-t", "64", "-b", "8", "-p", "16", "-pb", "%02x%064x" % (2 + (y & 1), x), "-pk", "1", "-pke", "ffffff", "-w", "18"]
    r = _host(geo + ["-buckets", "12289"], tmp_path)
    assert r.returncode == 0 and "KEY[1]: 0x" + "%064x" % key in r.stdout and "12289 buckets (not a power of two)" in r.stdout and "12289 lines of 128 bytes" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    r = _host(geo + ["-htsz", "13.585"], tmp_path)
    assert r.returncode == 0 and "KEY[1]: 0x" + "%064x" % key in r.stdout and "buckets (extended table)" in r.stdout
    r = _host(geo + ["-buckets", "24577"], tmp_path)                                  # load 10.67: 64-byte lines (the host's rule: up to 12.5 per bucket)
    assert r.returncode == 0 and "KEY[1]: 0x" + "%064x" % key in r.stdout and "24577 lines of 64 bytes" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    r = _host(geo + ["-buckets", "8192", "-sf", "1"], tmp_path)
    assert r.returncode == 0 and "KEY[1]: 0x" + "%064x" % key in r.stdout and "Search in file" in r.stdout and "not a power of two" not in r.stdout


def _bench(args, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_three_startup_strategies(tmp_path):
    """`bench.py --gpus 2 --same-device --force-ext --startup-strategy X`: the N-rank line names its strategy, carries every rank's start-up stages, all ranks
    verify equal (checksums + one launch everybody runs), and the three strategies end with the same table and the same hits"""
    common = ["--w", "24", "--htsz", "21", "--tiles-per-launch", "24", "--steps", "2", "--warmup", "1", "--warmup-s", "0", "--sustain-s", "0", "--no-cpu-baseline", "--no-solve",
              "--no-pmc", "--no-refquirks-leg", "--gpus", "2", "--same-device", "--force-ext"]
    res = {}
    for st in ("broadcast", "local", "allgather"):
        dump = str(tmp_path / (st + ".json"))
        d = _bench(common + ["--startup-strategy", st, "--dump-hits", dump])
        assert d["startup_strategy"] == st and d["n_gpus"] == 2 and d["table_checksum_equal"] and d["replica_hits_equal"]
        assert all(r["startup_stages"] and r["startup_stages"]["install_s"] > 0 for r in d["per_rank"])
        if st == "local":
            assert d["table_broadcast_GB"] == 0 and all(r["table_build"] for r in d["per_rank"])
        else:
            assert d["table_broadcast_GB"] > 0.05
        with open(dump) as f:
            res[st] = (d["per_rank"][0]["table_checksums"], json.load(f)["hits"])
    assert res["broadcast"] == res["local"] == res["allgather"]
    assert len(res["local"][1]) >= 1


def test_lanes_two_jobs_side_by_side_keep_list_order_and_recovery(tmp_path):
    """BASELINE config 4's shape in small: 12 public keys over a fixed range whose jobs are a launch or two long.  `-lanes 2` (the automatic choice for such jobs) searches two
    keys side by side, each on an engine of its own on the one GPU; win.txt, the console and "Found n of m" must be what `-lanes 1` gives, in list order -- including a key
    outside the range ("Reached end of space") in the middle of the list; and -wl resumes at a list position with both lanes."""
    import hashlib
    from pybsgs import ecpy
    t, b, p, wexp, htsz = 64, 8, 16, 16, 13
    lo, hi = 1 << 40, (1 << 41) - 1
    keys, st = [], 0xBEEF
    for i in range(12):
        st, r = ecpy.splitmix64(st)
        keys.append(lo + r % (hi - lo))
    keys[5] = hi + 12345678901                                    # not in the range: its job ends with "Reached end of space"
    infile = tmp_path / "pubs.txt"
    infile.write_text("\n".join("%064x%064x" % ecpy.mul(k) for k in keys) + "\n")
    geo = ["-t", str(t), "-b", str(b), "-p", str(p), "-w", str(wexp), "-htsz", str(htsz), "-infile", str(infile), "-pk", "%x" % lo, "-pke", "%x" % hi]
    res = {}
    for lanes in ("1", "2"):
        d = tmp_path / ("l" + lanes)
        d.mkdir()
        r = _host(geo + ["-lanes", lanes], d)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        win = (d / "win.txt").read_bytes().decode().split("\r\n")
        res[lanes] = ([l for l in win if l.startswith("KEY[")], [l for l in r.stdout.splitlines() if l.startswith(("Findpubkey", "KEY[", "Reached end"))], r.stdout)
        assert "Found 11 of 12" in r.stdout
    assert res["1"][0] == res["2"][0] == ["KEY[%d]: 0x%064x" % (i + 1, k) for i, k in enumerate(keys) if i != 5]
    assert res["1"][1] == res["2"][1]                              # same lines, same order on the console
    assert res["2"][2].count("memory") == 2 and res["1"][2].count("memory") == 1
    # recovery at list position 7 with two lanes: positions 7..12 are searched (position 7 from the saved counter), nothing before
    gstep = 4 * t * b * p * (1 << wexp)
    cnt = 1 + max(0, (keys[6] - lo) // gstep - 1) * gstep
    fp = hashlib.sha1(("%d%d%d%d%s%s%d" % (t, b, p, 1 << wexp, "%x" % lo, "%x" % hi, htsz)).encode()).hexdigest()
    d = tmp_path / "rec"
    d.mkdir()
    (d / "currentwork.txt").write_bytes(("7\r\n%064x%064x\r\n%064x\r\n%s\r\n" % (ecpy.mul(keys[6]) + (cnt, fp))).encode())
    r = _host(geo + ["-lanes", "2", "-wl", str(d / "currentwork.txt")], d)
    assert r.returncode == 0 and "Recovery: listpos 7" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    win = [l for l in (d / "win.txt").read_bytes().decode().split("\r\n") if l.startswith("KEY[")]
    assert win == ["KEY[%d]: 0x%064x" % (i + 1, k) for i, k in enumerate(keys) if i >= 6]


def test_overflow_fingerprint_in_the_headers_of_over_full_lines():
    """Round 5: the header of an over-full line is 0x80000000 | fingerprint -- bits min((hash >> 16) & 31, 30) and min((hash >> 21) & 31, 30) set for every hash of the bucket
    that lives only in the overflow set -- and the probe asks the set only for a hash whose bits are set (both of them in the kernels of tables with any number of buckets and
    in the 128-byte-line kernels, the first one in the kernel of 2^htsz-bucket tables: giant_kernel.hip.h).  (i) the headers the builder writes are exactly that,
    recomputed here from the set; (ii) the same table with plain 0xFFFFFFFF headers (no fingerprint: a table built elsewhere) is accepted and gives the same
    hit lists; (iii) a header that lacks the bit of one of its set-only hashes is refused at install: the probe would never find that key."""
    import numpy as np
    import torch
    import pybsgs
    from pybsgs import ecpy
    t, b, p = 64, 2, 16
    for w, htsz, lay, words in ((1 << 16, 12, pybsgs.TABLE_LINES64_LIST, 16), (1 << 17, 12, pybsgs.TABLE_LINES128_LIST, 32)):     # load 16 / 32: half the lines over-full, by a few entries
        dev = pybsgs.Device(0)
        items = 1 << htsz
        cap = dev.ext_overflow_capacity(w, htsz, lay)
        lines = torch.empty(items * words, dtype=torch.int32, device="cuda:0")
        ovf = torch.empty(cap, dtype=torch.int64, device="cuda:0")
        n_ovf, n_over = dev.build_baby_table_ext_device(w, htsz, lay, lines.data_ptr(), ovf.data_ptr(), cap)
        dev.install_table_ext_device(lines.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)
        torch.cuda.synchronize()
        L = lines.cpu().numpy().view(np.uint32).reshape(items, words)
        S = ovf[:n_ovf].cpu().numpy().view(np.uint64)
        S = S[S != np.uint64(0xFFFFFFFFFFFFFFFF)]
        hdr, bound = L[:, 0], L[:, words - 1]
        over = hdr >= 0x80000000
        assert int(over.sum()) == n_over and 0.2 * items < n_over < 0.9 * items, n_over
        assert np.all(hdr[~over] <= words - 1)
        sb, sh = (S >> np.uint64(32)).astype(np.int64), (S & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        only = over[sb] & (sh != bound[sb])                      # keys the set alone holds (a line's last word is also in the set: the bound)
        fp = np.zeros(items, dtype=np.uint32)
        np.bitwise_or.at(fp, sb[only], (np.uint32(1) << np.minimum((sh[only] >> np.uint32(16)) & np.uint32(31), np.uint32(30))).astype(np.uint32))
        np.bitwise_or.at(fp, sb[only], (np.uint32(1) << np.minimum((sh[only] >> np.uint32(21)) & np.uint32(31), np.uint32(30))).astype(np.uint32))
        assert np.array_equal(hdr[over], (np.uint32(0x80000000) | fp[over]))
        sparse = float(np.mean([bin(int(v) & 0x7FFFFFFF).count("1") for v in hdr[over]]))
        assert 1.0 <= sparse <= 14.0, sparse                     # a few set-only hashes per over-full line, two bits each: a few bits of 31 -- that is what makes the filter bite
        # (ii) hit lists with the fingerprint == hit lists of the same table without one
        A = ecpy.addpubg(w)
        dev.generate_g2(A[0], A[1], t, b, p)
        ms, cs = _centres(w, t * b * p)
        # table keys that ONLY the set holds, reached as P - G2[i] / P + G2[i] from crafted centres: k = m -+ (i + 1) * 2w
        xs_only = {}
        for k in range(1, 4000):
            x = ecpy.mul(k)[0]
            bk, h = x & (items - 1), (x >> 32) & 0xFFFFFFFF
            if over[bk] and h > bound[bk]:
                xs_only[k] = (bk, h)
        assert len(xs_only) > 50
        ks = sorted(xs_only)[:6]
        cs = cs + [ecpy.mul((k + (i + 3) * 2 * w) % N) for i, k in enumerate(ks)]        # P + G2[i + 2] = P - (i + 3) * 2w * G = k*G: code 1
        with_fp = [dev.step(c[0], c[1], 65536) for c in cs]
        for i, k in enumerate(ks):
            assert (1, i + 2) in [tuple(h) for h in with_fp[len(cs) - len(ks) + i][0]], (k, i)
        lines2 = lines.clone()
        lines2.view(items, words)[torch.from_numpy(over).to("cuda:0"), 0] = -1          # 0xFFFFFFFF: every bit set = always ask the set
        dev.install_table_ext_device(lines2.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)
        assert [dev.step(c[0], c[1], 65536) for c in cs] == with_fp
        # (iii) one missing bit: the first of a set-only hash's two, then the second
        bk, h = xs_only[ks[0]]
        for shift in (16, 21):
            bad = lines.clone()
            bad[bk * words] = int(np.array([int(hdr[bk]) & ~(1 << min((h >> shift) & 31, 30))], dtype=np.uint32).view(np.int32)[0])
            torch.cuda.synchronize()
            with pytest.raises(pybsgs.BsgsError, match="fingerprint"):
                dev.install_table_ext_device(bad.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)
        dev.close()
