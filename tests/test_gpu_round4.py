"""Round-4 GPU tests:
  * replica verification: device-side table checksums are equal across byte-identical replicas and change with ONE flipped bit, in every table
    layout; bench.py's N > 1 line carries the evidence and a corrupted replica on rank 1 turns the run red; the C++ host does the same after
    bsgs_broadcast_tables;
  * the reference's own default geometry (-t 256 -b 132 -p 400 -w 25 -htsz 25, 1_9_7File.pb:181-184, 168): per key on the shipped kernel."""
import json
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")


@pytest.fixture(scope="module")
def O():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.mark.parametrize("layout", [2, 4, 1, 3])
def test_table_checksums_equal_across_replicas_and_one_bit_changes_them(O, layout):
    import pybsgs
    from test_gpu_round2 import _random_table
    t, b, p, w, htsz = 64, 4, 8, 1 << 16, 12                       # 16 per bucket: 64-byte lines overflow (CSR / overflow set in use)
    g2 = O.build_g2(t, b, p, w)
    gpu = _random_table(O, random.Random(99), w, htsz, [])
    a, c = pybsgs.Device(0), pybsgs.Device(0)
    a.upload_g2(g2, t, b, p)
    a.upload_htgpu(gpu, 1 << htsz, w, layout)
    pybsgs.broadcast_tables([a, c])
    sa, sc = a.table_checksum(), c.table_checksum()
    assert sa == sc and sa[3] != 0
    assert (sa[0] != 0) == (layout != 1) and (sa[1] != 0) == (layout == 4) and (sa[2] != 0) == (layout != 4)
    # the same table uploaded afresh gives the same sums (they describe the contents, not the allocation)
    e = pybsgs.Device(0)
    e.upload_g2(g2, t, b, p)
    e.upload_htgpu(gpu, 1 << htsz, w, layout)
    assert e.table_checksum() == sa
    # one bit, anywhere: first word, a middle byte, the last byte the sums cover
    nbytes = ((64 if layout in (2, 4) else 128) << htsz) if layout != 1 else 4 * ((1 << htsz) + 1) + 4 * w      # (an odd number of 32-bit words)
    for off in (0, nbytes // 2 + 3, nbytes - 1):
        c.debug_corrupt_table(off, 0x04)
        bad = c.table_checksum()
        assert bad != sa and bad[3] == sa[3], off
        c.debug_corrupt_table(off, 0x04)                            # flip back
        assert c.table_checksum() == sa
    with pytest.raises(pybsgs.BsgsError):
        c.debug_corrupt_table(1 << 40, 1)
    # position-dependent: another geometry holds the same multiset of giants' bytes in another order
    e.upload_g2(O.build_g2(t, b * 2, p // 2, w), t, b * 2, p // 2)
    assert e.table_checksum()[3] != sa[3] and e.table_checksum()[:3] == sa[:3]
    for d in (a, c, e):
        d.close()


def _bench(args, env_extra=None, timeout=1500, expect_rc=0):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    # (ranks exit with 3 on a failed verification; torch.distributed.run reports any failed rank as 1)
    assert (r.returncode != 0) if expect_rc == "nonzero" else (r.returncode == expect_rc), (r.returncode, r.stdout[-3000:] + r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("table", ["csr_image", "extended"])
def test_bench_turns_red_when_rank_1_holds_a_corrupted_replica(table):
    """`bench.py --gpus 2 --same-device` with BENCH_CORRUPT_RANK=1: rank 1 flips one bit of the table it received; the run must end with a
    non-zero exit code, no rate, and say which rank differs"""
    common = ["--w", "26", "--htsz", "25", "--tiles-per-launch", "48", "--steps", "2", "--warmup", "1", "--warmup-s", "0", "--sustain-s", "0",
              "--no-cpu-baseline", "--no-solve", "--no-pmc", "--gpus", "2", "--same-device"] + (["--force-ext", "--startup-strategy", "broadcast"] if table == "extended" else [])
    bad = _bench(common, env_extra={"BENCH_CORRUPT_RANK": "1"}, expect_rc="nonzero")
    assert bad["value"] is None and bad["error"] == "replica verification FAILED" and bad["ranks_differing_from_rank0"] == [1]
    assert bad["verification"]["table_checksum_equal"] is False
    assert bad["checksums_per_rank"][0][0] != bad["checksums_per_rank"][1][0] and bad["checksums_per_rank"][0][3] == bad["checksums_per_rank"][1][3]


def test_host_verifies_replicas_and_stops_on_a_corrupted_one(tmp_path):
    """bsgs_mi355x -d 0,0: after bsgs_broadcast_tables the engines' checksums and the hits of one probe tile are compared; with one bit flipped
    in engine 1's table (BSGS_TEST_CORRUPT_ENGINE=1) the run stops before searching"""
    from pybsgs import ecpy
    key = 0xABCDE
    x, y = ecpy.mul(key)
    args = [HOST, "-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "12", "-dir", str(tmp_path), "-d", "0,0",
            "-pb", "%02x%064x" % (2 + (y & 1), x), "-pk", "1", "-pke", "ffffff"]
    ok = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0, ok.stdout[-2000:] + ok.stderr[-2000:]
    assert "Replica verification: 2 engines hold identical tables" in ok.stdout and "KEY[1]: 0x" + "%064x" % key in ok.stdout
    bad = subprocess.run(args, capture_output=True, text=True, timeout=600, env=dict(os.environ, BSGS_TEST_CORRUPT_ENGINE="1"))
    assert bad.returncode != 0 and "replica verification FAILED" in (bad.stdout + bad.stderr) and "KEY[1]" not in bad.stdout
    skip = subprocess.run(args + ["-noverify"], capture_output=True, text=True, timeout=600, env=dict(os.environ, BSGS_TEST_CORRUPT_ENGINE="1"))
    assert skip.returncode == 0 and "Replica verification" not in skip.stdout


def test_overflow_bound_invariant_is_checked_at_install(O):
    """the probe never looks into the overflow set for a hash below an over-full line's last word, so a lines + overflow-set table
    that does not keep its smallest hashes in the line would miss hits silently.  Such a table is refused: (i) a built table whose one line's
    bound was raised; (ii) an htGPU image whose buckets are NOT sorted, asked for in a LIST layout -- which the exact layouts still search."""
    import torch
    import pybsgs
    from test_gpu_round2 import _random_table
    w, htsz, lay = 1 << 16, 10, pybsgs.TABLE_LINES64_LIST           # 64 per bucket: every 64-byte line is over-full
    dev = pybsgs.Device(0)
    cap = dev.ext_overflow_capacity(w, htsz, lay)
    lines = torch.empty((1 << htsz) * 16, dtype=torch.int32, device="cuda:0")
    ovf = torch.empty(cap, dtype=torch.int64, device="cuda:0")
    n_ovf, n_over = dev.build_baby_table_ext_device(w, htsz, lay, lines.data_ptr(), ovf.data_ptr(), cap)
    assert n_over == 1 << htsz
    dev.install_table_ext_device(lines.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)      # as built: accepted
    keep = int(lines[16 * 77 + 15])
    lines[16 * 77 + 15] = -1                                         # line 77: last word 0xFFFFFFFF -> its set entries are now "below the bound"
    torch.cuda.synchronize()
    with pytest.raises(pybsgs.BsgsError, match="overflow bound"):
        dev.install_table_ext_device(lines.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)
    lines[16 * 77 + 15] = keep
    lines[16 * 5 + 3] = -1                                           # line 5: an entry above its last word
    torch.cuda.synchronize()
    with pytest.raises(pybsgs.BsgsError, match="overflow bound"):
        dev.install_table_ext_device(lines.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)
    # (ii) an image with one bucket's hashes reversed
    import numpy as np
    gpu = _random_table(O, random.Random(3), w, 12, [])              # 16 per bucket at htsz 12
    img = np.frombuffer(gpu, dtype=np.uint32).copy()
    items = 1 << 12
    bkt = next(k for k in range(items) if int(img[k + 1]) - int(img[k]) >= 18)       # an over-full bucket (more than the 15 entries of a line)
    lo, hi = int(img[bkt]), int(img[bkt + 1])
    img[items + 1 + lo: items + 1 + hi] = img[items + 1 + lo: items + 1 + hi][::-1]
    with pytest.raises(pybsgs.BsgsError, match="overflow bound"):
        dev.upload_htgpu(img.tobytes(), items, w, pybsgs.TABLE_LINES64_LIST)
    dev.upload_htgpu(gpu, items, w, pybsgs.TABLE_LINES64_LIST)      # the sorted image: fine
    dev.close()


def test_shipped_kernel_per_key_parity_at_the_reference_default_geometry(O):
    """The reference's own defaults -t 256 -b 132 -p 400 (1_9_7File.pb:181-184): 33792 reference threads x 400 giants; the engine batches them as 16896 threads x
    800 giants -- the only quad-chain batch length in use that is not a power of two (200 groups of four, 66 blocks per tile: not a multiple of 8, so the
    plain block -> tile map).  Every key of 7 engine threads in 3 tiles of a walk launch, one by one, on the production instantiation; then a ONE-tile launch
    (the reference's launch pattern) on the narrow batching this geometry gets (33792 x 400 -> 67584 x 200: 200 is still a multiple of 4)."""
    import numpy as np
    import pybsgs
    from pybsgs import ecpy
    t, b, p, w, htsz = 256, 132, 400, 1 << 25, 13
    dev = pybsgs.Device(0)
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    g2 = np.frombuffer(dev.download_g2(64 * t * b * p), dtype=np.uint8)
    Ti, pi = dev.engine_geometry()
    assert (Ti, pi) == (16896, 800)
    ratio = pi // p
    _, stride = ecpy.tile_stride(t, b, p, w)
    p0 = ecpy.mul(0x7654321 * 2 * w + 99)
    dev.set_walk(p0, stride)
    NT, first = 40, 5000
    centres = dev.walk_centres(first, NT)
    tiles = [0, 23, NT - 1]
    qs = [0, 255, 256, 8447, 8448, Ti - 257, Ti - 1]                 # block boundaries, the middle, the last block (66 blocks: Ti = 66 * 256)
    keys = []
    for tl in tiles:
        for q in qs:
            keys.append(O.tile_slice_keys(centres[tl], g2, t, b, p, q * ratio, (q + 1) * ratio).reshape(-1))
    allkeys = np.concatenate(keys)
    assert len(allkeys) == len(tiles) * len(qs) * 2 * pi
    gpu_img, _ = O.pack_tables_from_keys(allkeys, htsz)
    dev.upload_htgpu(gpu_img, 1 << htsz, len(allkeys), pybsgs.TABLE_LINES64)
    dev.set_tiles_per_launch(NT)
    hits, n, _ = dev.run_walk(first, NT, 65536)
    assert dev.last_kernel() == "giant_pair2_kernel<2, false, true>" and dev.last_batching() == (Ti, pi) and n == len(hits)
    got = {}
    for tile, code, idx in hits:
        got.setdefault(tile, set()).add((code, idx))
    ht = np.frombuffer(gpu_img, dtype=np.uint8)
    for tl in tiles:
        mine = got.get(tl, set())
        for q in qs:
            lo, hi = q * pi, (q + 1) * pi
            for i in range(lo, hi):
                assert (2, i) in mine and ((1, i) in mine or (4, i) in mine), (tl, q, i)
            ref, nref, _ = O.tile_slice_digest(centres[tl], g2, t, b, p, ht, htsz, q * ratio, (q + 1) * ratio, max_hits=8192)
            assert sorted((c, i) for c, i in mine if lo <= i < hi) == sorted(ref), (tl, q)
    # one tile per launch: narrow batching, same hits for that tile
    dev.set_tiles_per_launch(0)
    one, n1, _ = dev.run_walk(first + 23, 1, 65536)
    assert dev.last_batching()[1] in (200, 400) and dev.last_batching()[0] * dev.last_batching()[1] == t * b * p
    assert sorted((c, i) for _, c, i in one) == sorted(got[23])
    dev.close()
