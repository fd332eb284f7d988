"""CPU tier: the C++ host's GPU-independent logic (bsgs-cuda_amd/host/host_selftest.cpp: bsgs_mi355x -selftest) against hashlib and plain
Python integers: SHA1 / configuration fingerprint (currentwork.txt, 1_9_7File.pb:4635-4636), host EC arithmetic
(csrc/host_secp.h), public-key parsing, the tile dispenser (GetJob, 1_9_7File.pb:2077-2092, 5046-5064) and the table-free
resolver of the extended mode."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def selftest(*items):
    if not os.path.exists(EXE):
        pytest.skip("host binary not built (run __graft_entry__.build())")
    res = subprocess.run([EXE, "-selftest"] + [str(x) for x in items], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return [l.split() for l in res.stdout.splitlines()]


def pt_hex(p):
    return ["%064x" % p[0], "%064x" % p[1]]


def test_sha1_and_fingerprint():
    msgs = ["abc", "a" * 55, "b" * 56, "c" * 64, "Thequickbrownfoxjumpsoverthelazydog" * 5]
    out = selftest(*[x for m in msgs for x in ("sha1", m)])
    assert [o[1] for o in out] == [hashlib.sha1(m.encode()).hexdigest() for m in msgs]
    # Str(t)+Str(b)+Str(p)+Str(w)+pk+pke+Str(htsz)
    fp = selftest("fingerprint")[0][1]
    assert fp == hashlib.sha1(b"25688130982162051" + b"8000000000000000" + b"ffffffffffffffff" + b"28").hexdigest()


def test_host_ec_arithmetic_matches_python_ints():
    from pybsgs import ecpy
    ks = [1, 2, 3, 0x1E9AD, 2**64 - 1, 2**128 + 12345, 2**255 + 7, N - 1, N - 2, 0xF7051F27B09112D4]
    out = selftest(*[x for k in ks for x in ("mul", "%x" % k)])
    assert [o[1:] for o in out] == [pt_hex(ecpy.mul(k)) for k in ks]
    out = selftest("multiples", "5", "1000", "multiples", "%x" % (N - 3), "257")
    assert out[0][1:] == pt_hex(ecpy.mul(5000)) and out[1][1:] == pt_hex(ecpy.mul((N - 3) * 257 % N))


def test_pubkey_parsing():
    from pybsgs import ecpy
    x, y = ecpy.mul(0x16f7027bbf8454a5c)
    comp = ("03" if y & 1 else "02") + "%064x" % x
    other = ("02" if y & 1 else "03") + "%064x" % x
    unc = "04%064x%064x" % (x, y)
    raw = "%064x%064x" % (x, y)
    out = selftest("parse", comp, "parse", other, "parse", unc, "parse", raw, "parse", "02" + "%064x" % 5, "parse", "04" + "11" * 64)
    assert out[0][1:] == pt_hex((x, y)) + [comp]
    assert out[1][1:] == pt_hex((x, ecpy.P - y)) + [other]
    assert out[2][1:] == pt_hex((x, y)) + [comp] and out[3][1:] == pt_hex((x, y)) + [comp]
    assert out[4][1] == "invalid" and out[5][1] == "invalid"        # x = 5 is not on the curve; random bytes are not a point


def test_dispenser_sequence():
    """cnt_j = 1 + j * 4*maxnonce*w ;  centre_j = pub - cnt_j*G - (p*w)*G   """
    from pybsgs import ecpy
    t, b, p, w, n = 64, 8, 16, 65536, 7
    key = 0x1E9AD
    pub = ecpy.mul(key)
    out = selftest("jobs", t, b, p, w, n, "%064x%064x" % pub)
    assert len(out) == n
    gstep = 4 * t * b * p * w
    for j, o in enumerate(out):
        cnt = 1 + j * gstep
        assert int(o[1], 16) == cnt
        assert o[2:] == pt_hex(ecpy.mul((key - cnt - p * w) % N))


def test_table_free_resolver():
    """MiniBsgs.find(m*G, w): every b' in [1, w] with x(b'G) = x(mG), i.e. b' = m or b' = n - m when in range"""
    w = 1 << 20
    ms = [1, 77, w, w - 1, 524289, (1 << 15) + 1, N - 1, N - 77, N - w, w + 1, w + 5, 2 * w, N - w - 1, 0x123456789ABCDEF]
    out = selftest("minibsgs", w, *["%x" % m for m in ms])
    assert out[0] == ["minibsgs_bits", "17"]            # lw / 2 + 7 stored-multiple bits (24 at -w 34)
    for m, o in zip(ms, out[1:]):
        expect = sorted({b for b in (m, N - m) if 1 <= b <= w})
        assert [int(v) for v in o[2:]] == expect, hex(m)
    w = 3000000                                   # not a power of two
    ms = [w, w - 1, w + 1, 2999999, 1500000, N - w]
    out = selftest("minibsgs", w, *["%x" % m for m in ms])
    for m, o in zip(ms, out[1:]):
        assert [int(v) for v in o[2:]] == sorted({b for b in (m, N - m) if 1 <= b <= w}), hex(m)


def test_tune_advice_for_mi355x():
    """the MI355X replacement of Tune (1_9_7File.pb:324-431 sizes -t/-b/-p/-w from free memory and SM count): -w / -htsz from
    free HBM; a 288 GB part is offered the extended table (-w 34 -htsz 31), small parts stay inside the reference format"""
    out = selftest("tune", 290 * 10**9, "tune", 25 * 10**9, "tune", 8 * 10**9)
    big, mid, small = out
    assert big[:5] == ["tune", "-w", "31.52", "-htsz", "31"] and big[5:] == ["ext", "1", "-w", "34", "-htsz", "31"]
    assert mid[:5] == ["tune", "-w", "29.00", "-htsz", "27"] and mid[6] == "0"
    assert small[:5] == ["tune", "-w", "27.00", "-htsz", "25"] and small[6] == "0"
    # the advice never exceeds the reference's table-format limit (1_9_7File.pb:4412-4418) nor the memory it was given
    for o, free in ((big, 290e9), (mid, 25e9), (small, 8e9)):
        w, htsz = 2 ** float(o[2]), int(o[4])
        assert w < 3069485951 * 1.005 and 64 * 2**htsz + 4 * 2**htsz + 4 * w < free       # (-w is printed with two decimals)


def test_tune_plan_for_a_range():
    """Tune for a RANGE (the reference's Tune, 1_9_7File.pb:324-431, knows the GPU only): the table that minimises build + worst-case search.
    A 64-bit range wants a table of 2^30..2^31 points built in GPU memory (no files to bring to the host: 0.3 s all in), an 80-bit range the largest table one
    GPU holds (36 * 2^30 points on 3 * 2^30 bucket lines of 64 bytes), a small GPU stays within its memory, more GPUs shorten the search and never shrink the table."""
    def plan(free, bits, gpus):
        o = selftest("plan", free, bits, gpus)[0]
        f = o[o.index("|") + 1:]
        return {"flags": " ".join(o[1:o.index("|")]), "w": float(f[1]), "htsz": int(f[3]), "ext": f[5] == "1", "build": float(f[7]), "search": float(f[9]), "total": float(f[11])}
    big = 285 * 10**9
    p64 = plan(big, 64, 1)
    assert p64["ext"] and 30 <= p64["w"] <= 31 and p64["total"] < 0.4 and p64["flags"].endswith("-ext")
    p80 = plan(big, 80, 1)
    # 36 * 2^30 points on 3 * 2^30 lines of 64 bytes: the count is no power of two, so the flags name it in decimal (as the reference's -w takes counts, 1_9_7File.pb:1009-1022)
    assert p80["w"] == 35.17 and p80["htsz"] == 3221225472 and "-w 38654705664 -buckets 3221225472" in p80["flags"] and p80["search"] < 600
    p120 = plan(big, 120, 8)
    assert p120["w"] == 35.17 and abs(p120["search"] / plan(big, 120, 1)["search"] - 1 / 8) < 1e-3
    # a GPU with 20 GiB less keeps the 2^35 table on the same lines (its overflow set is 16 GiB instead of 32)
    p80s = plan(big - 20 * 2**30, 80, 1)
    assert p80s["w"] == 35 and "-w 35 -buckets 3221225472" in p80s["flags"]
    small = plan(25 * 10**9, 64, 1)
    assert small["w"] <= 30 and 64 * 2 ** small["htsz"] < 25e9
    tiny = plan(8 * 10**9, 80, 1)
    assert tiny["w"] <= 29 and 64 * 2 ** tiny["htsz"] < 8e9
    # monotone: a larger range never gets a smaller table; and every plan's total is its two parts
    ws = [plan(big, b, 1) for b in (40, 48, 56, 64, 72, 80, 100)]
    assert all(a["w"] <= b["w"] for a, b in zip(ws, ws[1:]))
    assert all(abs(q["build"] + q["search"] - q["total"]) < 2e-3 * max(1.0, q["total"]) for q in ws)


def test_checkpoint_is_the_minimum_in_flight_counter():
    """saveCurentCNT (1_9_7File.pb:3897-3931): the saved counter is the smallest one any GPU has not finished, or the
    dispenser's next counter when all GPUs are idle"""
    assert selftest("checkpoint", "0901", "0301", "-", "0601")[0] == ["save", "%064x" % 0x301]
    assert selftest("checkpoint", "0901", "-", "-")[0] == ["save", "%064x" % 0x901]
    assert selftest("checkpoint", "0901", "0a01", "0b01")[0] == ["save", "%064x" % 0x901]


def test_reference_table_limits_and_unsafe_question():
    """the reference's limits on -w / -htsz for tables in its format and its UNSAFE-mode question (1_9_7File.pb:4412-4472): -w below 3069485951, -htsz below 32,
    "type Y" above the per--htsz duplicate limits (27: 1331331443, 28: 1777178603), and the "-htsz parametr is to low" warning (log2(w) - htsz > 3)."""
    if not os.path.exists(EXE):
        pytest.skip("host binary not built (run __graft_entry__.build())")

    def limits(w, htsz, answer=""):
        r = subprocess.run([EXE, "-selftest", "limits", str(w), str(htsz)], input=answer, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        return r.stdout

    assert limits(1 << 30, 28).splitlines() == ["limits ok"]                                   # BASELINE config 4's table
    assert "limits -w should be less or equil to 3069485951" in limits(3069485951, 29)
    assert limits(3069485950, 29).splitlines()[-1] == "limits ok"
    assert "limits -htsz should be less than 32" in limits(1 << 20, 32)
    for htsz, lim in ((27, 1331331443), (28, 1777178603)):
        assert limits(lim, htsz).splitlines()[-1] == "limits ok" and "UNSAFE" not in limits(lim, htsz)
        out = limits(lim + 1, htsz, "Y\n")
        assert "With -htsz %d value -w should be less or equil to %d" % (htsz, lim) in out and "To continue in UNSAFE mode type Y and press ENTER" in out
        assert out.splitlines()[-1] == "limits ok"
        for answer in ("n\n", "y\n", "", "Yes\n"):
            assert limits(lim + 1, htsz, answer).splitlines()[-1] == "limits exit", answer
    out = limits(1 << 30, 26)                                                                    # 30 - 26 > 3
    assert "WARNING! -htsz parametr is to low, should be at least 28" in out and out.splitlines()[-1] == "limits ok"
    assert "WARNING" not in limits(1 << 30, 27)


def test_onlygen_on_the_host_cpu_config1_and_small_fixture(tmp_path):
    """BASELINE config 1 as it is worded -- "onlygen baby-step table build -w 20 -htsz 18 on host CPU (plumbing, no GPU; bit-exact vs the reference htCPU file)":
    `bsgs_mi355x -onlygen -cpugen` builds htGPU / htCPU / g2 with the host's own EC arithmetic, touches no GPU (this test runs in the CPU tier) and writes the
    reference's file names and bytes (sha256 from the plain-Python generator, tests/golden/gen_golden.py; formats: 1_9_7File.pb:3232-3444, 1831-1903; the reference's
    CPU-only generator: onlygen1_9_6File.pb:2915-3204).  Then the small fixture's geometry (-w 10 -htsz 8 -t 2 -b 2 -p 4): byte for byte the fixture's images."""
    import json
    if not os.path.exists(EXE):
        pytest.skip("host binary not built (run __graft_entry__.build())")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")                    # even on a GPU box: nothing to find
    with open(os.path.join(ROOT, "tests", "golden", "cfg1_digests.json")) as f:
        d = json.load(f)
    args = ["-onlygen", "-cpugen", "-dir", str(tmp_path), "-t", str(d["g2_t"]), "-b", str(d["g2_b"]), "-p", str(d["g2_p"]), "-w", "20", "-htsz", "18"]
    r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "Generate HT with 1048576 items on the host CPU" in r.stdout and "onlygen: files ready" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    for name, size, sha in ((d["htgpu_name"], d["htgpu_size"], d["htgpu_sha256"]), (d["htcpu_name"], d["htcpu_size"], d["htcpu_sha256"]), (d["g2_name"], d["g2_size"], d["g2_sha256"])):
        blob = open(os.path.join(tmp_path, name), "rb").read()
        assert len(blob) == size and hashlib.sha256(blob).hexdigest() == sha, name
    r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=600, env=env)      # a second run loads the files
    assert r.returncode == 0 and "Both HT files exist" in r.stdout and "Load BIN file" in r.stdout
    with open(os.path.join(ROOT, "tests", "golden", "small_w1024_ht8_t2_b2_p4.json")) as f:
        fx = json.load(f)
    small = tmp_path / "small"
    small.mkdir()
    r = subprocess.run([EXE, "-onlygen", "-cpugen", "-dir", str(small), "-t", str(fx["t"]), "-b", str(fx["b"]), "-p", str(fx["p"]), "-w", "10", "-htsz", str(fx["htsz"])],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    gx = "79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798"
    assert open(small / ("%s_%d_%d_htGPUv0.BIN" % (gx, fx["w"], 1 << fx["htsz"])), "rb").read() == bytes.fromhex(fx["htgpu"])
    assert open(small / ("%d_%d_%d_%d_g2.BIN" % (fx["t"], fx["b"], fx["p"], fx["w"])), "rb").read() == bytes.fromhex(fx["g2"])
    # files in the reference's format only
    r = subprocess.run([EXE, "-onlygen", "-cpugen", "-dir", str(small), "-w", "20", "-htsz", "17", "-ext"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "-cpugen builds files in the reference`s format" in r.stderr


def test_htcpu_searched_in_its_file_like_the_reference_default(tmp_path):
    """-sf 1, the reference's default (isFilesearch, 1_9_7File.pb:178): htCPU is not loaded, a lookup is two reads of the file (ReadHTpackFile / compareHTpackFile,
    1_9_7File.pb:3056-3099).  The file lookup must return what the in-RAM lookup returns -- the position k - 1 of k*G for members, nothing for other keys -- on a table
    built by -cpugen (w = 2^16 in 2^10 buckets: 64 entries per bucket)."""
    from pybsgs import ecpy
    if not os.path.exists(EXE):
        pytest.skip("host binary not built (run __graft_entry__.build())")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([EXE, "-onlygen", "-cpugen", "-dir", str(tmp_path), "-t", "2", "-b", "2", "-p", "4", "-w", "16", "-htsz", "10"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    f = os.path.join(tmp_path, "79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798_65536_1024_htCPUv0.BIN")
    ks = [1, 2, 3, 1000, 32768, 65535, 65536]
    keys = [ecpy.mul(k)[0] & (2**64 - 1) for k in ks]
    outside = [ecpy.mul(k)[0] & (2**64 - 1) for k in (65537, 70000, 2**40 + 7)]
    out = selftest("htlookup", f, 10, *["%x" % k for k in keys + outside])
    for k, o in zip(ks, out):
        ram, fil = o[o.index("ram") + 1:o.index("|")], o[o.index("file") + 1:]
        assert ram == fil == [str(k - 1)], (k, o)
    for o in out[len(ks):]:
        assert o[o.index("ram") + 1:o.index("|")] == o[o.index("file") + 1:] == [], o
