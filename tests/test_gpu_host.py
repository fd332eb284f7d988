"""GPU end-to-end tests of the C++ host (bsgs-cuda_amd/host/): the reference's command line,
file formats and outputs, with the reference's own known-key vectors (1_9_7File.pb:189, 191, 200-203)."""
import hashlib
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")

PUB_1E9AD = ("e1e5e6f7b0b8d67604e3940c87bf06b814cedc486112b9956c68e3d78b1bd812"
             "97fe4f65fbd6e9f7eb1eea80b144d1487f2a9b0aeae5fcf6f43b41491641884e")
PUB_65BIT = "036d05521c67b9cc1c0ef906b42215c7120c7302c34d9316a2726199bedac50936"
PUB_PUZZLE64 = "03100611c54dfef604163b8358f7b7fac13ce478e02cb224ae16d45526b25d9d4d"


def run(args, cwd, timeout=600):
    assert os.path.exists(EXE), "host binary missing: run __graft_entry__.build()"
    res = subprocess.run([EXE, "-dir", str(cwd)] + args, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    return res.stdout


def win_lines(cwd):
    with open(os.path.join(cwd, "win.txt"), "rb") as f:
        return f.read().decode().split("\r\n")


def test_onlygen_config1_files(tmp_path):
    """BASELINE config 1: onlygen -w 20 -htsz 18 -> reference-named files, byte-exact (sha256 from the plain-Python generator)"""
    with open(os.path.join(ROOT, "tests", "golden", "cfg1_digests.json")) as f:
        d = json.load(f)
    run(["-onlygen", "-t", str(d["g2_t"]), "-b", str(d["g2_b"]), "-p", str(d["g2_p"]), "-w", "20", "-htsz", "18"], tmp_path)
    for name, size, sha in ((d["htgpu_name"], d["htgpu_size"], d["htgpu_sha256"]), (d["htcpu_name"], d["htcpu_size"], d["htcpu_sha256"]),
                            (d["g2_name"], d["g2_size"], d["g2_sha256"])):
        blob = open(os.path.join(tmp_path, name), "rb").read()
        assert len(blob) == size and hashlib.sha256(blob).hexdigest() == sha, name
    # second run loads the files instead of regenerating them
    out = run(["-onlygen", "-t", str(d["g2_t"]), "-b", str(d["g2_b"]), "-p", str(d["g2_p"]), "-w", "20", "-htsz", "18"], tmp_path)
    assert "Both HT files exist" in out and "Load BIN file" in out


def test_key_1e9ad(tmp_path):
    out = run(["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14", "-pb", PUB_1E9AD, "-pk", "1"], tmp_path)
    lines = win_lines(tmp_path)
    assert lines[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD
    assert lines[1].endswith("Pub: 02" + PUB_1E9AD[:64]) and lines[1].startswith(" " * 3)
    assert "KEY[1]" in out


def test_key_1e9ad_files_built_on_the_host_cpu_and_htcpu_searched_in_its_file(tmp_path):
    """the two host-side modes of the reference around the same search: files made by the CPU-only generator (-onlygen -cpugen: onlygen1_9_6File.pb) are what the
    engine consumes, and with -sf 1 (the reference's default, 1_9_7File.pb:178) the resolver reads htCPU from its file (two reads per hit) instead of loading it"""
    geo = ["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14"]
    out = run(["-onlygen", "-cpugen"] + geo, tmp_path)
    assert "on the host CPU" in out and "onlygen: files ready" in out
    out = run(geo + ["-pb", PUB_1E9AD, "-pk", "1"], tmp_path)                            # -sf 1 is the default, as in the reference
    assert "Both HT files exist" in out and "htCPU is searched in its file" in out
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD
    os.remove(os.path.join(tmp_path, "win.txt"))
    out = run(geo + ["-pb", PUB_1E9AD, "-pk", "1", "-sf", "0"], tmp_path)                 # in RAM: the same key
    assert "htCPU is searched in its file" not in out and "Search in RAM" in out and win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD
    # -cpugen without -onlygen: the missing files are built on the CPU, then the GPU searches
    fresh = tmp_path / "fresh"
    fresh.mkdir()
    out = run(geo + ["-pb", PUB_1E9AD, "-pk", "1", "-cpugen"], fresh)
    assert "on the host CPU" in out and win_lines(fresh)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD


@pytest.mark.parametrize("geo", [["-t", "64", "-b", "8", "-p", "16", "-w", "100003", "-htsz", "14"],      # -w above 36: a decimal count (1_9_7File.pb:1009-1022)
                                 ["-t", "96", "-b", "5", "-p", "6", "-w", "77777", "-htsz", "13"],        # 480 engine threads: a ragged last wave
                                 ["-t", "32", "-b", "3", "-p", "10", "-w", "65537", "-htsz", "16"]])      # 96 engine threads, one entry per bucket
def test_ragged_geometry_and_decimal_w(tmp_path, geo):
    """baby-step counts that are not powers of two, thread counts that are not multiples of a wave: the tables are built, saved,
    found again on the second run, and the key is recovered from both ends of the range; an odd -p is refused like the reference
    refuses it (1_9_7File.pb:4616-4618)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    w = int(geo[7])
    out = run(geo + ["-pb", PUB_1E9AD, "-pk", "1"], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD
    assert any(("_%d_" % w) in f and f.endswith("_htGPUv0.BIN") for f in os.listdir(tmp_path))
    key = 0x7000000000 + 12345
    out = run(geo + ["-pb", "%064x%064x" % ecpy.mul(key), "-pk", "7000000000", "-pke", "7100000000"], tmp_path)
    assert "Both HT files exist" in out
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % key
    key = 0x7100000000 - 3                                                   # the far end of the range
    run(geo + ["-pb", "%064x%064x" % ecpy.mul(key), "-pk", "7000000000", "-pke", "7100000000"], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % key
    odd = list(geo); odd[5] = "7"
    res = subprocess.run([EXE, "-dir", str(tmp_path)] + odd + ["-pb", PUB_1E9AD, "-pk", "1"], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "-p must be even" in res.stderr


def test_key_65bit_default_range_and_puzzle64(tmp_path):
    """the reference's default job (1_9_7File.pb:191, 197, 210) and the puzzle-64 vector (200-203)"""
    geo = ["-t", "256", "-b", "64", "-p", "256", "-w", "26", "-htsz", "24"]
    run(geo + ["-pb", PUB_65BIT], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x16f7027bbf8454a5c
    out = run(geo + ["-pb", PUB_PUZZLE64, "-pk", "8000000000000000", "-pke", "ffffffffffffffff"], tmp_path)
    assert "Both HT files exist" in out                                  # tables were reused from the first run
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0xf7051f27b09112d4


def test_infile_sequential_and_recovery(tmp_path):
    """-infile: several keys searched one after the other; -wl: resume from a currentwork.txt (4 CRLF lines,
    SHA1 configuration fingerprint, 1_9_7File.pb:3911-3917, 4635-4686)"""
    import oracle_lib as O
    keys = [0x1E9AD, 2, 0x7A5B3C]
    pubs = []
    for k in keys:
        x, y = O.pt_mul(k)
        pubs.append("%064x%064x" % (x, y))
    infile = tmp_path / "pubs.txt"
    infile.write_text("\n".join(pubs) + "\n")
    geo = ["-t", "64", "-b", "4", "-p", "8", "-w", "14", "-htsz", "12"]
    run(geo + ["-infile", str(infile), "-pk", "1"], tmp_path)
    lines = win_lines(tmp_path)
    assert [lines[0], lines[2], lines[4]] == ["KEY[%d]: 0x%064x" % (i + 1, k) for i, k in enumerate(keys)]
    # the checkpoint never names a finished job: when job 2 ended it was rewritten for position 3, from its start -- the timer (-wt) alone would have left
    # position 1 or 2 standing
    cw = (tmp_path / "currentwork.txt").read_bytes().decode().split("\r\n")
    assert cw[0] == "3" and cw[1] == pubs[2] and int(cw[2], 16) == 1, cw
    # a restart from that checkpoint with win.txt still in place: position 3 is reported already -> not searched again, no second KEY[3]
    out = run(geo + ["-infile", str(infile), "-pk", "1", "-wl", str(tmp_path / "currentwork.txt")], tmp_path)
    assert win_lines(tmp_path) == lines and "Found 0 of 3" in out
    # recovery: position 3, counter just below the tile that holds the key
    t, b, p, w, htsz = 64, 4, 8, 1 << 14, 12
    gstep = 4 * t * b * p * w
    cnt = 1 + (max(0, (keys[2] - 1) // gstep - 1)) * gstep
    fp = hashlib.sha1(("%d%d%d%d%s%s%d" % (t, b, p, w, "1", "1ffffffffffffffff", htsz)).encode()).hexdigest()
    rec = tmp_path / "currentwork.txt"
    rec.write_bytes(("3\r\n%s\r\n%064x\r\n%s\r\n" % (pubs[2], cnt, fp)).encode())
    os.remove(tmp_path / "win.txt")
    out = run(geo + ["-infile", str(infile), "-pk", "1", "-wl", str(rec)], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[3]: 0x%064x" % keys[2]
    assert "Recovery: listpos 3" in out
    # a fingerprint made with other settings is refused
    rec.write_bytes(("3\r\n%s\r\n%064x\r\n%s\r\n" % (pubs[2], cnt, "0" * 40)).encode())
    res = subprocess.run([EXE, "-dir", str(tmp_path)] + geo + ["-infile", str(infile), "-pk", "1", "-wl", str(rec)], capture_output=True, text=True)
    assert res.returncode != 0


def test_config4_style_many_keys_64bit_range(tmp_path):
    """BASELINE config 4 in small: -infile with 16 public keys searched sequentially over the fixed 64-bit range
    8000000000000000..ffffffffffffffff (devices are loaded once, the dispenser is re-seeded per key)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    keys, st = [], 0xC0FFEE
    for _ in range(16):
        st, r = ecpy.splitmix64(st)
        keys.append((1 << 63) + (r >> 1))
    infile = tmp_path / "pubs.txt"
    infile.write_text("\n".join("%064x%064x" % ecpy.mul(k) for k in keys) + "\n")
    out = run(["-t", "256", "-b", "128", "-p", "256", "-w", "28", "-htsz", "26", "-infile", str(infile),
               "-pk", "8000000000000000", "-pke", "ffffffffffffffff"], tmp_path)
    lines = win_lines(tmp_path)
    got = [int(l.split("0x")[1], 16) for l in lines if l.startswith("KEY[")]
    assert got == keys
    # devices are opened and loaded once, not once per key; round 5: jobs this short (1025 tiles) run two at a time, each on an engine of its own on the one GPU
    assert out.count("memory") == 2 and "two public keys searched side by side" in out
    assert [l for l in out.splitlines() if l.startswith("Findpubkey")] == ["Findpubkey  : " + ("03" if ecpy.mul(k)[1] & 1 else "02") + "%064x" % ecpy.mul(k)[0] for k in keys]   # console in list order


def test_extended_table_mode_small_and_puzzle64(tmp_path):
    """-ext: bucket lines + overflow list built in GPU memory, no HT files, hits resolved by the checker's own small BSGS
    instead of an htCPU lookup -- same keys as the file-backed runs above"""
    out = run(["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14", "-pb", PUB_1E9AD, "-pk", "1", "-ext"], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD
    assert "extended table" in out and not [f for f in os.listdir(tmp_path) if "htGPU" in f or "htCPU" in f]
    run(["-t", "256", "-b", "64", "-p", "256", "-w", "26", "-htsz", "22", "-ext", "-pb", PUB_PUZZLE64,
         "-pk", "8000000000000000", "-pke", "ffffffffffffffff"], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0xf7051f27b09112d4


def test_config3_style_w34_80bit_range(tmp_path):
    """BASELINE config 3 geometry: single key in an 80-bit range with -w 34 -htsz 31 (2^34 baby points, ~130 GiB of HBM,
    beyond the reference's file format).  The key sits 2^65 into the range so the test stays short."""
    import sys
    import torch
    from conftest import free_hbm
    if free_hbm(200 * 2**30) < 200 * 2**30:
        pytest.skip("needs ~150 GiB of free HBM")
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    key = (1 << 79) + 0x2123456789ABCDEF0
    out = run(["-t", "256", "-b", "256", "-p", "256", "-w", "34", "-htsz", "31", "-pb", "%064x%064x" % ecpy.mul(key),
               "-pk", "80000000000000000000", "-pke", "ffffffffffffffffffff"], tmp_path, timeout=900)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % key
    assert "17179869184 items" in out


def test_key_in_the_last_tile_of_a_range(tmp_path):
    """1_9_7File.pb:2512-2518 tests the tile counter AFTER the launch, so the first tile whose counter exceeds the range
    width is still searched; a key just below -pke lives there (a tile reaches 2w*maxnonce - p*w below its counter).
    (Found by the 1000-key config-4 run: 4 keys near ffff... were missed before the dispenser did the same.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    # -t 64 -b 8 -p 16 -w 16: tile stride 2^31, counters 1 + j*2^31; width 3*2^31 - 5 ends between tile 2 and tile 3
    start, width = 1, 3 * 2**31 - 5
    key = start + width - 10
    run(["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14", "-pb", "%064x%064x" % ecpy.mul(key),
         "-pk", "%x" % start, "-pke", "%x" % (start + width)], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % key
    # and a key beyond -pke + one tile is not looked for: the job ends with "Reached end of space"
    out = run(["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14", "-pb", "%064x%064x" % ecpy.mul(start + width + 3 * 2**31),
               "-pk", "%x" % start, "-pke", "%x" % (start + width)], tmp_path)
    assert "Reached end of space" in out and "Found 0 of 1" in out


def test_default_end_range_applies_without_pke(tmp_path):
    """the reference always has an end of range: -pke defaults to 1ffffffffffffffff (1_9_7File.pb:210, 4897-4936).  A key
    beyond it ends the job with "Reached end of space" instead of searching forever; an -infile job moves on to the next key."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    geo = ["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14"]
    pk = 0x1ffffffffff000000                                     # 2^24 below the default end (2^65 - 1)
    inside, outside = pk + 0x123456, pk + 2**40
    infile = tmp_path / "pubs.txt"
    infile.write_text("%064x%064x\n%064x%064x\n" % (ecpy.mul(outside) + ecpy.mul(inside)))
    out = run(geo + ["-infile", str(infile), "-pk", "%x" % pk], tmp_path, timeout=120)
    assert "END RANGE= %064x" % 0x1ffffffffffffffff in out and "WIDTH RANGE=" in out
    assert out.count("Reached end of space") == 1 and "Found 1 of 2" in out
    assert win_lines(tmp_path)[0] == "KEY[2]: 0x%064x" % inside
    # the reference's check of the start against the (default) end
    res = subprocess.run([EXE, "-dir", str(tmp_path)] + geo + ["-pb", PUB_65BIT, "-pk", "2ffffffffffffffff"], capture_output=True, text=True)
    assert res.returncode != 0 and "End range" in res.stderr


def test_two_engines_share_one_dispenser_and_checkpoint_is_min_in_flight(tmp_path):
    """-d 0,0: two driver threads (two engines, the second one a device-to-device replica of the first) share the GetJob
    dispenser (1_9_7File.pb:2077-2092, 4769-4815).  Every tile is handed out exactly once, both threads work, every
    checkpoint equals the smallest counter in flight (1_9_7File.pb:3904-3911), and the key is found."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    t, b, p, w = 64, 8, 16, 1 << 16
    gstep = 4 * t * b * p * w
    key = 1 + 3000 * gstep + 12345                               # ~3000 tiles = ~63 batches of 48
    log = tmp_path / "jobs.log"
    out = run(["-t", str(t), "-b", str(b), "-p", str(p), "-w", "16", "-htsz", "14", "-pb", "%064x%064x" % ecpy.mul(key), "-pk", "1",
               "-d", "0,0", "-joblog", str(log)], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x%064x" % key
    assert "replicated to 1 more GPU engine" in out and out.count("job finished") == 2
    taken, inflight, slots, next_cnt, saves = [], {}, set(), 1, 0
    for line in log.read_text().splitlines():
        f = line.split()
        if f[0] == "take":
            slot, first, n, cnt = int(f[1]), int(f[2]), int(f[3]), int(f[4], 16)
            assert cnt == 1 + first * gstep and first == (taken[-1][0] + taken[-1][1] if taken else 0)   # consecutive, no gap, no overlap
            taken.append((first, n))
            inflight[slot] = cnt
            slots.add(slot)
            next_cnt = cnt + n * gstep
        elif f[0] == "done":
            inflight.pop(int(f[1]))
        elif f[0] == "save":
            saves += 1
            assert int(f[1], 16) == min(list(inflight.values()) + [next_cnt])
    assert slots == {0, 1} and saves > 0
    assert sum(n for _, n in taken) >= 3000 and taken[0][0] == 0


def test_puzzle64_at_config2_flags(tmp_path):
    """BASELINE config 2 at its exact flags: -t 256 -b 256 -p 256 -w 26 -htsz 25 on the puzzle-64 vector
    (1_9_7File.pb:200-203); the measured time-to-solve is printed and, on the GPU box, recorded under gpurun_out/."""
    import time
    geo = ["-t", "256", "-b", "256", "-p", "256", "-w", "26", "-htsz", "25"]
    run(geo + ["-onlygen"], tmp_path)                            # table + giants files first: the solve below loads them
    t0 = time.time()
    out = run(geo + ["-pb", PUB_PUZZLE64, "-pk", "8000000000000000", "-pke", "ffffffffffffffff"], tmp_path)
    wall = time.time() - t0
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0xf7051f27b09112d4
    job = [l for l in out.splitlines() if l.startswith("Job time")][0].split()
    job_s, tiles = float(job[2].rstrip("s,")), int(job[3])
    # functional: the key, and that the whole range up to it was walked.  The time itself belongs to the records (bench.py `measured_solve`: 1.7 s); here only
    # a bound no healthy MI355X comes near, so that a busy or slower box does not turn a correctness test red
    assert tiles >= (0xf7051f27b09112d4 - 0x8000000000000000) // (4 * 2**24 * 2**26) and job_s < 30
    rec = {"config": " ".join(geo), "key": "0xf7051f27b09112d4", "job_time_s": job_s, "tiles": tiles, "giant_steps": tiles * 2**25,
           "giant_steps_per_s": tiles * 2**25 / job_s, "process_wall_s_including_file_load_and_upload": wall}
    print("puzzle64 at config-2 flags:", json.dumps(rec))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "puzzle64_config2.json"), "w") as f:
            json.dump(rec, f)
    except OSError:
        pass


def test_config4_true_flags_100_keys(tmp_path):
    """BASELINE config 4 at its true flags (-t 256 -b 256 -p 256 -w 30 -htsz 28): 100 public keys searched sequentially over the
    fixed 64-bit range, devices loaded once, the dispenser re-seeded per key"""
    import sys
    import time
    import torch
    from conftest import free_hbm
    if free_hbm(60 * 2**30) < 60 * 2**30:
        pytest.skip("needs ~40 GiB of free HBM")
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    keys, st = [], 0xC0FFEE4
    for _ in range(100):
        st, r = ecpy.splitmix64(st)
        keys.append((1 << 63) + (r >> 1))
    infile = tmp_path / "pubs.txt"
    G2k = {}
    lines = []
    for k in keys:
        x, y = ecpy.mul(k)
        lines.append(("03" if y & 1 else "02") + "%064x" % x)      # compressed form: exercises the decompression path too
    infile.write_text("\n".join(lines) + "\n")
    t0 = time.time()
    out = run(["-t", "256", "-b", "256", "-p", "256", "-w", "30", "-htsz", "28", "-infile", str(infile),
               "-pk", "8000000000000000", "-pke", "ffffffffffffffff"], tmp_path, timeout=1500)
    wall = time.time() - t0
    got = [int(l.split("0x")[1], 16) for l in win_lines(tmp_path) if l.startswith("KEY[")]
    assert got == keys
    job_times = [float(l.split()[2].rstrip("s,")) for l in out.splitlines() if l.startswith("Job time")]
    rec = {"keys": len(keys), "found": len(got), "wall_s": wall, "sum_job_time_s": sum(job_times), "max_job_time_s": max(job_times)}
    print("config 4 (true flags, 100 keys):", json.dumps(rec))
    try:
        with open(os.path.join(ROOT, "gpurun_out", "config4_100keys.json"), "w") as f:
            json.dump(rec, f)
    except OSError:
        pass


def test_config5_style_two_engines_extended_table_120bit_range(tmp_path):
    """BASELINE config 5 in small: a 120-bit range, the extended table (beyond the reference's file format) on two engines -- each builds
    its own replica (the default start-up strategy for extended tables; tests/test_gpu_round5.py runs all three) --, both driver threads
    sharing the dispenser.  (Config 5 proper is 8 GPUs with -w 34; one 288 GB GPU holds two engines at -w 33 -htsz 30.)"""
    import sys
    import torch
    from conftest import free_hbm
    if free_hbm(220 * 2**30) < 220 * 2**30:
        pytest.skip("needs ~180 GiB of free HBM")
    sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
    from pybsgs import ecpy
    key = (1 << 119) + (300 << 59) + 0x123456789ABCDEF               # 300 tiles of 2^59 keys into the range: both engines get batches
    out = run(["-t", "256", "-b", "256", "-p", "256", "-w", "33", "-htsz", "30", "-d", "0,0", "-pb", "%064x%064x" % ecpy.mul(key),
               "-pk", "%x" % (1 << 119), "-pke", "%x" % ((1 << 120) - 1)], tmp_path, timeout=1200)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % key
    # round 5: an extended table is BUILT by every engine (start-up strategy "local": no link traffic; the builds are deterministic, so the replica verification
    # compares them like copies); the giants come from the host's image on every engine
    assert out.count("strategy local") == 2 and out.count("engine(s) ready in") == 1 and out.count("job finished") == 2
    assert "Replica verification: 2 engines hold identical tables" in out
    # 64 GiB of bucket lines per engine: the first engine's allocator held a memory group back for its chain scratch and
    # the scratch came from it; the replica allocates its lines through the same allocator (on its OWN GPU in config 5 proper; here it
    # shares GPU 0 with the first engine, so whether a whole group is still free for it depends on what the first one left)
    place = [ln for ln in out.splitlines() if "chain scratch in" in ln]
    # (where the scratch ends up is the grader's decision, taken on timings: tests/test_gpu_round3.py::test_recv_buffers_above_40GiB_reserve_a_memory_group pins the
    # reserve on an engine that has the GPU to itself; here two engines share one GPU and memory a previous test freed may still be being wiped: both must HAVE scratch)
    assert len(place) == 2 and "engine 0" in place[0] and " 0 piece(s)" not in place[0] and "reserved group:" in place[0]
    assert "engine 1" in place[1] and " 0 piece(s)" not in place[1]
    print("\n".join(place))
    assert "WIDTH RANGE=" in out and "= 2^119" in out


def test_host_centres_and_reference_quirk_flags(tmp_path):
    """-hostcentres (tile centres added on the host and uploaded, the reference's way) and -refquirks (NEGMODP bug reproduced) find
    the same keys as the defaults"""
    geo = ["-t", "64", "-b", "8", "-p", "16", "-w", "16", "-htsz", "14", "-pb", PUB_1E9AD, "-pk", "1"]
    out = run(geo + ["-hostcentres"], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD and "Checker:" in out
    out = run(geo + ["-refquirks"], tmp_path)
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD and "Reference-quirk mode" in out
    out = run(geo + ["-tune"], tmp_path)                                  # start-up choice of the buffer placement by timed launches
    assert win_lines(tmp_path)[0] == "KEY[1]: 0x" + "%064x" % 0x1E9AD and "placement tuned" in out
