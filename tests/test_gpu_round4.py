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
    assert r.returncode == expect_rc, (r.returncode, r.stdout[-3000:] + r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("table", ["csr_image", "extended"])
def test_bench_turns_red_when_rank_1_holds_a_corrupted_replica(table):
    """`bench.py --gpus 2 --same-device` with BENCH_CORRUPT_RANK=1: rank 1 flips one bit of the table it received; the run must end with a
    non-zero exit code, no rate, and say which rank differs"""
    common = ["--w", "26", "--htsz", "25", "--tiles-per-launch", "48", "--steps", "2", "--warmup", "1", "--warmup-s", "0", "--sustain-s", "0",
              "--no-cpu-baseline", "--no-solve", "--no-pmc", "--gpus", "2", "--same-device"] + (["--force-ext"] if table == "extended" else [])
    bad = _bench(common, env_extra={"BENCH_CORRUPT_RANK": "1"}, expect_rc=3)
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
