#!/usr/bin/env python3
"""BASELINE config 4: -infile with N public keys searched sequentially over the fixed 64-bit range 8000000000000000..
ffffffffffffffff at -w 30 -htsz 28 on one GPU (k_n = 2^63 + splitmix64(n) >> 1).  Prints one JSON line.

  tools/config4_run.py [N=1000] [workdir=/tmp/cfg4] ["extra host flags", e.g. "-lanes 3"]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
from pybsgs import ecpy  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    wd = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cfg4"
    extra = sys.argv[3].split() if len(sys.argv) > 3 else []
    os.makedirs(wd, exist_ok=True)
    keys, st = [], 0xC0FFEE
    for _ in range(n):
        st, r = ecpy.splitmix64(st)
        keys.append((1 << 63) + (r >> 1))
    with open(os.path.join(wd, "pubs.txt"), "w") as f:
        for k in keys:
            f.write("%064x%064x\n" % ecpy.mul(k))
    exe = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")
    t0 = time.time()
    res = subprocess.run([exe, "-dir", wd, "-t", "256", "-b", "256", "-p", "256", "-w", "30", "-htsz", "28", "-infile", os.path.join(wd, "pubs.txt"),
                          "-pk", "8000000000000000", "-pke", "ffffffffffffffff"] + extra, capture_output=True, text=True)
    dt = time.time() - t0
    got = {}
    for l in open(os.path.join(wd, "win.txt"), "rb").read().decode().split("\r\n"):
        if l.startswith("KEY["):
            got[int(l[4:l.index("]")])] = int(l.split("0x")[1], 16)
    missing = [i + 1 for i in range(n) if got.get(i + 1) != keys[i]]
    job = [float(l.split()[2][:-2]) for l in res.stdout.splitlines() if l.startswith("Job time")]
    tiles = [int(l.split()[3]) for l in res.stdout.splitlines() if l.startswith("Job time")]
    print(json.dumps({"config": "-infile %d pubkeys, range 8000000000000000..ffffffffffffffff, -t 256 -b 256 -p 256 -w 30 -htsz 28%s, 1 GPU" % (n, "".join(" " + e for e in extra)),
                      "keys_correct": n - len(missing), "keys": n, "missing": [(i, "%x" % keys[i - 1]) for i in missing[:20]], "wall_s_total_incl_table_build_and_file_save": dt,
                      "search_s_sum": sum(job), "search_s_mean_per_key": sum(job) / max(len(job), 1), "search_s_max": max(job) if job else None,
                      "tiles_total": sum(tiles), "returncode": res.returncode,
                      "startup": [l for l in res.stdout.splitlines() if l.startswith("[startup]") or l.startswith("Short jobs")]}))


if __name__ == "__main__":
    main()
