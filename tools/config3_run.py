#!/usr/bin/env python3
"""BASELINE config 3 for real: one public key in an 80-bit range, -t 256 -b 256 -p 256 -w 34 -htsz 31 (2^34 baby points, 128 GiB of
bucket lines built in GPU memory), the key `frac` of the way into the range.  Prints one JSON line (measured time-to-solve).

  tools/config3_run.py [frac=0.25] [workdir=/tmp/cfg3] [table flags, default "-w 34 -htsz 31"; round 5: "-w 35 -buckets 1610612736", or "-w auto" = Tune's choice for the range]
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
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    wd = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cfg3"
    table = (sys.argv[3] if len(sys.argv) > 3 else "-w 34 -htsz 31").split()
    wlog = 35 if ("35" in table or "auto" in table) else 34
    os.makedirs(wd, exist_ok=True)
    start, end = 1 << 79, (1 << 80) - 1
    key = start + int(frac * (end - start)) + 0x9E3779B97F4A7C15
    exe = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")
    t0 = time.time()
    res = subprocess.run([exe, "-dir", wd, "-t", "256", "-b", "256", "-p", "256"] + table + ["-pb", "%064x%064x" % ecpy.mul(key),
                          "-pk", "%x" % start, "-pke", "%x" % end], capture_output=True, text=True)
    dt = time.time() - t0
    found = None
    try:
        for l in open(os.path.join(wd, "win.txt"), "rb").read().decode().split("\r\n"):
            if l.startswith("KEY["):
                found = int(l.split("0x")[1], 16)
    except OSError:
        pass
    import re
    wcount = 2**wlog
    for l in res.stdout.splitlines():                   # the count the host says it uses ("Items number set to 2^35.17=38654705664"): -w auto may pick one that is no power of two
        m = re.search(r"Items number set to [^=]*=\s*(\d+)", l)
        if m:
            wcount = int(m.group(1))
    job = [l for l in res.stdout.splitlines() if l.startswith("Job time")]
    rec = {"config": "single pubkey, 80-bit range 2^79..2^80-1, -t 256 -b 256 -p 256 %s, 1 GPU" % " ".join(table), "key_fraction_into_range": frac,
           "key": "%x" % key, "found": found == key, "process_wall_s_incl_table_build": dt, "returncode": res.returncode}
    if job:
        f = job[0].split()
        rec.update({"job_time_s": float(f[2].rstrip("s,")), "tiles": int(f[3]), "giant_steps": int(f[3]) * 2**25,
                    "giant_steps_per_s": int(f[3]) * 2**25 / float(f[2].rstrip("s,")), "keys_covered": int(f[3]) * 4 * 2**24 * wcount, "baby_points": wcount})
    rec["startup"] = [l for l in res.stdout.splitlines() if l.startswith("[startup]") or l.startswith("Tune for this range") or l.startswith("-w auto")]
    rec["verification"] = [l for l in res.stdout.splitlines() if l.startswith("Table verification") or l.startswith("Replica verification")]
    chk = [l for l in res.stdout.splitlines() if l.startswith("Checker:")]
    if chk:
        rec["checker"] = chk[0]
    if res.returncode:
        rec["stderr"] = res.stderr[-500:]
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
