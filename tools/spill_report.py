#!/usr/bin/env python3
"""Register / scratch report of every kernel of the library, from the compiler's own remarks (-Rpass-analysis=kernel-resource-usage; cross-compilation, no GPU):
   tools/spill_report.py [out.json]   -> one row per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, occupancy.
tests/test_abi.py asserts the budget of the hot kernels from the same remarks."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bsgs-cuda_amd", "csrc")
TUS = ["tile_lines64", "tile_lines128", "tile_lines64_any", "baby_builder", "table_install", "bsgs_hip", "diagnostics", "placement", "startup"]
KEYS = {"vgprs": r"VGPRs", "agprs": r"AGPRs", "sgprs": r"TotalSGPRs", "vgpr_spill": r"VGPRs Spill", "sgpr_spill": r"SGPRs Spill", "scratch_bytes_per_lane": r"ScratchSize \[bytes/lane\]",
        "waves_per_simd": r"Occupancy \[waves/SIMD\]", "lds_bytes_per_block": r"LDS Size \[bytes/block\]"}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def report(tus=TUS, extra=()):
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for tu in tus:
            r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Rpass-analysis=kernel-resource-usage", *extra, "-c", "-o",
                                os.path.join(tmp, tu + ".o"), os.path.join(CSRC, tu + ".hip")], capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stderr[-2000:])
            for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
                row = {"tu": tu, "mangled": blk.split()[0]}
                for k, pat in KEYS.items():
                    m = re.search(r"remark: [^\n]*\s" + pat + r": (\d+)", blk)
                    row[k] = int(m.group(1)) if m else None
                rows.append(row)
    dm = demangle([r["mangled"] for r in rows])
    for r in rows:
        r["kernel"] = dm[r["mangled"]].replace("(TileArgs)", "").strip()
    return rows


if __name__ == "__main__":
    rows = report()
    bad = [r for r in rows if r["vgpr_spill"] or r["scratch_bytes_per_lane"]]
    for r in sorted(rows, key=lambda r: (r["tu"], r["kernel"])):
        flag = "  <-- spills" if r in bad else ""
        print("%-18s %-78s VGPR %3s AGPR %3s SGPR %3s  spilled V %3s S %3s  scratch %4s B  %s waves%s" % (r["tu"], r["kernel"][:78], r["vgprs"], r["agprs"], r["sgprs"], r["vgpr_spill"], r["sgpr_spill"],
              r["scratch_bytes_per_lane"], r["waves_per_simd"], flag))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump({"what": "compiler remarks (-Rpass-analysis=kernel-resource-usage), hipcc -O3 --offload-arch=gfx950, per kernel of libbsgs_hip.so", "kernels": rows}, f, indent=1)
