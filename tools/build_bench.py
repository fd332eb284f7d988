#!/usr/bin/env python3
"""tools/build_bench.py W HTSZ [ext] -- time the GPU baby-table builder alone: reference-format images (htGPU, as bench.py
builds them) or, with `ext` / W > 32, the extended table straight into bucket lines.  One JSON line; run under rocprofv3 --kernel-trace --stats
for the per-kernel split.  BSGS_BUILD_VERBOSE=1 makes the library print its own stage times on stderr."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
import torch  # noqa: E402
import pybsgs  # noqa: E402

wexp, htsz = float(sys.argv[1]), int(sys.argv[2])
ext = len(sys.argv) > 3 or wexp > 32
w = int(2 ** wexp)
reps = int(os.environ.get("REPS", "2"))
dev = pybsgs.Device(0)
torch.cuda.synchronize()
times = []
for r in range(reps):
    t0 = time.time()
    if ext:
        dev.build_baby_table_ext(w, htsz, 4 if w / (1 << htsz) <= 9 else 5)
    else:
        img = torch.empty((1 << htsz) + 1 + w, dtype=torch.int32, device="cuda:0")
        dev.build_baby_tables_device(w, htsz, img.data_ptr())
        torch.cuda.synchronize()
        t_img = time.time() - t0
        dev.upload_htgpu_device(img.data_ptr(), 1 << htsz, w, 0)
    torch.cuda.synchronize()
    times.append(time.time() - t0)
print(json.dumps({"w": w, "w_log2": wexp, "htsz": htsz, "extended": ext, "seconds": times, "points_per_s": w / min(times), "table": dev.table_info(),
                  "modmul_G_per_s": dev.bench_modmul()}))
dev.close()
