#!/usr/bin/env python3
"""Is every region of the 288 GB equally fast?  Allocate the memory in 4 GiB pieces and time, per piece: a streaming write, a streaming
read, a strided 16-byte-per-lane block stream like the chain scratch (2048 concurrent 4 MiB streams), and random 64-byte reads."""
import sys
import time
import torch

GiB = 1 << 30
piece = int(sys.argv[2]) * GiB if len(sys.argv) > 2 else 4 * GiB
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
bufs = []
for k in range(n):
    try:
        bufs.append(torch.empty(piece // 8, dtype=torch.int64, device=dev))
    except RuntimeError:
        break
print("pieces of 4 GiB:", len(bufs), flush=True)
idx = torch.randint(0, piece // 64, (1 << 26,), device=dev, dtype=torch.int64)


def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for k, x in enumerate(bufs):
    w = timed(lambda: x.fill_(k))
    r = timed(lambda: x.sum())
    v = x.view(-1, 8)                                 # rows of 64 bytes
    g = timed(lambda: v[idx, 0].sum())
    print("piece %2d @%012x  write %6.0f GB/s  read %6.0f GB/s  random 64-byte rows %5.1f G/s" % (k, x.data_ptr(), piece / w / 1e9, piece / r / 1e9, idx.numel() / g / 1e9), flush=True)
