#!/usr/bin/env python3
"""Does the physical placement of the engine's buffers change the launch time?  One process, the default workload: time launches, then
re-create one buffer at a time (bucket lines / chain scratch / giants), optionally behind a dummy allocation that shifts the placement,
and time again.  (Run-to-run the bench alternates between ~164 and ~177 ms per 192-tile launch on one box: profiles/r02d_repeat.log.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
import torch  # noqa: E402
os.environ.setdefault("BSGS_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bsgs-cuda_amd", "build", "libbsgs_hip_test.so"))   # debug_realloc: test library only
import pybsgs  # noqa: E402
from pybsgs import ecpy  # noqa: E402

t, b, p, wexp, htsz = 256, 256, 256, 30, 28
w, items = 1 << wexp, 1 << htsz
dev = pybsgs.Device(0)
img = torch.empty(items + 1 + w, dtype=torch.int32, device="cuda:0")
dev.build_baby_tables_device(w, htsz, img.data_ptr())
A = ecpy.addpubg(w)
_, D = ecpy.tile_stride(t, b, p, w)
p0 = ecpy.mul(0x5EED5EED)


def measure(tag, n=4):
    tpl = dev.tiles_per_launch()
    dev.set_walk(p0, D)
    dev.enqueue_walk(0, tpl)
    dev.collect()
    for k in range(n):
        dev.enqueue_walk((k + 1) * tpl, tpl)
    _, _, ms = dev.collect()
    addr, gbps = dev.debug_buffers()
    cp = dev.chain_placement()
    g = cp.get("grades_kept_first") or [0.0]
    print("%-44s %7.2f ms per %d-tile launch = %.2f G/s   lines@%x chain@%x g2@%x  lines random read %.0f GB/s  | scratch: separation %s, kept %.1f..%.1f, all %.1f..%.1f of %d graded" % (
        tag, ms / n, tpl, tpl * 2**25 / (ms / n * 1e-3) / 1e9, addr[0], addr[1], addr[2], gbps, cp.get("separation_seen"), cp.get("worst_kept_grade_G_per_s", 0), cp.get("best_grade_G_per_s", 0),
        min(g), max(g), len(g)), flush=True)


dev.upload_htgpu_device(img.data_ptr(), items, w, 0)
dev.generate_g2(A[0], A[1], t, b, p)
measure("initial")
measure("again")
if len(sys.argv) > 1 and sys.argv[1] == "tune":
    # does the placement bsgs_tune_placement picks keep its level afterwards?
    measure("as started"); measure("again")
    for rnd in range(2):
        t0 = time.time()
        r = dev.tune_placement(int(os.environ.get("PROBE_CANDIDATES", "3")))
        print("tune_placement:", r, "%.1f s" % (time.time() - t0), flush=True)
        for rep in range(5):
            measure("after tuning, measurement %d" % (rep + 1))
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "stable":
    # is the level of a chain-scratch placement stable right after the allocation churn that produced it?
    measure("as started")
    for k in range(6):
        dev.debug_realloc(1, (k % 3) << 30)
        for rep in range(int(os.environ.get("PROBE_REPS", "4"))):
            measure("chain placement %d, measurement %d" % (k + 1, rep + 1))
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "kick":
    # is the level an operating point of the power manager that a burst of extra load can move?
    import threading
    measure("as started"); measure("again")
    kicker = pybsgs.Device(0)
    for rnd in range(3):
        stop = threading.Event()

        def burn():
            while not stop.is_set():
                kicker.bench_modmul()                      # pure multiply-add load on another stream (~50 ms per call)
        th = threading.Thread(target=burn)
        th.start()
        measure("WITH a concurrent arithmetic load (round %d)" % (rnd + 1))
        stop.set(); th.join()
        measure("after the load"); measure("after the load, again")
    kicker.close()
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "each":
    measure("first"); measure("second"); measure("third")
    for which, name in ((1, "chain"), (0, "lines"), (2, "giants"), (1, "chain")):
        for k in range(8):
            dev.debug_realloc(which, (k % 3) << 30)
            measure("%s moved %d" % (name, k + 1))
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "stream":
    measure("first")
    for k in range(6):
        dev.debug_realloc(3)
        measure("new stream %d" % (k + 1))
    for k in range(4):
        dev.debug_realloc(4)
        measure("new hit buffer + centres %d" % (k + 1))
    for k in range(4):
        dev.debug_realloc(1); dev.debug_realloc(2); dev.debug_realloc(0)
        measure("chain + giants + lines moved %d" % (k + 1))
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "reroll":
    # close the engine, open a new one, load everything again: does the level change inside one process?
    measure("first engine")
    for k in range(int(os.environ.get("PROBE_REROLLS", "7"))):
        dev.close()
        dev = pybsgs.Device(0)
        dev.upload_htgpu_device(img.data_ptr(), items, w, 0)
        dev.generate_g2(A[0], A[1], t, b, p)
        t0 = time.time()
        dev.prepare()                                            # the graded allocation of the chain scratch
        print("   prepare (chain scratch placed by grade): %.2f s" % (time.time() - t0), flush=True)
        measure("engine %d (everything re-created)" % (k + 2))
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "xcd":
    measure("production launches")
    for k in range(4):
        prof, ms = dev.xcd_profile(1000 * (k + 1), dev.tiles_per_launch())
        ends = [e for e, _ in prof]
        print("launch %.2f ms; per XCD: last block ends at %s ms (spread %.1f ms), blocks %s" % (
            ms, " ".join("%.1f" % e for e in ends), max(ends) - min(ends), [n for _, n in prof]), flush=True)
    dev.close()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "idle":
    # does an idle gap move the operating point?
    for gap in (0.0, 0.05, 0.2, 0.5, 1.0, 2.0, 0.0, 0.5, 0.5, 0.5, 2.0, 2.0, 5.0, 0.0):
        time.sleep(gap)
        measure("after %.2f s idle" % gap)
    dev.close()
    sys.exit(0)
names = {0: "bucket lines (16 GiB, random access)", 1: "chain scratch (48 GiB, streamed)", 2: "giants (1 GiB, through L2)"}
for which in (1, 1, 1, 2, 2, 2, 0, 0, 0, 1, 2, 0):
    for spacer in (0, 3 << 30):
        dev.debug_realloc(which, spacer)
        measure("moved: %s%s" % (names[which][:14], " (+3 GiB spacer)" if spacer else ""))
dev.close()
