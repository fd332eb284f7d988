#!/usr/bin/env python3
"""tools/startup_strategy_times.py [w_log2=34] [htsz=31] -- what one engine of an N-GPU start-up spends BUILDING under each strategy (one GPU is enough to measure it):
    local / broadcast   the whole table (bsgs_build_baby_table_ext_device)
    allgather           the slice an engine owns when N = 2, 4, 8 engines share the work (bsgs_build_baby_table_ext_slice: every point generated, 1/N filed)
Together with the link figure (153 GB/s per xGMI link) these give the expected start-up seconds of DESIGN.md 7.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))
import torch  # noqa: E402,F401
import pybsgs  # noqa: E402


def main():
    wl = int(sys.argv[1]) if len(sys.argv) > 1 else 34
    htsz = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    w = 1 << wl
    lay = pybsgs.TABLE_LINES64_LIST if htsz <= 31 else pybsgs.TABLE_LINES128_LIST
    buckets = htsz if htsz > 31 else 1 << htsz
    line = 64 if lay == pybsgs.TABLE_LINES64_LIST else 128
    dev = pybsgs.Device(0)
    t0 = time.time()
    lines, ovf, cap = dev.alloc_table_ext_recv(w, htsz, lay)
    out = {"w_log2": wl, "buckets": buckets, "line_bytes": line, "table_GiB": buckets * line / 2**30, "buffers_s": time.time() - t0, "slice_build_s": {}}
    lst = torch.empty(max(cap // 2, 1), dtype=torch.int64, device="cuda:0")
    for n in (8, 4, 2):
        if buckets % n:
            continue
        t0 = time.time()
        n_list, over = dev.build_baby_table_ext_slice(w, htsz, lay, lines, 0, n, lst.data_ptr(), lst.numel())
        out["slice_build_s"][str(n)] = {"seconds": time.time() - t0, "overflow_entries": n_list, "overfull_lines": over}
    del lst
    t0 = time.time()
    n_ovf, n_over = dev.build_baby_table_ext_device(w, htsz, lay, lines, ovf, cap)
    out["whole_build_s"] = time.time() - t0
    t0 = time.time()
    dev.install_table_ext_device(lines, ovf, n_ovf, n_over, w, htsz, lay)
    out["install_and_validate_s"] = time.time() - t0
    c = dev.table_census()
    out["census_total_equals_w"] = c["total"] == w
    link = 153e9
    tbl = buckets * line + out["slice_build_s"].get("8", {}).get("overflow_entries", 0) * 8 * 8
    out["expected_at_8_gpus_s"] = {
        "local": out["whole_build_s"],
        "broadcast": out["whole_build_s"] + tbl / link,
        "allgather_direct_links": out["slice_build_s"].get("8", {}).get("seconds", 0) + tbl / 8 / link,
        "allgather_ring": out["slice_build_s"].get("8", {}).get("seconds", 0) + tbl * 7 / 8 / link,
        "how": "build seconds measured here; transfers priced at one xGMI link of 153 GB/s per destination (broadcast: the table over one link per destination, all seven at once; all-gather: an "
               "engine receives 7/8 of the table -- over seven links at once if the collective uses them all, over one if it is a ring); installation (%.2f s, validation included) comes on top of each" % out["install_and_validate_s"]}
    dev.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
