"""bench_support.py -- the measuring instruments of bench.py (the driver's entry point stays bench.py): power / clock sampling, the box-independent figures, the structural
verification of the table a rank is about to search, the torch.distributed launcher, the rocprofv3 child passes (PMC counters, kernel trace) and their arithmetic, the
measured puzzle-64 solve.  Nothing here touches the oracle: the CPU-baseline leg lives in bench.py."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH_PY = os.path.join(ROOT, "bench.py")

import torch  # noqa: E402

class PowerSampler:
    """socket power and shader clock of THIS GPU from its hwmon files, sampled every 50 ms while the timed region runs
    (the tile kernel is power-capped: DESIGN.md 6).  Silent no-op when the files are not there."""

    def __init__(self, device_index):
        self.dir, self.samples, self._stop, self._th = None, [], threading.Event(), None
        try:
            import glob
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for card in glob.glob("/sys/class/drm/card*/device"):
                if want in os.path.realpath(card).lower():
                    hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                    if hw:
                        self.dir = hw[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self._read("power1_input") / 1e6, self._read("freq1_input") / 1e6))
            except Exception:
                return
            self._stop.wait(0.05)

    def start(self):
        if self.dir:
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join()
        s = self.samples[len(self.samples) // 4:]               # drop the ramp at the start of the region
        if not s:
            return None
        out = {"socket_W_mean": sum(x[0] for x in s) / len(s), "sclk_MHz_mean": sum(x[1] for x in s) / len(s), "samples": len(s),
               "source": "hwmon power1_input / freq1_input of this GPU during the timed region"}
        try:
            out["power_cap_W"] = self._read("power1_cap") / 1e6
        except Exception:
            pass
        return out


def box_independent(per_rank, idle_default_W=245.0):
    """Two figures that do not move with the box the driver happened to get (the rate itself does, by 6 %: the shader clock the 1.4 kW cap leaves differs
    from box to box): giant steps per shader-clock GIGA-cycle -- summed over the ranks, each rank's rate over ITS clock -- and socket energy per giant step above idle.
    A regression of the kernel shows in both whatever the box; a slow box shows in neither."""
    rates = [(r["giant_steps_per_s"], r.get("sclk_MHz"), r.get("socket_W"), r.get("idle_W")) for r in per_rank]
    if not rates or any(c is None or not c for _, c, _, _ in rates):
        return {"value_per_GHz": None, "nJ_per_giant_step": None}
    per_ghz = sum(v / (c / 1000.0) for v, c, _, _ in rates)
    idle = [i if i else idle_default_W for _, _, _, i in rates]
    nj = [((w - i) / v * 1e9) if (w and v) else None for (v, _, w, _), i in zip(rates, idle)]
    # clock-normalised: time per giant step = a / f + b (a clock-bound and a memory-bound part, fitted to twelve default lines on boxes between 1.64 and 1.78 GHz:
    # profiles/r07_box_independent_figures.json, where the rate spreads 6.2 % and this figure 2.1 %); every rank's rate is brought to the reference clock
    a, b, ref, src = 0.02898, 0.007916, 1.74, "built-in constants"
    try:
        with open(os.path.join(ROOT, "profiles", "r07_box_independent_figures.json")) as f:
            mdl = json.load(f)["model"]
        a, b, ref, src = mdl["a_ns_GHz"], mdl["b_ns"], mdl["reference_GHz"], "profiles/r07_box_independent_figures.json"
    except Exception:
        pass
    normalised = sum(v * (a / (c / 1000.0) + b) / (a / ref + b) for v, c, _, _ in rates)
    return {"value_per_GHz": per_ghz, "value_per_GHz_unit": "giant steps per second and GHz of sampled shader clock (sum over ranks)",
            "value_clock_normalised": normalised,
            "value_clock_normalised_how": "every rank's rate x (a / sclk + b) / (a / %.2f GHz + b), a = %.5f ns GHz, b = %.6f ns (%s): what this kernel does at %.2f GHz; valid for the "
                                          "default workload on 64-byte lines" % (ref, a, b, src, ref),
            "nJ_per_giant_step": (sum(nj) / len(nj)) if all(x is not None for x in nj) else None,
            "nJ_per_giant_step_how": "(socket W during the timed region - idle W sampled before the first launch, %s) / giant steps per second, mean over ranks" % ["%.0f" % i for i in idle]}


def structural_verification(dev, ecpy, w, maxnonce, A, seed=0xB5650000):
    """What the reference does with every table it builds or loads before it searches (checkHT 1_9_7File.pb:3599-3627 called :3717, checkHTpackFile :3101-3134
    called :3731 / :4859, checkGiantArr :1524-1559 called :1941), on THIS rank's engine, before the timed region:
      census   bsgs_table_census: entries in lines + overflow set - bound copies == w, no malformed line, no unsorted line
      babies   1024 sampled k in [1, w] (32 runs of 32 consecutive k: the first, the last, 30 random): x(k*G) mod 2^64 found through the shipped probe
               (bsgs_table_lookup); 256 sampled k in (w, 2w] not found (32-bit hash collisions apart: at most 2)
      giants   1024 sampled giants (32 runs of 32): the device's giant i == (i + 1) * ADDPUBG
    Returns the record for the JSON line; raises on any failure (no rate is reported for a table that does not verify)."""
    t0 = time.time()
    st = seed ^ w

    def rnd(n):
        nonlocal st
        st, z = ecpy.splitmix64(st)
        return z % n

    def runs(starts, unit, first_multiple):
        ks, pts = [], []
        for s0 in starts:
            q = ecpy.mul(first_multiple(s0), unit)
            for j in range(32):
                ks.append(s0 + j)
                pts.append(q)
                q = ecpy.add(q, unit)
        return ks, pts
    span = max(w - 31, 1)
    k_in, p_in = runs([1, span] + [1 + rnd(span) for _ in range(30)], ecpy.G, lambda k: k)
    k_in, p_in = zip(*[(k, q) for k, q in zip(k_in, p_in) if 1 <= k <= w])
    k_out, p_out = runs([w + 1] + [w + 1 + rnd(w) for _ in range(7)], ecpy.G, lambda k: k)
    gspan = max(maxnonce - 31, 1)
    g_i, g_pts = runs([0, gspan - 1] + [rnd(gspan) for _ in range(30)], A, lambda i: i + 1)
    g_i, g_pts = zip(*[(i, q) for i, q in zip(g_i, g_pts) if i < maxnonce])
    t_samples = time.time() - t0
    c = dev.table_census()
    if c["total"] != w or c["malformed_lines"] or c["unsorted_lines"]:
        raise SystemExit("bench.py: table verification FAILED: census %r where w = %d" % (c, w))
    found = dev.table_lookup([q[0] & 0xFFFFFFFFFFFFFFFF for q in list(p_in) + list(p_out)])
    missing = [k for k, f in zip(k_in, found) if not f]
    extra = sum(found[len(k_in):])
    if missing or extra > 2:
        raise SystemExit("bench.py: table verification FAILED: %d of %d sampled k*G (k <= w) not found (first k = %s), %d of %d beyond w found" % (
            len(missing), len(k_in), missing[:1], extra, len(k_out)))
    got = dev.sample_g2(list(g_i))
    wrong = [i for i, a, b2 in zip(g_i, got, g_pts) if a != b2]
    if wrong:
        raise SystemExit("bench.py: giants verification FAILED: %d of %d sampled giants are not (i + 1) * ADDPUBG (first: %d)" % (len(wrong), len(g_i), wrong[0]))
    return {"census_total": c["total"], "w": w, "overfull_lines": c["overfull_lines"], "set_keys": c["set_keys"], "malformed_lines": 0, "unsorted_lines": 0,
            "sampled_kG_found": "%d/%d" % (len(k_in), len(k_in)), "sampled_beyond_w_found": "%d/%d" % (extra, len(k_out)), "sampled_giants_ok": "%d/%d" % (len(g_i), len(g_i)),
            "seconds": round(time.time() - t0, 3), "of_which_python_ec_samples_s": round(t_samples, 3),
            "how": "bsgs_table_census + bsgs_table_lookup (the shipped probe) + bsgs_sample_g2 on this rank's engine, before the timed region; "
                   "mirrors checkHT / checkHTpackFile / checkGiantArr (1_9_7File.pb:3599-3627, 3101-3134, 1524-1559)"}


def respawn_under_torchrun(n, same_device=False):
    """`python bench.py --gpus N` without a launcher: become N ranks (one process per GPU; --same-device: all on cuda:0) on 127.0.0.1"""
    import socket
    import subprocess
    dry = os.environ.get("BENCH_PRINT_SPAWN") == "1"          # CPU test hook: show the launcher line instead of running it
    have = torch.cuda.device_count()
    if have < (1 if same_device else n) and not dry:
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH_PY] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if dry:
        print(json.dumps(cmd))
        raise SystemExit(0)
    raise SystemExit(subprocess.call(cmd, env=env))


def load_pmc_profile(cfg):
    """the latest committed rocprofv3 PMC summary (profiles/r*_pmc_traffic.json) and whether it was taken on THIS configuration
    (kernel variant, geometry, table): only then do its per-step figures describe this run"""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        with open(path) as f:
            pm = json.load(f)
    except Exception:
        return None, None, False
    prof_cfg = pm.get("config") or {"w": 30.0, "htsz": 28, "t": 256, "b": 256, "p": 256, "layout": "lines64", "variant": "10"}   # round-1 files: the default workload
    same = all(str(prof_cfg.get(k)) == str(cfg.get(k)) for k in ("w", "htsz", "t", "b", "p", "layout", "variant"))
    return pm, os.path.basename(path), same


def measured_solve(timeout_s=600):
    """BASELINE.json's second metric, MEASURED: the C++ host (reference CLI) solves the puzzle-64 vector (1_9_7File.pb:200-203) at config-2
    flags -t 256 -b 256 -p 256 -w 26 -htsz 25; `job_time_s` is the host's own "Job time" (search only: tables and giants loaded before)."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "bsgs-cuda_amd", "build", "bsgs_mi355x")
    if not os.path.exists(exe):
        return {"value": None, "note": "host binary missing: %s" % exe}
    geo = ["-t", "256", "-b", "256", "-p", "256", "-w", "26", "-htsz", "25"]
    pub, key = "03100611c54dfef604163b8358f7b7fac13ce478e02cb224ae16d45526b25d9d4d", 0xf7051f27b09112d4
    tmp = tempfile.mkdtemp(prefix="bsgs_solve_")
    try:
        t0 = time.time()
        r = subprocess.run([exe, "-dir", tmp] + geo + ["-onlygen"], capture_output=True, text=True, timeout=timeout_s)
        gen_s = time.time() - t0
        if r.returncode:
            return {"value": None, "note": "onlygen failed: %s" % (r.stdout[-300:] + r.stderr[-300:])}
        t0 = time.time()
        r = subprocess.run([exe, "-dir", tmp] + geo + ["-pb", pub, "-pk", "8000000000000000", "-pke", "ffffffffffffffff"],
                           capture_output=True, text=True, timeout=timeout_s)
        wall = time.time() - t0
        if r.returncode:
            return {"value": None, "note": "solve failed: %s" % (r.stdout[-300:] + r.stderr[-300:])}
        with open(os.path.join(tmp, "win.txt"), "rb") as f:
            found = f.read().decode().split("\r\n")[0]
        ok = found == "KEY[1]: 0x" + "%064x" % key
        job = [ln for ln in r.stdout.splitlines() if ln.startswith("Job time")][0].split()
        job_s, tiles = float(job[2].rstrip("s,")), int(job[3])
        # the COLD path: an empty directory -> key, ONE command (tables and giants built on the GPU, the three files written as the reference does, then the search)
        cold = {"value": None}
        tmp2 = tempfile.mkdtemp(prefix="bsgs_cold_")
        try:
            t0 = time.time()
            rc = subprocess.run([exe, "-dir", tmp2] + geo + ["-pb", pub, "-pk", "8000000000000000", "-pke", "ffffffffffffffff"], capture_output=True, text=True, timeout=timeout_s)
            cold_wall = time.time() - t0
            with open(os.path.join(tmp2, "win.txt"), "rb") as f:
                ok2 = f.read().decode().split("\r\n")[0] == "KEY[1]: 0x" + "%064x" % key
            cjob = [ln for ln in rc.stdout.splitlines() if ln.startswith("Job time")][0].split()
            stages = [ln for ln in rc.stdout.splitlines() if ln.startswith("[startup]")]
            cold = {"value": cold_wall if ok2 else None, "unit": "s", "key_found": ok2, "job_time_s": float(cjob[2].rstrip("s,")), "startup_stages": stages,
                    "what": "process wall of ONE bsgs_mi355x command in an empty directory: GPU table build (2^26 points) + giants (2^24) + writing htGPU/htCPU/g2 files (2.1 GB) + upload + search"}
        except Exception as e:
            cold = {"value": None, "note": "failed: %r" % (e,)}
        finally:
            shutil.rmtree(tmp2, ignore_errors=True)
        # ... and the best this chip does for the same vector when the host picks the table itself (`-w auto`: Tune for the range, host_tune.cpp tune_plan): again an
        # empty directory -> key, ONE command
        best = {"value": None}
        tmp3 = tempfile.mkdtemp(prefix="bsgs_best_")
        try:
            t0 = time.time()
            rb = subprocess.run([exe, "-dir", tmp3, "-t", "256", "-b", "256", "-p", "256", "-w", "auto", "-pb", pub, "-pk", "8000000000000000", "-pke", "ffffffffffffffff"],
                                capture_output=True, text=True, timeout=timeout_s)
            best_wall = time.time() - t0
            with open(os.path.join(tmp3, "win.txt"), "rb") as f:
                ok3 = f.read().decode().split("\r\n")[0] == "KEY[1]: 0x" + "%064x" % key
            bjob = [ln for ln in rb.stdout.splitlines() if ln.startswith("Job time")][0].split()
            best = {"value": best_wall if ok3 else None, "unit": "s", "key_found": ok3, "job_time_s": float(bjob[2].rstrip("s,")),
                    "tune": [ln for ln in rb.stdout.splitlines() if ln.startswith("Tune for this range") or ln.startswith("-w auto")],
                    "startup_stages": [ln for ln in rb.stdout.splitlines() if ln.startswith("[startup]")],
                    "what": "process wall of ONE bsgs_mi355x -w auto command in an empty directory: Tune picks the table for the 2^63-key range (an extended table: no files, no htCPU), "
                            "the GPU builds it, the resolver's own multiples of G are computed on the host behind the start-up, then the search"}
        except Exception as e:
            best = {"value": None, "note": "failed: %r" % (e,)}
        finally:
            shutil.rmtree(tmp3, ignore_errors=True)
        return {"value": job_s if ok else None, "unit": "s", "key_found": ok, "job_time_s": job_s, "tiles": tiles, "giant_steps": tiles * 2 ** 25,
                "giant_steps_per_s": tiles * 2 ** 25 / job_s, "process_wall_s": wall, "onlygen_wall_s": gen_s, "cold": cold, "cold_best": best,
                "config": "bsgs_mi355x " + " ".join(geo) + " -pb <puzzle 64> -pk 8000000000000000 -pke ffffffffffffffff (1_9_7File.pb:200-203); "
                          "measured once after the timed regions, after this process released its own tables and scratch"}
    except Exception as e:
        return {"value": None, "note": "failed: %r" % (e,)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# the production instantiations of the tile kernel as rocprofv3 names them: <line size, PHASE_PROBE = false, QUAD = true | false>
_PROD_KERNEL = __import__("re").compile(r"giant_pair2_kernel<\d, false, (true|false)>")


CAL_BYTES = 1 << 34           # bytes each calibration kernel of a counter pass touches (bsgs_bench_random_read: 2^28 lines of 64 bytes; bsgs_bench_stream: one pass)


def _last(rows, n):
    return rows[-n:] if n and len(rows) > n else rows


def pmc_this_run(child_args, steps_per_launch, counted, parent_ms, trace_args=None, trace_counted=None, timeout_s=500):
    """HBM bytes and VALU figures of the tile kernel measured NOW, on this box: this script is re-run as a short child (same configuration, same launch
    size) under `rocprofv3 --pmc`, one counter group per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined with traces).
    A child first runs launches until eight in a row are within 1 % of the fastest it has seen (the parent handed tens of GiB back just before, and the
    driver wipes freed memory in bursts that slow the GPU for seconds) and only its LAST `counted` dispatches of the production kernel are used.
    FETCH_SIZE / WRITE_SIZE are KiB (x 1024).  Every pass also runs the four calibration kernels over 2^34 known bytes each -- random 64-byte lines
    (the probe pattern), coalesced 16-byte-per-lane loads, the same by LDS-DMA, non-temporal 16-byte stores -- so that what the counters report per
    byte of each pattern is measured in the same process (MI355X_MICROARCH.md, HBM: coalesced reads are tallied at 1/2)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return {"error": "rocprofv3 not on PATH"}
    res = {"how": "child runs of this script under rocprofv3 --pmc <group>, one group per pass; each child settles (eight launches in a row within 1 %% of its fastest) and the "
                  "means are over its last %d dispatches of the production kernel; FETCH_SIZE/WRITE_SIZE KiB x 1024; calibration kernels over 2^34 bytes each in every pass" % counted,
           "parent_ms_per_launch": parent_ms, "passes": {}}
    cal_kernels = {"mb_gups_kernel<4>": "random_64B_lines", "mb_stream_read_kernel": "coalesced_16B_loads", "mb_stream_read_lds_kernel": "coalesced_16B_lds_dma",
                   "mb_stream_write_nt_kernel": "nt_16B_stores"}
    tmp = tempfile.mkdtemp(prefix="bsgs_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")

    def child_json(stdout):
        out = None
        for ln in stdout.splitlines():
            if ln.startswith("{"):
                out = json.loads(ln)
        return out

    def child_info(child, t0):
        info = {"seconds": round(time.time() - t0, 1)}
        if child:
            pw = (child.get("alu") or {}).get("power") or {}
            info.update({"child_ms_per_launch_hip_events": child["roofline"]["avg_launch_ms"], "child_settle_launches": child.get("settle_launches"),
                         "child_sclk_MHz": pw.get("sclk_MHz_mean"), "child_socket_W": pw.get("socket_W_mean")})
        return info
    try:
        for grp in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "VALUBusy"]):
            d = os.path.join(tmp, grp[0])
            cmd = ["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--", sys.executable, BENCH_PY] + child_args
            t0 = time.time()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            except subprocess.TimeoutExpired:
                res["passes"][grp[0]] = {"error": "timeout"}
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode or not files:
                res["passes"][grp[0]] = {"error": "rc %d: %s" % (r.returncode, (r.stderr or "")[-200:])}
                continue
            agg = {}
            with open(files[0]) as f:
                rows = sorted(csv.DictReader(f), key=lambda row: int(row.get("Dispatch_Id", 0) or 0))
            for row in rows:
                agg.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            info = child_info(child_json(r.stdout), t0)
            for (kern, ctr), v in agg.items():
                if _PROD_KERNEL.search(kern):
                    v = _last(v, counted)
                    info[ctr] = sum(v) / len(v)
                    info["dispatches"] = len(v)
                    info["kernel"] = kern[:80]
                for sub, name in cal_kernels.items():
                    if sub in kern and ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                        info.setdefault("calibration", {})[name] = _last(v, 1)[0] * 1024 / float(CAL_BYTES)
            res["passes"][grp[0]] = info
        # one more child under the kernel trace alone: rocprofv3's own duration of the tile kernel on THIS box, per dispatch, next to the HIP-event
        # figure of the same child and to the parent's ms_per_step
        try:
            d = os.path.join(tmp, "trace")
            # (this child follows the PARENT's protocol -- the same warm-up launches, >= 2 s of them, then settling, then the same K timed launches -- so that its
            # last K dispatches are the parent's timed region again, on another allocation: a three-launch child is not power-settled and read 3-4 % high)
            counted = trace_counted or counted
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable, BENCH_PY] + (trace_args or child_args)
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            info = child_info(child_json(r.stdout), t0)
            tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
            st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
            if tr:
                with open(tr[0]) as fh:
                    rows = [row for row in csv.DictReader(fh) if _PROD_KERNEL.search(row["Kernel_Name"])]
                rows.sort(key=lambda row: int(row["Start_Timestamp"]))
                dur = [(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6 for row in rows]
                tail = _last(dur, counted)
                info.update({"kernel": rows[-1]["Kernel_Name"][:80] if rows else None, "calls": len(dur), "counted": len(tail),
                             "avg_ms": sum(tail) / len(tail) if tail else None, "min_ms": min(tail) if tail else None, "max_ms": max(tail) if tail else None,
                             "avg_ms_all_calls_including_settling": sum(dur) / len(dur) if dur else None,
                             "ratio_to_parent_ms_per_step": (sum(tail) / len(tail) / parent_ms) if (tail and parent_ms) else None,
                             "how": "rocprofv3 --kernel-trace around a child run of this script; the child's last %d dispatches of the production kernel (after settling), from the per-dispatch trace" % counted})
            elif st:
                with open(st[0]) as fh:
                    for row in csv.DictReader(fh):
                        if _PROD_KERNEL.search(row["Name"]):
                            info.update({"kernel": row["Name"][:80], "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6, "how": "kernel_stats.csv (all calls)"})
            else:
                info["error"] = "rc %d: %s" % (r.returncode, (r.stderr or "")[-200:])
            res["kernel_trace"] = info
        except Exception as e:
            res["kernel_trace"] = {"error": repr(e)}
        f, wr, va = res["passes"].get("FETCH_SIZE", {}), res["passes"].get("WRITE_SIZE", {}), res["passes"].get("SQ_INSTS_VALU", {})
        cal = dict(f.get("calibration") or {})
        cal.update({k: v for k, v in (wr.get("calibration") or {}).items() if k == "nt_16B_stores"})
        res["calibration_ratios"] = cal
        if "FETCH_SIZE" in f:
            res["fetch_bytes_per_launch"] = f["FETCH_SIZE"] * 1024
            res["fetch_bytes_per_step"] = f["FETCH_SIZE"] * 1024 / steps_per_launch
            res["calibration_ratio_random_64B"] = cal.get("random_64B_lines")
        if "WRITE_SIZE" in wr:
            res["write_bytes_per_launch"] = wr["WRITE_SIZE"] * 1024
            res["write_bytes_per_step"] = wr["WRITE_SIZE"] * 1024 / steps_per_launch
        if "fetch_bytes_per_launch" in res and "write_bytes_per_launch" in res:
            res["bytes_per_launch_uncorrected"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
            res["bytes_per_step_uncorrected"] = res["bytes_per_launch_uncorrected"] / steps_per_launch
        if "SQ_INSTS_VALU" in va:
            res["valu_instructions_per_step"] = va["SQ_INSTS_VALU"] * 64 / steps_per_launch
        if "VALUBusy" in va:
            res["valu_busy_percent"] = va["VALUBusy"]
            res["valu_busy_launch_ms"] = va.get("child_ms_per_launch_hip_events")
    except Exception as e:
        res["error"] = repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def load_fetch_breakdown():
    """the latest committed split of the tile kernel's FETCH_SIZE into its three streams (profiles/r*_fetch_breakdown.json: counter passes on the shipped
    library and on the two builds that drop one stream each -- no chain traffic, every giant read served from one cached KiB)"""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fetch_breakdown.json")))[-1]
        with open(path) as f:
            return json.load(f), os.path.basename(path)
    except Exception:
        return None, None


def corrected_traffic(m, steps_per_launch):
    """roofline.traffic with the counter correction applied: the raw FETCH_SIZE of this run is split into probe / stored products / giants by the committed
    breakdown's fractions, each share divided by what the counter reports per byte of ITS access pattern (this run's calibration kernels), WRITE_SIZE
    divided by the non-temporal store ratio."""
    bd, bd_name = load_fetch_breakdown()
    cal = m.get("calibration_ratios") or {}
    if not m.get("fetch_bytes_per_step") or not bd:
        return None
    raw = m["fetch_bytes_per_step"]
    fr = bd["fractions_of_raw_fetch"]
    r_rand = cal.get("random_64B_lines") or 1.0
    r_lds = cal.get("coalesced_16B_lds_dma") or cal.get("coalesced_16B_loads") or 1.0
    r_ld = cal.get("coalesced_16B_loads") or 1.0
    r_st = cal.get("nt_16B_stores") or 1.0
    probe, chain, giants = raw * fr["probe"] / r_rand, raw * fr["chain"] / r_lds, raw * fr["giants"] / r_ld
    write = (m.get("write_bytes_per_step") or 0.0) / r_st
    return {"fetch_breakdown_B_per_step": {"probe": probe, "chain": chain, "giants": giants}, "write_B_per_step": write,
            "bytes_per_step": probe + chain + giants + write, "bytes_per_launch": (probe + chain + giants + write) * steps_per_launch,
            "raw_fetch_B_per_step": raw, "raw_write_B_per_step": m.get("write_bytes_per_step"),
            "calibration_ratios_this_run": cal, "split_source": "profiles/%s (fractions of the raw counter: probe %.3f, chain %.3f, giants %.3f)" % (bd_name, fr["probe"], fr["chain"], fr["giants"]),
            "how": "raw FETCH_SIZE of this run x committed stream fractions, each share / this run's calibration ratio of its access pattern (probe: random 64-B lines; "
                   "chain: coalesced LDS-DMA; giants: coalesced loads); WRITE_SIZE / the non-temporal store ratio"}
