"""pybsgs -- thin ctypes binding of libbsgs_hip.so (include/bsgs_hip.h).

Plumbing only: every call goes straight to the C-ABI; there is NO CPU fallback.  Importing works
without a GPU (so CPU-only checks can verify the exported symbols); opening a device without an
MI355X raises BsgsError.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("BSGS_LIB_PATH") or os.path.join(PKG_ROOT, "build", "libbsgs_hip.so")   # override: A/B of alternative builds

TABLE_AUTO, TABLE_CSR, TABLE_LINES64, TABLE_LINES128, TABLE_LINES64_LIST, TABLE_LINES128_LIST = 0, 1, 2, 3, 4, 5
ERR_OVERFLOW = -5
ERR_DEGENERATE = -6
FLAG_REFERENCE_QUIRKS = 1

# every symbol include/bsgs_hip.h declares (checked by tests/test_abi.py)
NATIVE_SYMBOLS = [
    "bsgs_last_error", "bsgs_version", "bsgs_build_info", "bsgs_dev_count", "bsgs_dev_open", "bsgs_dev_close", "bsgs_dev_name",
    "bsgs_dev_meminfo", "bsgs_dev_cu_count", "bsgs_upload_g2", "bsgs_upload_g2_device", "bsgs_generate_g2",
    "bsgs_download_g2", "bsgs_upload_htgpu", "bsgs_upload_htgpu_device", "bsgs_table_info", "bsgs_step", "bsgs_run",
    "bsgs_enqueue", "bsgs_collect", "bsgs_dev_stream", "bsgs_steps_per_tile", "bsgs_selftest_fe", "bsgs_selftest_xs",
    "bsgs_bench_random_read", "bsgs_bench_stream", "bsgs_bench_modmul", "bsgs_set_tiles_per_launch", "bsgs_launch_count", "bsgs_build_baby_tables", "bsgs_build_baby_tables_device", "bsgs_build_baby_table_ext", "bsgs_ext_overflow_capacity", "bsgs_build_baby_table_ext_device", "bsgs_install_table_ext_device", "bsgs_profile_phases",
    "bsgs_set_walk", "bsgs_enqueue_walk", "bsgs_run_walk", "bsgs_walk_centres", "bsgs_set_flags", "bsgs_quirk_count", "bsgs_broadcast_tables",
    "bsgs_tiles_per_launch", "bsgs_engine_geometry", "bsgs_run_digest", "bsgs_selftest_lo64", "bsgs_compat_stats", "bsgs_debug_buffers", "bsgs_alloc_stats", "bsgs_tune_placement", "bsgs_chain_placement", "bsgs_chain_grades", "bsgs_debug_grade_rule", "bsgs_debug_xcd_profile",
    "bsgs_table_checksum", "bsgs_sample_g2", "bsgs_alloc_table_ext_recv", "bsgs_debug_last_kernel", "bsgs_compat_stats_ex", "bsgs_debug_table_owner", "bsgs_prepare", "bsgs_debug_last_batching", "bsgs_debug_narrow_batching",
    "bsgs_table_census", "bsgs_table_lookup", "bsgs_broadcast_tables_ex", "bsgs_startup_ext_tables", "bsgs_build_baby_table_ext_slice", "bsgs_build_overflow_set", "bsgs_debug_fabric_selftest", "bsgs_share_tables",
]
# exported by the TEST build only (build/libbsgs_hip_test.so = the shipped objects + csrc/test_hooks.hip; include/bsgs_hip.h under BSGS_TEST_HOOKS)
TEST_HOOK_SYMBOLS = ["bsgs_debug_corrupt_table", "bsgs_debug_realloc"]
TEST_LIB_PATH = os.path.join(PKG_ROOT, "build", "libbsgs_hip_test.so")
COMPAT_SYMBOLS = [
    "cuInit", "cuDeviceGetCount", "cuDeviceGet", "cuDeviceGetName", "cuDeviceTotalMem_v2", "cuDeviceComputeCapability",
    "cuDeviceGetAttribute", "cuCtxCreate_v2", "cuCtxDestroy_v2", "cuCtxSynchronize", "cuMemGetInfo_v2", "cuModuleLoadData",
    "cuModuleGetFunction", "cuModuleGetGlobal_v2", "cuFuncSetCacheConfig", "cuFuncSetBlockShape", "cuParamSetSize",
    "cuParamSeti", "cuMemAlloc_v2", "cuMemFree_v2", "cuMemcpyHtoD_v2", "cuMemcpyDtoH_v2", "cuLaunchGrid",
    # declared by the reference's Import block but never called by v1.9.7 (exported so that the unchanged block links)
    "cuDeviceTotalMem", "cuCtxCreate", "cuCtxDestroy", "cuMemAlloc", "cuMemFree", "cuMemcpyHtoD", "cuMemcpyDtoH", "cuModuleGetGlobal",
    "cuModuleLoad", "cuParamSetv", "cuLaunchGridAsync", "cuLaunch", "cuFuncSetSharedSize", "cuFuncGetAttribute", "cuGetErrorName",
    "cuEventCreate", "cuEventDestroy", "cuEventQuery", "cuEventRecord", "cuEventSynchronize", "cuStreamCreate", "cuStreamCreate_v2",
    "cuStreamDestroy", "cuStreamSynchronize", "cuStreamQuery",
]


class BsgsError(RuntimeError):
    pass


class Hit(C.Structure):
    _fields_ = [("code", C.c_uint32), ("idx", C.c_uint32)]


class HitEx(C.Structure):
    _fields_ = [("code", C.c_uint32), ("idx", C.c_uint32), ("tile", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's) and asks for it by a name the
    system copy does not answer to: if libbsgs_hip.so pulled in /opt/rocm's first, a later `import torch` loads a SECOND runtime and finds
    "No HIP GPUs".  So when torch is installed but not imported yet, its copy is loaded first (by path, globally) and libbsgs_hip.so binds
    to it through the SONAME -- the arrangement every `import torch; import pybsgs` process has anyway.  BSGS_NO_TORCH_HIP_PRELOAD=1 skips it."""
    import sys
    if "torch" in sys.modules or os.environ.get("BSGS_NO_TORCH_HIP_PRELOAD") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load the HIP extension; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BsgsError("HIP extension missing: %s (run `make -C bsgs-cuda_amd` or __graft_entry__.build())" % LIB_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, u8p = C.c_void_p, C.c_char_p
        L.bsgs_last_error.restype = C.c_char_p
        L.bsgs_version.restype = C.c_char_p
        L.bsgs_build_info.restype = C.c_char_p
        sig = {
            "bsgs_dev_count": [C.POINTER(C.c_int)],
            "bsgs_dev_open": [C.c_int, C.POINTER(vp)],
            "bsgs_dev_close": [vp],
            "bsgs_dev_name": [vp, C.c_char_p, C.c_int],
            "bsgs_dev_meminfo": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
            "bsgs_dev_cu_count": [vp, C.POINTER(C.c_int)],
            "bsgs_upload_g2": [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32],
            "bsgs_upload_g2_device": [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32],
            "bsgs_generate_g2": [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32],
            "bsgs_download_g2": [vp, vp, C.c_size_t],
            "bsgs_upload_htgpu": [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32],
            "bsgs_upload_htgpu_device": [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32],
            "bsgs_table_info": [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
            "bsgs_step": [vp, u8p, u8p, C.POINTER(Hit), C.c_uint32, C.POINTER(C.c_uint32)],
            "bsgs_run": [vp, u8p, C.c_uint32, C.POINTER(HitEx), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)],
            "bsgs_enqueue": [vp, u8p, C.c_uint32],
            "bsgs_collect": [vp, C.POINTER(HitEx), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)],
            "bsgs_dev_stream": [vp, C.POINTER(vp)],
            "bsgs_steps_per_tile": [vp, C.POINTER(C.c_uint64)],
            "bsgs_selftest_fe": [vp, C.c_int, u8p, u8p, vp, C.c_uint32],
            "bsgs_selftest_xs": [vp, u8p, u8p, C.c_uint64, C.c_uint32, vp],
            "bsgs_bench_random_read": [vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)],
            "bsgs_bench_modmul": [vp, C.POINTER(C.c_double)],
            "bsgs_bench_stream": [vp, C.c_int, C.c_uint64, C.POINTER(C.c_double)],
            "bsgs_set_tiles_per_launch": [vp, C.c_uint32],
            "bsgs_launch_count": [vp, C.POINTER(C.c_uint64)],
            "bsgs_build_baby_tables": [vp, C.c_uint64, C.c_uint32, vp, vp, C.c_uint32],
            "bsgs_build_baby_tables_device": [vp, C.c_uint64, C.c_uint32, vp, vp],
            "bsgs_build_baby_table_ext": [vp, C.c_uint64, C.c_uint32, C.c_uint32],
            "bsgs_ext_overflow_capacity": [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)],
            "bsgs_build_baby_table_ext_device": [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
            "bsgs_install_table_ext_device": [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32],
            "bsgs_alloc_table_ext_recv": [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)],
            "bsgs_debug_last_kernel": [vp, C.c_char_p, C.c_int],
            "bsgs_table_checksum": [vp, C.POINTER(C.c_uint64)],
            "bsgs_table_census": [vp, C.POINTER(C.c_uint64)],
            "bsgs_broadcast_tables_ex": [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)],
            "bsgs_startup_ext_tables": [C.POINTER(vp), C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp],
            "bsgs_build_baby_table_ext_slice": [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
            "bsgs_build_overflow_set": [vp, vp, C.c_uint64, vp, C.c_uint64],
            "bsgs_debug_fabric_selftest": [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)],
            "bsgs_table_lookup": [vp, vp, C.c_uint64, vp],
            "bsgs_sample_g2": [vp, vp, C.c_uint32, vp],
            "bsgs_debug_corrupt_table": [vp, C.c_uint64, C.c_uint32],
            "bsgs_debug_table_owner": [vp, C.POINTER(C.c_int)],
            "bsgs_prepare": [vp],
            "bsgs_profile_phases": [vp, u8p, C.c_uint32, C.POINTER(C.c_float)],
            "bsgs_set_walk": [vp, u8p, u8p],
            "bsgs_enqueue_walk": [vp, C.c_uint64, C.c_uint32],
            "bsgs_run_walk": [vp, C.c_uint64, C.c_uint32, C.POINTER(HitEx), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)],
            "bsgs_walk_centres": [vp, C.c_uint64, C.c_uint32, vp],
            "bsgs_set_flags": [vp, C.c_uint32],
            "bsgs_quirk_count": [vp, C.POINTER(C.c_uint32)],
            "bsgs_broadcast_tables": [C.POINTER(vp), C.c_int],
            "bsgs_share_tables": [vp, vp],
            "bsgs_tiles_per_launch": [vp, C.POINTER(C.c_uint32)],
            "bsgs_engine_geometry": [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
            "bsgs_debug_last_batching": [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
            "bsgs_debug_narrow_batching": [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)],
            "bsgs_run_digest": [vp, u8p, C.c_uint32, vp, C.POINTER(HitEx), C.c_uint32, C.POINTER(C.c_uint32)],
            "bsgs_selftest_lo64": [vp, u8p, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)],
            "bsgs_debug_buffers": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double)],
            "bsgs_debug_realloc": [vp, C.c_int, C.c_uint64],
            "bsgs_debug_xcd_profile": [vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_float)],
        }
        for name, args in sig.items():
            fn = getattr(L, name, None)
            if fn is None:                       # an older build loaded through BSGS_LIB_PATH for an A/B run: the call site raises if it is ever needed
                continue
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = L
    return _lib


def _chk(rc, allow_overflow=False):
    if rc == 0 or (allow_overflow and rc == ERR_OVERFLOW):
        return rc
    raise BsgsError("bsgs error %d: %s" % (rc, lib().bsgs_last_error().decode()))


def le32(v):
    return int(v).to_bytes(32, "little")


def build_info():
    """the -D switches the loaded library was built with ("" = the shipped build; "WRONG-RESULTS:..." = a timing experiment)"""
    return lib().bsgs_build_info().decode()


def broadcast_tables(devices):
    """replicas of devices[0]'s giants and table on the other Device objects (device-to-device copies; the same GPU may appear twice)"""
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    _chk(lib().bsgs_broadcast_tables(arr, len(devices)))


def share_tables(owner, twin):
    """two engines on one GPU: `twin` probes `owner`'s table in place (borrowed) and gets its own copy of the giants"""
    _chk(lib().bsgs_share_tables(owner.h, twin.h))


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_PEER = 0, 1, 2
STARTUP_BROADCAST, STARTUP_LOCAL, STARTUP_ALLGATHER = 0, 1, 2


class StartupReport(C.Structure):
    _fields_ = [("alloc_s", C.c_double), ("build_s", C.c_double), ("transfer_s", C.c_double), ("set_s", C.c_double), ("install_s", C.c_double),
                ("prepare_s", C.c_double), ("total_s", C.c_double), ("bytes_received", C.c_uint64), ("strategy", C.c_uint32), ("transport", C.c_uint32)]


def startup_ext_tables(devices, w, htsz, layout, strategy, transport=TRANSPORT_AUTO):
    """extended table on every Device of one process by one of the three start-up strategies (include/bsgs_hip.h BSGS_STARTUP_*); returns one dict per engine"""
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    rep = (StartupReport * len(devices))()
    _chk(lib().bsgs_startup_ext_tables(arr, len(devices), w, htsz, layout, strategy, transport, C.cast(rep, C.c_void_p)))
    return [{k: getattr(r, k) for k, _ in StartupReport._fields_} for r in rep]


def broadcast_tables_ex(devices, transport=TRANSPORT_AUTO, what=3):
    """replicas of devices[0]'s giants (what & 1) and table (what & 2) on the other Devices; returns (transport used, seconds)"""
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    used, secs = C.c_uint32(), C.c_double()
    _chk(lib().bsgs_broadcast_tables_ex(arr, len(devices), transport, what, C.byref(used), C.byref(secs)))
    return used.value, secs.value


def fabric_selftest(devices, transport, nbytes):
    """(mismatching words after the broadcast, after the all-gather, transport used) of the transport test (bsgs_debug_fabric_selftest)"""
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    bad, used = (C.c_uint64 * 2)(), C.c_uint32()
    _chk(lib().bsgs_debug_fabric_selftest(arr, len(devices), transport, nbytes, bad, C.byref(used)))
    return int(bad[0]), int(bad[1]), used.value


def alloc_stats():
    """(bytes of big buffers obtained physically contiguous, bytes obtained as ordinary pages) by this process so far"""
    a, b = C.c_uint64(), C.c_uint64()
    lib().bsgs_alloc_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _chk(lib().bsgs_alloc_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def device_count():
    n = C.c_int(0)
    _chk(lib().bsgs_dev_count(C.byref(n)))
    return n.value


class Device:
    """One GPU, mirroring the reference's per-GPU driver thread `cuda()` (1_9_7File.pb:2095-2553)."""

    def __init__(self, device_id=0):
        self.h = C.c_void_p()
        _chk(lib().bsgs_dev_open(device_id, C.byref(self.h)))
        self.L = lib()

    def close(self):
        if self.h:
            self.L.bsgs_dev_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def name(self):
        buf = C.create_string_buffer(256)
        _chk(self.L.bsgs_dev_name(self.h, buf, 256))
        return buf.value.decode()

    def meminfo(self):
        f, t = C.c_uint64(), C.c_uint64()
        _chk(self.L.bsgs_dev_meminfo(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def upload_g2(self, image, t, b, p):
        buf = C.create_string_buffer(image, len(image)) if isinstance(image, (bytes, bytearray)) else image
        _chk(self.L.bsgs_upload_g2(self.h, C.cast(buf, C.c_void_p), t, b, p))

    def upload_g2_device(self, dptr, t, b, p):
        _chk(self.L.bsgs_upload_g2_device(self.h, C.c_void_p(dptr), t, b, p))

    def generate_g2(self, ax, ay, t, b, p):
        _chk(self.L.bsgs_generate_g2(self.h, le32(ax) + le32(ay), t, b, p))

    def download_g2(self, nbytes):
        buf = C.create_string_buffer(nbytes)
        _chk(self.L.bsgs_download_g2(self.h, C.cast(buf, C.c_void_p), nbytes))
        return buf.raw

    def upload_htgpu(self, image, ht_items, w, layout=TABLE_AUTO):
        buf = C.create_string_buffer(image, len(image)) if isinstance(image, (bytes, bytearray)) else image
        _chk(self.L.bsgs_upload_htgpu(self.h, C.cast(buf, C.c_void_p), ht_items, w, layout))

    def upload_htgpu_device(self, dptr, ht_items, w, layout=TABLE_AUTO):
        _chk(self.L.bsgs_upload_htgpu_device(self.h, C.c_void_p(dptr), ht_items, w, layout))

    def build_baby_tables(self, w, htsz, want_gpu=True, want_cpu=True, install_layout=0xFFFFFFFF):
        items = 1 << htsz
        g = C.create_string_buffer(4 * (items + 1) + 4 * w) if want_gpu else None
        c = C.create_string_buffer(4 * (items + 1) + 8 * w) if want_cpu else None
        _chk(self.L.bsgs_build_baby_tables(self.h, w, htsz, C.cast(g, C.c_void_p) if g else None,
                                           C.cast(c, C.c_void_p) if c else None, install_layout))
        return (g.raw if g else None), (c.raw if c else None)

    def build_baby_tables_device(self, w, htsz, htgpu_dptr, htcpu_dptr=None):
        _chk(self.L.bsgs_build_baby_tables_device(self.h, w, htsz, C.c_void_p(htgpu_dptr) if htgpu_dptr else None,
                                                  C.c_void_p(htcpu_dptr) if htcpu_dptr else None))

    def build_baby_table_ext(self, w, htsz, layout=TABLE_LINES64_LIST):
        """k*G, k = 1..w (w up to 2^36) straight into bucket lines + overflow list on the device (no CSR, no positions)"""
        _chk(self.L.bsgs_build_baby_table_ext(self.h, w, htsz, layout))

    def ext_overflow_capacity(self, w, htsz, layout=TABLE_LINES64_LIST):
        cap = C.c_uint64(0)
        _chk(self.L.bsgs_ext_overflow_capacity(w, htsz, layout, C.byref(cap)))
        return cap.value

    def build_baby_table_ext_device(self, w, htsz, layout, lines_dptr, ovf_dptr, ovf_cap):
        """build into caller-owned device buffers (source of an RCCL broadcast); returns (ovf_n, overflow_buckets)"""
        n, ob = C.c_uint64(0), C.c_uint64(0)
        _chk(self.L.bsgs_build_baby_table_ext_device(self.h, w, htsz, layout, C.c_void_p(lines_dptr), C.c_void_p(ovf_dptr), ovf_cap, C.byref(n), C.byref(ob)))
        return n.value, ob.value

    def build_baby_table_ext_slice(self, w, htsz, layout, lines_dptr, part, nparts, list_dptr, list_cap):
        """the lines of 1/nparts of the buckets, in place inside the full line buffer; returns (entries in the slice's overflow list, over-full lines of the slice)"""
        n, ob = C.c_uint64(0), C.c_uint64(0)
        _chk(self.L.bsgs_build_baby_table_ext_slice(self.h, w, htsz, layout, C.c_void_p(lines_dptr), part, nparts, C.c_void_p(list_dptr), list_cap, C.byref(n), C.byref(ob)))
        return n.value, ob.value

    def build_overflow_set(self, list_dptr, n, set_dptr, slots):
        _chk(self.L.bsgs_build_overflow_set(self.h, C.c_void_p(list_dptr), n, C.c_void_p(set_dptr), slots))

    def install_table_ext_device(self, lines_dptr, ovf_dptr, ovf_n, overflow_buckets, w, htsz, layout):
        _chk(self.L.bsgs_install_table_ext_device(self.h, C.c_void_p(lines_dptr), C.c_void_p(ovf_dptr), ovf_n, overflow_buckets, w, htsz, layout))

    def alloc_table_ext_recv(self, w, htsz, layout):
        """engine-owned receive buffers for a broadcast extended table: (lines_dptr, ovf_dptr, ovf_cap slots)"""
        lines, ovf, cap = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _chk(self.L.bsgs_alloc_table_ext_recv(self.h, w, htsz, layout, C.byref(lines), C.byref(ovf), C.byref(cap)))
        return lines.value, ovf.value, cap.value

    def prepare(self):
        _chk(self.L.bsgs_prepare(self.h))

    def table_checksum(self):
        """device-side 64-bit checksums of (bucket lines, overflow set, CSR image, giants): equal across byte-identical replicas"""
        s = (C.c_uint64 * 4)()
        _chk(self.L.bsgs_table_checksum(self.h, s))
        return [int(x) for x in s]

    def table_census(self):
        """one streaming pass over the installed table (the reference's checkHT / checkHTpack, 1_9_7File.pb:3599-3627, 3101-3134): what it holds, and whether
        entries in lines + keys in the overflow set - duplicates equals w"""
        c = (C.c_uint64 * 8)()
        _chk(self.L.bsgs_table_census(self.h, c))
        names = ("line_entries", "overfull_lines", "set_keys", "duplicates", "malformed_lines", "unsorted_lines", "w", "total")
        return dict(zip(names, (int(x) for x in c)))

    def table_lookup(self, keys64):
        """batched membership through the shipped probe: keys64 = iterable of 64-bit keys (low 64 bits of x) -> list of bool"""
        import array
        a = array.array("Q", keys64)
        n = len(a)
        if not n:
            return []
        out = (C.c_uint8 * n)()
        addr, _ = a.buffer_info()
        _chk(self.L.bsgs_table_lookup(self.h, C.c_void_p(addr), n, C.cast(out, C.c_void_p)))
        return [bool(x) for x in out]

    def _test_hook(self, name):
        fn = getattr(self.L, name, None)
        if fn is None:
            raise BsgsError("%s is a TEST hook: it lives in %s only (set BSGS_LIB_PATH to it before pybsgs loads its library); the shipped library does not export it" % (name, TEST_LIB_PATH))
        return fn

    def debug_corrupt_table(self, byte_offset, xor_mask=1):
        """TEST BUILD hook: flip bits of one byte of the installed table (what the verification must catch)"""
        _chk(self._test_hook("bsgs_debug_corrupt_table")(self.h, byte_offset, xor_mask))

    def sample_g2(self, indices):
        """giants by number as (x, y) integer pairs (what a host compares with (i + 1) * ADDPUBG before it searches: checkGiantArr 1_9_7File.pb:1524-1559)"""
        import array
        a = array.array("Q", indices)
        n = len(a)
        if not n:
            return []
        out = C.create_string_buffer(64 * n)
        addr, _ = a.buffer_info()
        _chk(self.L.bsgs_sample_g2(self.h, C.c_void_p(addr), n, C.cast(out, C.c_void_p)))
        raw = out.raw
        return [(int.from_bytes(raw[64 * i:64 * i + 32], "little"), int.from_bytes(raw[64 * i + 32:64 * i + 64], "little")) for i in range(n)]

    def table_owned(self):
        v = C.c_int()
        _chk(self.L.bsgs_debug_table_owner(self.h, C.byref(v)))
        return bool(v.value)

    def last_kernel(self):
        buf = C.create_string_buffer(128)
        _chk(self.L.bsgs_debug_last_kernel(self.h, buf, 128))
        return buf.value.decode()

    def last_batching(self):
        """(threads, giants per thread) of the most recent tile launch: engine_geometry() for launches that fill the GPU, more threads x shorter
        batches for small ones (bsgs_hip.hip pick_batching)"""
        a, b = C.c_uint32(), C.c_uint32()
        _chk(self.L.bsgs_debug_last_batching(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def table_info(self):
        lay, nb, ov = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _chk(self.L.bsgs_table_info(self.h, C.byref(lay), C.byref(nb), C.byref(ov)))
        return lay.value, nb.value, ov.value

    def step(self, px, py, max_hits=4096):
        hits = (Hit * max_hits)()
        n = C.c_uint32()
        _chk(self.L.bsgs_step(self.h, le32(px), le32(py), hits, max_hits, C.byref(n)), allow_overflow=True)
        return [(hits[i].code, hits[i].idx) for i in range(min(n.value, max_hits))], n.value

    def run(self, centres, max_hits=65536):
        """centres: list of (x, y) ints.  Returns (hits [(tile, code, idx)], total, kernel_ms)."""
        blob = b"".join(le32(x) + le32(y) for x, y in centres)
        return self.run_raw(blob, len(centres), max_hits)

    def run_raw(self, blob, ntiles, max_hits=65536):
        hits = (HitEx * max_hits)()
        n, ms = C.c_uint32(), C.c_float()
        _chk(self.L.bsgs_run(self.h, blob, ntiles, hits, max_hits, C.byref(n), C.byref(ms)), allow_overflow=True)
        return [(hits[i].tile, hits[i].code, hits[i].idx) for i in range(min(n.value, max_hits))], n.value, ms.value

    def enqueue_raw(self, blob, ntiles):
        _chk(self.L.bsgs_enqueue(self.h, blob, ntiles))

    def collect(self, max_hits=65536):
        hits = (HitEx * max_hits)()
        n, ms = C.c_uint32(), C.c_float()
        _chk(self.L.bsgs_collect(self.h, hits, max_hits, C.byref(n), C.byref(ms)), allow_overflow=True)
        return [(hits[i].tile, hits[i].code, hits[i].idx) for i in range(min(n.value, max_hits))], n.value, ms.value

    def set_tiles_per_launch(self, n):
        _chk(self.L.bsgs_set_tiles_per_launch(self.h, n))

    def tiles_per_launch(self):
        n = C.c_uint32()
        _chk(self.L.bsgs_tiles_per_launch(self.h, C.byref(n)))
        return n.value

    def engine_geometry(self):
        a, b = C.c_uint32(), C.c_uint32()
        _chk(self.L.bsgs_engine_geometry(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_flags(self, flags):
        _chk(self.L.bsgs_set_flags(self.h, flags))

    def quirk_count(self):
        """giants of the resident G2 that reference-quirk mode re-computes (their Gy trips the reference's NEGMODP)"""
        n = C.c_uint32()
        _chk(self.L.bsgs_quirk_count(self.h, C.byref(n)))
        return n.value

    # ---- device-side tile walk: centre of tile k = P0 + k*stride, derived on the GPU ----
    def set_walk(self, p0, stride):
        _chk(self.L.bsgs_set_walk(self.h, le32(p0[0]) + le32(p0[1]), le32(stride[0]) + le32(stride[1])))

    def enqueue_walk(self, first, ntiles):
        _chk(self.L.bsgs_enqueue_walk(self.h, first, ntiles))

    def run_walk(self, first, ntiles, max_hits=65536):
        hits = (HitEx * max_hits)()
        n, ms = C.c_uint32(), C.c_float()
        _chk(self.L.bsgs_run_walk(self.h, first, ntiles, hits, max_hits, C.byref(n), C.byref(ms)), allow_overflow=True)
        return [(hits[i].tile, hits[i].code, hits[i].idx) for i in range(min(n.value, max_hits))], n.value, ms.value

    def walk_centres(self, first, ntiles):
        out = C.create_string_buffer(64 * ntiles)
        _chk(self.L.bsgs_walk_centres(self.h, first, ntiles, C.cast(out, C.c_void_p)))
        r = out.raw
        return [(int.from_bytes(r[64 * k:64 * k + 32], "little"), int.from_bytes(r[64 * k + 32:64 * k + 64], "little")) for k in range(ntiles)]

    def run_digest(self, centres, max_hits=65536):
        """like run(); also returns per (tile, engine thread) the (xor, sum) of every 64-bit key probed"""
        import numpy as np
        ntiles = len(centres)
        threads, _ = self.engine_geometry()
        blob = b"".join(le32(x) + le32(y) for x, y in centres)
        dg = np.zeros((ntiles, threads, 2), dtype=np.uint64)
        hits = (HitEx * max_hits)()
        n = C.c_uint32()
        _chk(self.L.bsgs_run_digest(self.h, blob, ntiles, dg.ctypes.data_as(C.c_void_p), hits, max_hits, C.byref(n)), allow_overflow=True)
        return dg, [(hits[i].tile, hits[i].code, hits[i].idx) for i in range(min(n.value, max_hits))], n.value

    def launch_count(self):
        n = C.c_uint64()
        _chk(self.L.bsgs_launch_count(self.h, C.byref(n)))
        return n.value

    def steps_per_tile(self):
        s = C.c_uint64()
        _chk(self.L.bsgs_steps_per_tile(self.h, C.byref(s)))
        return s.value

    def stream(self):
        s = C.c_void_p()
        _chk(self.L.bsgs_dev_stream(self.h, C.byref(s)))
        return s.value

    def selftest_fe(self, op, a_list, b_list):
        n = len(a_list)
        a = b"".join(le32(v) for v in a_list)
        b = b"".join(le32(v) for v in b_list)
        out = C.create_string_buffer(32 * n)
        _chk(self.L.bsgs_selftest_fe(self.h, op, a, b, C.cast(out, C.c_void_p), n))
        return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(n)]

    def selftest_lo64(self, a_list, b_list, iters):
        """(mismatches, exact-path cases, cases) of the low-64 squaring path vs the full-width arithmetic"""
        n = len(a_list)
        out = (C.c_uint64 * 3)()
        _chk(self.L.bsgs_selftest_lo64(self.h, b"".join(le32(v) for v in a_list), b"".join(le32(v) for v in b_list), n, iters, out))
        return int(out[0]), int(out[1]), int(out[2])

    def selftest_xs(self, px, py, first, count):
        out = C.create_string_buffer(96 * count)
        _chk(self.L.bsgs_selftest_xs(self.h, le32(px), le32(py), first, count, C.cast(out, C.c_void_p)))
        r = []
        for k in range(count):
            v = [int.from_bytes(out.raw[96 * k + 32 * i:96 * k + 32 * i + 32], "little") for i in range(3)]
            r.append((v[0], v[1], v[2] & 1))
        return r

    def profile_phases(self, blob, ntiles):
        ms = (C.c_float * 3)()
        _chk(self.L.bsgs_profile_phases(self.h, blob, ntiles, ms))
        return [float(x) for x in ms]

    def bench_random_read(self, footprint_bytes, granule=64):
        g, r = C.c_double(), C.c_double()
        _chk(self.L.bsgs_bench_random_read(self.h, footprint_bytes, granule, C.byref(g), C.byref(r)))
        return g.value, r.value

    def bench_stream(self, kind, nbytes):
        """GB/s of one pass over nbytes in a streaming pattern of the tile kernel (0 coalesced loads, 1 the same by LDS-DMA, 2 non-temporal stores)"""
        g = C.c_double()
        _chk(self.L.bsgs_bench_stream(self.h, kind, nbytes, C.byref(g)))
        return g.value

    def debug_buffers(self, measure=True):
        a = (C.c_uint64 * 5)()
        g = C.c_double()
        _chk(self.L.bsgs_debug_buffers(self.h, a, C.byref(g) if measure else None))
        return [int(x) for x in a], g.value

    def xcd_profile(self, first, ntiles):
        """([(ms until XCD x finished its last block, blocks it ran)] x 8, launch ms)"""
        out = (C.c_uint64 * 16)()
        ms = C.c_float()
        _chk(self.L.bsgs_debug_xcd_profile(self.h, first, ntiles, out, C.byref(ms)))
        return [(out[2 * x] / 1e5, int(out[2 * x + 1])) for x in range(8)], ms.value

    def chain_placement(self):
        """how the chain scratch was placed (bsgs_chain_placement)"""
        info, grade = (C.c_uint32 * 5)(), (C.c_float * 2)()
        self.L.bsgs_chain_placement.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        _chk(self.L.bsgs_chain_placement(self.h, info, grade))
        g, n, sep = (C.c_float * 64)(), C.c_uint32(), C.c_uint32()
        self.L.bsgs_chain_grades.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _chk(self.L.bsgs_chain_grades(self.h, g, 64, C.byref(n), C.byref(sep)))
        return {"pieces": int(info[0]), "tiles_per_piece": int(info[1]), "graded": int(info[2]), "handed_back": int(info[3]),
                "from_reserved_group": bool(info[4]), "best_grade_G_per_s": round(grade[0], 2), "worst_kept_grade_G_per_s": round(grade[1], 2),
                "separation_seen": bool(sep.value) or bool(info[4]), "grades_kept_first": [round(g[k], 1) for k in range(min(n.value, 64))]}

    def tune_placement(self, candidates=3):
        """start-up tuning of where chain scratch and bucket lines lie (bsgs_tune_placement):
        {"chain_ms": [...], "lines_ms": [...], "kept": (i, j), "final_ms": ms}"""
        ms = (C.c_float * (2 * candidates))()
        chosen = (C.c_uint32 * 2)()
        fin = C.c_float()
        self.L.bsgs_tune_placement.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        _chk(self.L.bsgs_tune_placement(self.h, candidates, ms, chosen, C.byref(fin)))
        return {"chain_ms": [round(ms[k], 2) for k in range(candidates) if ms[k] > 0], "lines_ms": [round(ms[candidates + k], 2) for k in range(candidates) if ms[candidates + k] > 0],
                "kept": (int(chosen[0]), int(chosen[1])), "final_ms": round(fin.value, 2)}

    def debug_realloc(self, which, spacer_bytes=0):
        _chk(self._test_hook("bsgs_debug_realloc")(self.h, which, spacer_bytes))

    def bench_modmul(self):
        g = C.c_double()
        _chk(self.L.bsgs_bench_modmul(self.h, C.byref(g)))
        return g.value
