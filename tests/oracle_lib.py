"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")


class Fe(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]

    @classmethod
    def from_int(cls, v):
        f = cls()
        for i in range(4):
            f.l[i] = (v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
        return f

    def to_int(self):
        return sum(int(self.l[i]) << (64 * i) for i in range(4))


class Pt(C.Structure):
    _fields_ = [("x", Fe), ("y", Fe)]

    @classmethod
    def from_ints(cls, x, y):
        p = cls()
        p.x = Fe.from_int(x)
        p.y = Fe.from_int(y)
        return p

    def to_ints(self):
        return (self.x.to_int(), self.y.to_int())


class Hit(C.Structure):
    _fields_ = [("code", C.c_uint32), ("idx", C.c_uint32)]


class Job(C.Structure):
    _fields_ = [("t", C.c_uint32), ("b", C.c_uint32), ("p", C.c_uint32), ("w", C.c_uint64),
                ("htsz", C.c_uint32), ("maxnonce", C.c_uint64), ("addpubg", Pt), ("center_big", Fe),
                ("center", Pt), ("prkaddbig", Fe), ("pubaddbig", Pt), ("priv_big", Fe),
                ("pubkey_big", Pt), ("realpub", Pt), ("findpub", Pt), ("glob_key", Fe), ("glob_pub", Pt)]


def build():
    """(Re)build liboracle.so with gcc if it is missing or older than its sources."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("curve64_ref.c", "bsgs_ref.c", "cpu_fast.c", "curve64_ref.h", "bsgs_ref.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        pfe, ppt = C.POINTER(Fe), C.POINTER(Pt)
        u8p = C.c_void_p
        sig = {
            "o_sethex32": (C.c_int, [pfe, C.c_char_p]),
            "o_gethex32": (None, [C.c_char_p, pfe]),
            "o_addX64": (C.c_uint64, [pfe, pfe, pfe]),
            "o_subX64": (C.c_uint64, [pfe, pfe, pfe]),
            "o_andX64": (None, [pfe, pfe, pfe]),
            "o_addModX64": (None, [pfe, pfe, pfe, pfe]),
            "o_subModX64": (None, [pfe, pfe, pfe, pfe]),
            "o_mulModX64": (None, [pfe, pfe, pfe]),
            "o_squareModX64": (None, [pfe, pfe]),
            "o_modInvX64": (None, [pfe, pfe, pfe]),
            "o_DBLTX64": (None, [ppt, ppt]),
            "o_ADDPTX64": (None, [ppt, ppt, ppt]),
            "o_PTMULX64": (None, [ppt, ppt, pfe]),
            "o_YfromX64": (None, [pfe, pfe]),
            "o_fillarrayN": (None, [u8p, C.c_size_t, ppt]),
            "o_build_baby_tables": (C.c_int, [C.c_uint64, C.c_uint32, u8p, u8p]),
            "o_pack_tables_from_keys": (C.c_int, [u8p, C.c_uint64, C.c_uint32, u8p, u8p]),
            "o_ht_filename": (None, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int]),
            "o_g2_filename": (None, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]),
            "o_htgpu_probe": (C.c_int, [u8p, C.c_uint64, C.c_uint64]),
            "o_htcpu_lookup": (C.c_int, [u8p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.c_int]),
            "o_addpubg": (None, [ppt, C.c_uint64]),
            "o_build_g2": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, u8p, u8p]),
            "o_g2_unpack": (None, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]),
            "o_tile_ref": (C.c_uint64, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64,
                                        C.c_uint32, C.POINTER(Hit), C.c_uint64]),
            "o_tile_ref_slice": (C.c_uint64, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64,
                                              C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(Hit), C.c_uint64]),
            "o_tile_ref_slice_digest": (C.c_uint64, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64,
                                                     C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(Hit), C.c_uint64, u8p]),
            "o_tile_ref_slice_keys": (None, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, u8p]),
            "o_fast_unpack_g2": (None, [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, u8p]),
            "o_fast_tile_slice_mt": (C.c_int, [ppt, u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, u8p, C.c_uint64, C.c_int,
                                               C.POINTER(C.c_uint64)]),
            "o_fast_tile_slice_keys_mt": (C.c_int, [ppt, u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, u8p]),
            "o_fast_keys_of_scalars_mt": (C.c_int, [u8p, C.c_uint64, u8p, C.c_int]),
            "o_bench_port_mt": (C.c_int, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
            "o_bench_fast_mt": (C.c_int, [ppt, u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
            "o_tile_xs": (C.c_int, [ppt, ppt, C.c_uint32, pfe, pfe, pfe]),
            "o_bucket_ext": (C.c_uint32, [C.c_uint64, C.c_uint64]),
            "o_ext_probe": (C.c_int, [u8p, C.c_uint64, C.c_uint64, C.c_uint64]),
            "o_tile_ref_ext": (C.c_uint64, [ppt, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64, C.c_uint64, C.c_uint32,
                                            C.c_uint64, C.c_uint64, C.c_int, C.POINTER(Hit), C.c_uint64]),
            "o_job_init": (C.c_int, [C.POINTER(Job), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                     C.c_uint32, pfe, ppt, pfe]),
            "o_getjob": (None, [C.POINTER(Job), pfe, ppt]),
            "o_resolve_hit": (C.c_int, [C.POINTER(Job), u8p, C.c_uint64, C.c_uint32, C.c_uint32, pfe, ppt, pfe]),
            "o_parse_pubkey": (C.c_int, [ppt, C.c_char_p]),
            "o_compress_pub": (None, [C.c_char_p, ppt]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


# ---- convenience wrappers (ints in, ints out) ---------------------------------------------
P_INT = 2**256 - 2**32 - 977
N_INT = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX_INT = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY_INT = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def fe_op(name, *ints, mod=None):
    L = lib()
    r = Fe()
    args = [C.byref(Fe.from_int(v)) for v in ints]
    if mod is not None:
        args.append(C.byref(Fe.from_int(mod)))
    getattr(L, name)(C.byref(r), *args)
    return r.to_int()


def pt_mul(k, pt=(GX_INT, GY_INT)):
    r = Pt()
    lib().o_PTMULX64(C.byref(r), C.byref(Pt.from_ints(*pt)), C.byref(Fe.from_int(k)))
    return r.to_ints()


def pt_add(a, b):
    r = Pt()
    lib().o_ADDPTX64(C.byref(r), C.byref(Pt.from_ints(*a)), C.byref(Pt.from_ints(*b)))
    return r.to_ints()


def pt_neg(a):
    return (a[0], (-a[1]) % P_INT)


def build_baby_tables(w, htsz):
    items = 1 << htsz
    gpu = C.create_string_buffer(4 * (items + 1) + 4 * w)
    cpu = C.create_string_buffer(4 * (items + 1) + 8 * w)
    rc = lib().o_build_baby_tables(w, htsz, C.cast(gpu, C.c_void_p), C.cast(cpu, C.c_void_p))
    assert rc == 0
    return gpu.raw, cpu.raw


def pack_tables_from_keys(keys, htsz):
    import numpy as np
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    w = len(keys)
    items = 1 << htsz
    gpu = C.create_string_buffer(4 * (items + 1) + 4 * w)
    cpu = C.create_string_buffer(4 * (items + 1) + 8 * w)
    rc = lib().o_pack_tables_from_keys(keys.ctypes.data_as(C.c_void_p), w, htsz,
                                       C.cast(gpu, C.c_void_p), C.cast(cpu, C.c_void_p))
    assert rc == 0
    return gpu.raw, cpu.raw


def build_g2(t, b, p, w):
    maxnonce = t * b * p
    packed = C.create_string_buffer(64 * maxnonce)
    rc = lib().o_build_g2(t, b, p, w, C.cast(packed, C.c_void_p), None)
    assert rc == 0
    return packed.raw


def g2_unpack(packed, t, b, p, i):
    r = Pt()
    buf = C.create_string_buffer(packed, len(packed)) if isinstance(packed, bytes) else packed
    lib().o_g2_unpack(C.byref(r), C.cast(buf, C.c_void_p), t, b, p, i)
    return r.to_ints()


def tile_ref(P, g2, t, b, p, htgpu, htsz, flags=0, max_hits=4096):
    hits = (Hit * max_hits)()
    g2b = C.create_string_buffer(g2, len(g2)) if isinstance(g2, bytes) else g2
    htb = C.create_string_buffer(htgpu, len(htgpu)) if isinstance(htgpu, bytes) else htgpu
    n = lib().o_tile_ref(C.byref(Pt.from_ints(*P)), C.cast(g2b, C.c_void_p), t, b, p,
                         C.cast(htb, C.c_void_p), 1 << htsz, flags, hits, max_hits)
    return [(hits[i].code, hits[i].idx) for i in range(min(n, max_hits))], n


def tile_xs(P, Gpt, flags=0):
    xm, xp, xd = Fe(), Fe(), Fe()
    eq = lib().o_tile_xs(C.byref(Pt.from_ints(*P)), C.byref(Pt.from_ints(*Gpt)), flags,
                         C.byref(xm), C.byref(xp), C.byref(xd))
    return eq, xm.to_int(), xp.to_int(), xd.to_int()


def tile_slice_digest(P, g2buf, t, b, p, htbuf, htsz, tid0, tid1, flags=0, max_hits=4096):
    """threads [tid0, tid1) of a tile: (sorted hits [(code, idx)], total, digest uint64[(tid1-tid0), 2]); g2buf / htbuf are
    ctypes buffers or numpy arrays (htbuf may be None: digest only)"""
    import numpy as np
    hits = (Hit * max_hits)()
    dg = np.zeros((tid1 - tid0, 2), dtype=np.uint64)
    ptr = lambda x: None if x is None else (x.ctypes.data_as(C.c_void_p) if hasattr(x, "ctypes") else C.cast(x, C.c_void_p))  # noqa: E731
    n = lib().o_tile_ref_slice_digest(C.byref(Pt.from_ints(*P)), ptr(g2buf), t, b, p, ptr(htbuf), 1 << htsz, flags,
                                      tid0, tid1, hits, max_hits, dg.ctypes.data_as(C.c_void_p))
    return [(hits[i].code, hits[i].idx) for i in range(min(n, max_hits))], n, dg


def tile_slice_keys(P, g2buf, t, b, p, tid0, tid1, flags=0):
    """every 64-bit key the reference threads [tid0, tid1) of a tile probe: uint64[(tid1-tid0), p, 2] = (x(P-G), x(P+G) | x(2P)) per giant"""
    import numpy as np
    keys = np.zeros((tid1 - tid0, p, 2), dtype=np.uint64)
    ptr = g2buf.ctypes.data_as(C.c_void_p) if hasattr(g2buf, "ctypes") else C.cast(g2buf, C.c_void_p)
    lib().o_tile_ref_slice_keys(C.byref(Pt.from_ints(*P)), ptr, t, b, p, flags, tid0, tid1, keys.ctypes.data_as(C.c_void_p))
    return keys


def tile_ref_ext(P, g2, t, b, p, ck, buckets, flags=0, tid0=0, tid1=None, phase0=True, max_hits=65536):
    """the tile model over a table with ANY number of buckets, given as its ascending composite keys (bucket << 32 | hash), numpy uint64:
    (sorted hits [(code, idx)], total).  Whole tile incl. the probe of P itself by default; a thread slice with tid0 / tid1 / phase0=False."""
    import numpy as np
    ck = np.ascontiguousarray(ck, dtype=np.uint64)
    hits = (Hit * max_hits)()
    g2b = C.create_string_buffer(g2, len(g2)) if isinstance(g2, bytes) else g2
    ptr = g2b.ctypes.data_as(C.c_void_p) if hasattr(g2b, "ctypes") else C.cast(g2b, C.c_void_p)
    n = lib().o_tile_ref_ext(C.byref(Pt.from_ints(*P)), ptr, t, b, p, ck.ctypes.data_as(C.c_void_p), len(ck), buckets, flags,
                             tid0, t * b if tid1 is None else tid1, 1 if phase0 else 0, hits, max_hits)
    return [(hits[i].code, hits[i].idx) for i in range(min(n, max_hits))], n


def ext_probe(ck, buckets, key64):
    import numpy as np
    ck = np.ascontiguousarray(ck, dtype=np.uint64)
    return lib().o_ext_probe(ck.ctypes.data_as(C.c_void_p), len(ck), buckets, key64)


def fast_tile_slice_keys(P, g2buf, t, b, p, tid0, tid1, nthreads=1):
    """every probed 64-bit key of the reference threads [tid0, tid1) by the fast CPU implementation: uint64[(tid1-tid0), p, 2]"""
    import numpy as np
    ptr = lambda x: x.ctypes.data_as(C.c_void_p) if hasattr(x, "ctypes") else C.cast(x, C.c_void_p)  # noqa: E731
    count = (tid1 - tid0) * p
    plain = np.empty(8 * count, dtype=np.uint64)
    lib().o_fast_unpack_g2(ptr(g2buf), t, b, p, tid0 * p, count, plain.ctypes.data_as(C.c_void_p))
    keys = np.zeros((tid1 - tid0, p, 2), dtype=np.uint64)
    lib().o_fast_tile_slice_keys_mt(C.byref(Pt.from_ints(*P)), plain.ctypes.data_as(C.c_void_p), tid0 * p, p, tid0, tid1, nthreads,
                                    keys.ctypes.data_as(C.c_void_p))
    return keys


def fast_tile_slice(P, g2buf, t, b, p, htbuf, htsz, tid0, tid1, nthreads=1):
    """the best-effort CPU baseline (oracle/cpu_fast.c) over threads [tid0, tid1): (hits, digest_xor, digest_sum, seconds of
    the tile work only -- the giants are unpacked before the clock starts)"""
    import time
    import numpy as np
    ptr = lambda x: None if x is None else (x.ctypes.data_as(C.c_void_p) if hasattr(x, "ctypes") else C.cast(x, C.c_void_p))  # noqa: E731
    count = (tid1 - tid0) * p
    plain = np.empty(8 * count, dtype=np.uint64)
    lib().o_fast_unpack_g2(ptr(g2buf), t, b, p, tid0 * p, count, plain.ctypes.data_as(C.c_void_p))
    out = (C.c_uint64 * 3)()
    t0 = time.time()
    lib().o_fast_tile_slice_mt(C.byref(Pt.from_ints(*P)), plain.ctypes.data_as(C.c_void_p), tid0 * p, p, tid0, tid1, ptr(htbuf),
                               1 << htsz, nthreads, out)
    return int(out[0]), int(out[1]), int(out[2]), time.time() - t0


def fast_keys_of_scalars(scalars, nthreads=None):
    """low 64 bits of x(k*G) for every 64-bit scalar k (numpy uint64 array out): the expected keys of a sampled-membership check of a GPU-built
    baby table (the reference's checkHT / checkHTpack, 1_9_7File.pb:3599-3627, 3101-3134); oracle/cpu_fast.c on host threads"""
    import os
    import numpy as np
    k = np.ascontiguousarray(np.asarray(scalars, dtype=np.uint64))
    out = np.zeros(len(k), dtype=np.uint64)
    if len(k):
        rc = lib().o_fast_keys_of_scalars_mt(k.ctypes.data_as(C.c_void_p), len(k), out.ctypes.data_as(C.c_void_p), nthreads or min(64, os.cpu_count() or 1))
        assert rc == 0
    return out
