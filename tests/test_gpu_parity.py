"""GPU parity tests (pytest -m gpu): the HIP path, called through the C-ABI, against the CPU oracle
and the committed golden fixtures.  Integer work: bit-exact everywhere."""
import ctypes as C
import random
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 2**256 - 2**32 - 977


@pytest.fixture(scope="module")
def dev():
    import pybsgs
    d = pybsgs.Device(0)          # raises loudly when the HIP extension or the GPU is missing
    yield d
    d.close()


@pytest.fixture(scope="module")
def O():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


# ------------------------------------------------------------------------------------------ field
def test_fe_ops_match_python_ints(dev):
    rnd = random.Random(20260928)
    edge = [0, 1, 2, P - 1, P - 2, 2**255, 2**256 - 2**32 - 978, 0xFFFFFFFF, 2**64 - 1, 2**128 - 1, 2**192 + 5,
            0x1000003D1, P - 0x1000003D1, (P + 1) // 2]
    a = edge + [rnd.randrange(P) for _ in range(3000)]
    b = list(reversed(edge)) + [rnd.randrange(P) for _ in range(3000)]
    assert dev.selftest_fe(0, a, b) == [x * y % P for x, y in zip(a, b)]
    assert dev.selftest_fe(1, a, b) == [x * x % P for x in a]
    assert dev.selftest_fe(2, a, b) == [(x + y) % P for x, y in zip(a, b)]
    assert dev.selftest_fe(3, a, b) == [(x - y) % P for x, y in zip(a, b)]
    nz = [x if x else 7 for x in a[:512]]
    assert dev.selftest_fe(4, nz, nz) == [pow(x, -1, P) for x in nz]


def test_fe_mul_carry_patterns(dev):
    """operands that maximise column sums / carries in the Comba product and both folds"""
    pats = [2**256 - 1, 2**256 - 2**32, int("ffffffff00000000" * 4, 16), int("00000000ffffffff" * 4, 16),
            int("ffffffff" * 8, 16) - 977, P - 1, P, P + 1, 2**256 - 0x1000003D1 - 1]
    a = [x for x in pats for _ in pats]
    b = [y for _ in pats for y in pats]
    got = dev.selftest_fe(0, a, b)
    assert got == [x * y % P for x, y in zip(a, b)]


def test_fe_matches_oracle(dev, O):
    rnd = random.Random(7)
    a = [rnd.randrange(P) for _ in range(500)]
    b = [rnd.randrange(P) for _ in range(500)]
    assert dev.selftest_fe(0, a, b) == [O.fe_op("o_mulModX64", x, y) for x, y in zip(a, b)]
    assert dev.selftest_fe(3, a, b) == [O.fe_op("o_subModX64", x, y, mod=P) for x, y in zip(a, b)]


# ------------------------------------------------------------------------------------------ giants
def test_g2_roundtrip_and_generator(dev, O, small_fx):
    fx = small_fx
    img = bytes.fromhex(fx["g2"])
    dev.upload_g2(img, fx["t"], fx["b"], fx["p"])
    assert dev.download_g2(len(img)) == img
    A = tuple(int(v, 16) for v in fx["addpubg"])
    dev.generate_g2(A[0], A[1], fx["t"], fx["b"], fx["p"])
    assert dev.download_g2(len(img)) == img
    # a geometry with thousands of giants, against the oracle's CPU builder (reference giant(), 197:1418-1488)
    t, b, p, w = 64, 3, 20, 1 << 14
    ref = O.build_g2(t, b, p, w)
    Apt = O.Pt()
    O.lib().o_addpubg(C.byref(Apt), w)
    ax, ay = Apt.to_ints()
    dev.generate_g2(ax, ay, t, b, p)
    assert dev.download_g2(len(ref)) == ref
    dev.upload_g2(ref, t, b, p)
    assert dev.download_g2(len(ref)) == ref


def test_tile_x_coordinates_match_oracle(dev, O, small_fx):
    fx = small_fx
    img = bytes.fromhex(fx["g2"])
    t, b, p = fx["t"], fx["b"], fx["p"]
    dev.upload_g2(img, t, b, p)
    n = t * b * p
    giants = [O.g2_unpack(img, t, b, p, i) for i in range(n)]
    for Pt in [O.pt_mul(0xC0FFEE), O.pt_mul(2**200 + 12345), giants[5], O.pt_neg(giants[9])]:
        got = dev.selftest_xs(Pt[0], Pt[1], 0, n)
        for i in range(n):
            eq, xm, xp, xd = O.tile_xs(Pt, giants[i], 0)
            assert got[i] == (xm, xd if eq else xp, eq), (i, eq)


def test_gpu_baby_table_builder(dev, O, small_fx):
    """GPU builder (replaces GenBabys..packHTGPUFile, 197:1237-1328, 2555-2895, 3232-3444): byte-exact file images"""
    import hashlib
    import json
    import os
    fx = small_fx
    gpu, cpu = dev.build_baby_tables(fx["w"], fx["htsz"])
    assert gpu.hex() == fx["htgpu"] and cpu.hex() == fx["htcpu"]
    # a size that is not a multiple of anything convenient, against the oracle's CPU builder
    for w, htsz in ((70001, 13), (1 << 17, 16)):
        rg, rc = O.build_baby_tables(w, htsz)
        g, c = dev.build_baby_tables(w, htsz)
        assert g == rg and c == rc, (w, htsz)
    # BASELINE config 1 (-w 20 -htsz 18 onlygen): sha256 of the files from the plain-Python generator
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_digests.json")) as f:
        d = json.load(f)
    g, c = dev.build_baby_tables(d["w"], d["htsz"], install_layout=2)
    assert hashlib.sha256(g).hexdigest() == d["htgpu_sha256"] and hashlib.sha256(c).hexdigest() == d["htcpu_sha256"]
    assert dev.table_info()[0] == 2
    # the installed table answers probes: k*G is found as code 5 for k <= w and not for k = w + 1
    dev.upload_g2(bytes.fromhex(fx["g2"]), fx["t"], fx["b"], fx["p"])
    for k, expect in ((1, True), (d["w"], True), (d["w"] + 1, False), (12345, True)):
        P = O.pt_mul(k)
        hits, _ = dev.step(P[0], P[1])
        assert ((5, 0xFFFFFFFF) in hits) == expect, k


def test_direct_line_builder_matches_oracle_tables(dev, O, small_fx):
    """bsgs_build_baby_table_ext (scatter into bucket lines + overflow list, the w >= 2^32 path) at small sizes: the
    fixture's hit lists, and the oracle's tiles over the oracle's own table for loads of 4, 64 and 680 per bucket."""
    fx = small_fx
    dev.upload_g2(bytes.fromhex(fx["g2"]), fx["t"], fx["b"], fx["p"])
    for layout in (4, 5):
        dev.build_baby_table_ext(fx["w"], fx["htsz"], layout)
        assert dev.table_info()[0] == layout
        for tl in fx["tiles"] + fx["known_key"]["walk"]:
            hits, n = dev.step(int(tl["px"], 16), int(tl["py"], 16))
            assert [list(h) for h in hits] == tl["hits"], (layout, tl.get("kind", "walk"))
    t, b, p = 64, 3, 20
    rnd = random.Random(31)
    # 4 per bucket (no overflow), 16 (half the buckets hold exactly 14 / 15 / 16 entries: the edges of the line capacity and of the bound
    # word of over-full lines), 64 and 680 (every line over-full: most probes are settled by the bound, the rest by the overflow set)
    for w, htsz in ((1 << 14, 12), (1 << 14, 10), (1 << 14, 8), (70001, 7)):
        g2 = O.build_g2(t, b, p, w)
        gpu, _ = O.build_baby_tables(w, htsz)
        dev.upload_g2(g2, t, b, p)
        n = t * b * p
        # centres that do hit: m*G with m = +-(i+1)*2w + b'  and the table's own points
        ms = [rnd.randrange(1, n) * 2 * w + rnd.randrange(1, w) for _ in range(3)] + [-(rnd.randrange(1, n) * 2 * w) - 5, w, 1]
        N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
        centres = [O.pt_mul(m % N) for m in ms]
        for layout in (4, 5):
            dev.build_baby_table_ext(w, htsz, layout)
            lay, _, ovf = dev.table_info()
            load = w / (1 << htsz)
            assert lay == layout
            if not (layout == 5 and load == 16):                      # (16 per bucket in 31-entry lines: an over-full bucket is a coin toss)
                assert (ovf > 0) == (load > (40 if layout == 5 else 8))
            for Pt in centres:
                ref, nref = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
                hits, nh = dev.step(Pt[0], Pt[1], 65536)
                assert (nh, hits) == (nref, ref) and nref > 0, (w, htsz, layout)


# ------------------------------------------------------------------------------------------ tiles
LAYOUTS = [1, 2, 3, 4, 5]      # CSR, 64-byte lines, 128-byte lines, the same two with an overflow list instead of the CSR


@pytest.mark.parametrize("layout", LAYOUTS)
def test_small_fixture_tiles(dev, small_fx, layout):
    """hit lists of the committed plain-Python fixture (codes 1, 2, 5; x-equal tiles; empty tiles)"""
    fx = small_fx
    dev.upload_g2(bytes.fromhex(fx["g2"]), fx["t"], fx["b"], fx["p"])
    dev.upload_htgpu(bytes.fromhex(fx["htgpu"]), 1 << fx["htsz"], fx["w"], layout)
    assert dev.table_info()[0] == layout
    for tl in fx["tiles"] + fx["known_key"]["walk"]:
        hits, n = dev.step(int(tl["px"], 16), int(tl["py"], 16))
        assert n == len(hits)
        assert [list(h) for h in hits] == tl["hits"], tl.get("kind", "walk")


def _planted_case(O, seed, t, b, p, w, htsz, nplant, tiles):
    """real giants + a table of random keys into which the 64-bit keys of x(P +- G2[i]) are planted."""
    rnd = random.Random(seed)
    g2 = O.build_g2(t, b, p, w)
    n = t * b * p
    centres = [O.pt_mul(rnd.randrange(1, 2**128)) for _ in range(tiles)]
    keys = [rnd.getrandbits(64) for _ in range(w - nplant * tiles)]
    for Pt in centres:
        for _ in range(nplant):
            i = rnd.randrange(n)
            _, xm, xp, _ = O.tile_xs(Pt, O.g2_unpack(g2, t, b, p, i), 0)
            keys.append((xm if rnd.random() < 0.5 else xp) & (2**64 - 1))
    keys[0] = centres[0][0] & (2**64 - 1)              # code 5 on the first tile
    # duplicates of a (bucket,hash) pair, as real tables may contain (197:2797-2805)
    keys[1] = keys[2]
    gpu, _ = O.pack_tables_from_keys(np.array(keys, dtype=np.uint64), htsz)
    return g2, gpu, centres


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("htsz", [14, 12, 10])          # mean bucket load 4, 16, 64: lines / partial / all-overflow
def test_planted_tiles_match_oracle(dev, O, layout, htsz):
    t, b, p, w = 64, 5, 12, 1 << 16                      # T = 320 (tail wave: 64 live lanes of the second block)
    g2, gpu, centres = _planted_case(O, 1000 + htsz, t, b, p, w, htsz, 24, 4)
    dev.upload_g2(g2, t, b, p)
    dev.upload_htgpu(gpu, 1 << htsz, w, layout)
    lay, _, ovf = dev.table_info()
    assert lay == layout
    if layout in (2, 4) and htsz == 10:
        assert ovf > 1000                                # nearly every bucket overflows a 64-byte line
    total = 0
    for k, Pt in enumerate(centres):
        ref, nref = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
        hits, n = dev.step(Pt[0], Pt[1], 65536)
        assert n == nref and hits == ref, (k, layout, htsz)
        total += n
    assert total >= 24 * 4                               # planted hits found (+ deterministic false positives)
    # the same tiles queued back-to-back with one synchronisation
    hits, n, ms = dev.run(centres, 65536)
    ref_all = []
    for k, Pt in enumerate(centres):
        r, _ = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
        ref_all += [(k, c, i) for c, i in r]
    assert hits == ref_all and n == len(ref_all) and ms > 0


def test_false_positives_match_oracle(dev, O):
    """a table with 2^20 entries per bucket: random tiles hit only by 32-bit hash collision
    (about 2^-12 per probe); the GPU must report exactly the oracle's deterministic false positives.
    (Buckets this large live on the exact CSR path in every layout.)"""
    t, b, p, w, htsz = 64, 8, 32, 1 << 22, 2
    rnd = random.Random(99)
    g2 = O.build_g2(t, b, p, w)
    keys = np.frombuffer(np.random.default_rng(5).bytes(8 * w), dtype=np.uint64)
    gpu, _ = O.pack_tables_from_keys(keys, htsz)
    dev.upload_g2(g2, t, b, p)
    seen = 0
    for layout in (1, 2, 4):
        dev.upload_htgpu(gpu, 1 << htsz, w, layout)
        for s in range(3):
            Pt = O.pt_mul(rnd.randrange(1, 2**200))
            ref, nref = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
            hits, n = dev.step(Pt[0], Pt[1], 65536)
            assert (n, hits) == (nref, ref)
            seen += n
    assert seen > 0


def test_known_key_end_to_end(dev, O, small_fx):
    """key 0x1E9AD (1_9_7File.pb:189): dispenser walk on the GPU, resolver = oracle restatement of
    checkerThread; the recovered key must be bit-exact."""
    fx = small_fx
    kk = fx["known_key"]
    L = O.lib()
    gpu, cpu, g2 = (bytes.fromhex(fx[k]) for k in ("htgpu", "htcpu", "g2"))
    dev.upload_g2(g2, fx["t"], fx["b"], fx["p"])
    dev.upload_htgpu(gpu, 1 << fx["htsz"], fx["w"], 2)
    cb = C.create_string_buffer(cpu, len(cpu))
    job = O.Job()
    Q = O.Pt.from_ints(int(kk["qx"], 16), int(kk["qy"], 16))
    L.o_job_init(C.byref(job), fx["t"], fx["b"], fx["p"], fx["w"], fx["htsz"],
                 C.byref(O.Fe.from_int(int(kk["start"], 16))), C.byref(Q), None)
    found = None
    for tile in range(8):
        key, pub = O.Fe(), O.Pt()
        L.o_getjob(C.byref(job), C.byref(key), C.byref(pub))
        hits, _ = dev.step(*pub.to_ints())
        for code, idx in hits:
            out = O.Fe()
            if L.o_resolve_hit(C.byref(job), C.cast(cb, C.c_void_p), 1 << fx["htsz"], code, idx,
                               C.byref(key), C.byref(pub), C.byref(out)):
                found = out.to_int()
        if found:
            break
    assert found == 0x1E9AD


def test_cuda_compat_layer_runs_a_tile(small_fx):
    """the reference host's driver-API call sequence (1_9_7File.pb:2181-2353, 2442-2509) against the
    compat layer: one allocation laid out by the host, _A block, cuLaunchGrid, hit read-back."""
    import pybsgs
    fx = small_fx
    L = pybsgs.lib()
    t, b, p, w, htsz = fx["t"], fx["b"], fx["p"], fx["w"], fx["htsz"]
    maxnonce, items = t * b * p, 1 << htsz
    g2, gpu = bytes.fromhex(fx["g2"]), bytes.fromhex(fx["htgpu"])
    u64 = C.c_uint64
    for name in ("cuMemAlloc_v2", "cuModuleGetGlobal_v2"):
        getattr(L, name).restype = C.c_int
    L.cuMemcpyHtoD_v2.argtypes = [u64, C.c_void_p, u64]
    L.cuMemcpyDtoH_v2.argtypes = [C.c_void_p, u64, u64]
    L.cuMemFree_v2.argtypes = [u64]
    L.cuParamSeti.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.cuLaunchGrid.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.cuFuncSetBlockShape.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    assert L.cuInit(C.c_int64(0)) == 0
    ctx, mod, fn = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.cuCtxCreate_v2(C.byref(ctx), C.c_int64(4), C.c_int64(0)) == 0
    assert L.cuModuleLoadData(C.byref(mod), b"ptx text ignored") == 0
    assert L.cuModuleGetFunction(C.byref(fn), mod, b"_test1") == 0
    assert L.cuModuleGetFunction(C.byref(fn), mod, b"nope") == 500
    a_ptr, a_sz = u64(), u64()
    assert L.cuModuleGetGlobal_v2(C.byref(a_ptr), C.byref(a_sz), mod, b"_A") == 0 and a_sz.value == 120
    tab_bytes = 4 * (items + 1) + 4 * w
    puboffset = ((96 * maxnonce + 64 + 63) // 64) * 64 + 2048            # 1_9_7File.pb:2209-2216
    total = puboffset + tab_bytes + 4096
    base = u64()
    assert L.cuMemAlloc_v2(C.byref(base), u64(total)) == 0
    dptr = (base.value + 63) & ~63
    assert L.cuParamSetSize(fn, C.c_int64(8)) == 0
    assert L.cuParamSeti(fn, 0, dptr & 0xFFFFFFFF) == 0 and L.cuParamSeti(fn, 4, dptr >> 32) == 0
    assert L.cuFuncSetBlockShape(fn, t, 1, 1) == 0
    hdr = bytearray(2048)
    assert L.cuMemcpyHtoD_v2(dptr, bytes(hdr), 2048) == 0
    assert L.cuMemcpyHtoD_v2(dptr + 2048, g2, len(g2)) == 0
    assert L.cuMemcpyHtoD_v2(dptr + puboffset, gpu, len(gpu)) == 0
    A = bytearray(120)
    struct.pack_into("<I", A, 4, w)
    struct.pack_into("<I", A, 8, p)
    struct.pack_into("<I", A, 12, maxnonce)
    struct.pack_into("<Q", A, 96, puboffset)
    struct.pack_into("<I", A, 104, items + 1)
    struct.pack_into("<I", A, 112, items - 1)
    assert L.cuMemcpyHtoD_v2(a_ptr.value, bytes(A), 120) == 0
    for tl in fx["tiles"]:
        px, py = int(tl["px"], 16), int(tl["py"], 16)
        words = b"".join(struct.pack("<I", (v >> (32 * (7 - k))) & 0xFFFFFFFF) for v in (px, py) for k in range(8))
        assert L.cuMemcpyHtoD_v2(a_ptr.value + 32, words, 64) == 0
        assert L.cuLaunchGrid(fn, b, 1) == 0
        assert L.cuCtxSynchronize() == 0
        cnt = C.c_uint32()
        assert L.cuMemcpyDtoH_v2(C.byref(cnt), dptr, 4) == 0
        recs = (C.c_uint32 * (2 * max(cnt.value, 1)))()
        if cnt.value:
            assert L.cuMemcpyDtoH_v2(recs, dptr + 128, 8 * cnt.value) == 0
            zero = C.c_uint32(0)
            assert L.cuMemcpyHtoD_v2(dptr, C.byref(zero), 4) == 0          # host clears the counter (197:2502-2503)
        got = sorted(((recs[2 * i], recs[2 * i + 1]) for i in range(cnt.value)), key=lambda h: (h[1], h[0]))
        assert [list(h) for h in got] == tl["hits"], tl["kind"]
    assert L.cuMemFree_v2(base.value) == 0
    assert L.cuCtxDestroy_v2(ctx) == 0


@pytest.mark.parametrize("variant,kernel", [(13, "giant_pair2_kernel<2, false, true>"), (10, "giant_pair2_kernel<2, false, false>"), (0, "giant_tile_kernel<2>")])
def test_all_kernel_variants_agree_with_oracle(O, small_fx, variant, kernel):
    """the three tile kernels of the library -- chained, one stored product per four giants (the default); chained, one per pair; the per-giant
    fallback -- return the oracle's hit lists, many tiles per call; anything else in BSGS_KERNEL_VARIANT is refused"""
    import os
    import pybsgs
    os.environ["BSGS_KERNEL_VARIANT"] = "9"
    try:
        with pytest.raises(pybsgs.BsgsError):
            pybsgs.Device(0)
        os.environ["BSGS_KERNEL_VARIANT"] = str(variant)
        d = pybsgs.Device(0)
    finally:
        del os.environ["BSGS_KERNEL_VARIANT"]
    t, b, p, w, htsz = 64, 8, 12, 1 << 16, 14                       # T = 512: two 256-thread slices
    g2, gpu, centres = _planted_case(O, 4242, t, b, p, w, htsz, 12, 9)
    centres = centres + [O.pt_mul(k) for k in (5, 70000)] + [O.g2_unpack(g2, t, b, p, 77)]     # code 5 twice, an x-equal tile
    d.upload_g2(g2, t, b, p)
    d.upload_htgpu(gpu, 1 << htsz, w, 2)
    hits, n, _ = d.run(centres, 65536)
    ref_all = []
    for k, Pt in enumerate(centres):
        r, _ = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
        ref_all += [(k, c, i) for c, i in r]
    assert n == len(ref_all) and hits == ref_all
    assert d.last_kernel() == kernel
    # fixture tiles one at a time (sequence length 1)
    fx = small_fx
    d.upload_g2(bytes.fromhex(fx["g2"]), fx["t"], fx["b"], fx["p"])
    d.upload_htgpu(bytes.fromhex(fx["htgpu"]), 1 << fx["htsz"], fx["w"], 3)
    for tl in fx["tiles"]:
        h, _ = d.step(int(tl["px"], 16), int(tl["py"], 16))
        assert [list(x) for x in h] == tl["hits"]
    d.close()


def test_error_behaviour_of_the_c_abi(small_fx):
    """every entry point returns 0 or a negative code with a text from bsgs_last_error(); nothing falls back, nothing
    throws across the boundary (the reference pattern is exit("error <call>-<code>"), 1_9_7File.pb:2195-2197)"""
    import pybsgs
    d = pybsgs.Device(0)
    fx = small_fx
    Pt = (int(fx["tiles"][0]["px"], 16), int(fx["tiles"][0]["py"], 16))
    with pytest.raises(pybsgs.BsgsError, match="giants|g2|table|state"):
        d.step(*Pt)                                                   # nothing uploaded yet
    d.upload_g2(bytes.fromhex(fx["g2"]), fx["t"], fx["b"], fx["p"])
    with pytest.raises(pybsgs.BsgsError):
        d.step(*Pt)                                                   # giants but no table
    with pytest.raises(pybsgs.BsgsError, match="power of two"):
        d.upload_htgpu(bytes.fromhex(fx["htgpu"]), (1 << fx["htsz"]) - 1, fx["w"], 1)
    with pytest.raises(pybsgs.BsgsError, match="layout"):
        d.upload_htgpu(bytes.fromhex(fx["htgpu"]), 1 << fx["htsz"], fx["w"], 9)
    with pytest.raises(pybsgs.BsgsError):
        d.build_baby_table_ext(1 << 10, 8, 2)                         # the direct builder only makes the list formats
    with pytest.raises(pybsgs.BsgsError):
        d.build_baby_table_ext(1 << 37, 31, 4)                        # beyond 2^36
    with pytest.raises(pybsgs.BsgsError):
        d.upload_g2(bytes.fromhex(fx["g2"]), 0, fx["b"], fx["p"])
    # and the device still works after the failures
    d.upload_g2(bytes.fromhex(fx["g2"]), fx["t"], fx["b"], fx["p"])
    d.upload_htgpu(bytes.fromhex(fx["htgpu"]), 1 << fx["htsz"], fx["w"], 2)
    hits, n = d.step(*Pt)
    assert [list(h) for h in hits] == fx["tiles"][0]["hits"]
    d.close()


@pytest.mark.parametrize("layout", [1, 2, 4])
def test_odd_chain_length_takes_the_per_giant_kernel(dev, O, layout):
    """-p is even on the reference's command line (1_9_7File.pb:4616-4618) but the C-ABI accepts any p: an odd chain cannot be
    pair-batched, the engine falls back to the per-giant kernel (and its full-size chain scratch)"""
    t, b, p, w, htsz = 64, 3, 7, 1 << 14, 11
    g2, gpu, centres = _planted_case(O, 4242, t, b, p, w, htsz, 12, 3)
    dev.upload_g2(g2, t, b, p)
    dev.upload_htgpu(gpu, 1 << htsz, w, layout)
    for Pt in centres:
        ref, nref = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
        hits, n = dev.step(Pt[0], Pt[1], 65536)
        assert (n, hits) == (nref, ref) and nref >= 12
    # switching back to an even chain on the same device re-sizes the scratch correctly
    t, b, p = 64, 3, 8
    g2, gpu, centres = _planted_case(O, 4243, t, b, p, w, htsz, 12, 2)
    dev.upload_g2(g2, t, b, p)
    dev.upload_htgpu(gpu, 1 << htsz, w, layout)
    hits, n, _ = dev.run(centres, 65536)
    want = []
    for k, Pt in enumerate(centres):
        r, _ = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
        want += [(k, c, i) for c, i in r]
    assert hits == want


def test_hit_buffer_overflow_is_reported_not_hidden(dev, O):
    """more hits than the caller's buffer: the total is still reported, the first max_hits of the sorted list are
    returned and the call says BSGS_ERR_OVERFLOW (the reference's buffer holds 240 records per launch, 1_9_7File.pb:2209)"""
    import ctypes as C
    import pybsgs
    t, b, p, w, htsz = 64, 8, 32, 1 << 22, 2            # 2^20 entries per bucket: ~8 hash collisions per tile
    g2 = O.build_g2(t, b, p, w)
    keys = np.frombuffer(np.random.default_rng(5).bytes(8 * w), dtype=np.uint64)
    gpu, _ = O.pack_tables_from_keys(keys, htsz)
    dev.upload_g2(g2, t, b, p)
    dev.upload_htgpu(gpu, 1 << htsz, w, 4)
    Pt = O.pt_mul(0xABCDEF123456789)
    ref, nref = O.tile_ref(Pt, g2, t, b, p, gpu, htsz, 0, 65536)
    assert nref >= 4
    hits = (pybsgs.Hit * 2)()
    n = C.c_uint32()
    rc = dev.L.bsgs_step(dev.h, pybsgs.le32(Pt[0]), pybsgs.le32(Pt[1]), hits, 2, C.byref(n))
    assert rc == pybsgs.ERR_OVERFLOW and n.value == nref
    assert [(hits[i].code, hits[i].idx) for i in range(2)] == ref[:2]
    assert b"hits" in dev.L.bsgs_last_error()
    got, n2 = dev.step(Pt[0], Pt[1], 65536)             # the counter was reset: the next call is complete again
    assert (n2, got) == (nref, ref)
