"""Pin the CPU oracle (oracle/curve64_ref.c) against every known-answer vector the reference
holds for this path: the self-test of lib/Curve64.pb (Curve64.pb:3067-3397, values copied in
SURVEY.md Appendix D) and the known-key vectors of 1_9_7File.pb:189,191,200-203."""
import ctypes as C

import oracle_lib as O
from oracle_lib import Fe, Pt

P, N, GX, GY = O.P_INT, O.N_INT, O.GX_INT, O.GY_INT
A = 0x3fdc2a05828a06c18e057a8d9549bdc3ff05ee69a352342ce382aafeaeb98ef9
B = 0xdfcad171d3196bdb20eaaf272f8f9bcc6b5a47d4fe53d3d874e703cd2566197e


def h(s):
    return int(s.replace(" ", ""), 16)


def test_neg_gy():  # Curve64.pb:3088-3095
    assert O.fe_op("o_subModX64", P, GY, mod=P) == h("b7c52588d95c3b9aa25b0403f1eef75702e84bb7597aabe663b82f6f04ef2777")


def test_and_add_sub_raw():  # Curve64.pb:3190-3227
    L = O.lib()
    r = Fe()
    L.o_andX64(C.byref(r), C.byref(Fe.from_int(A)), C.byref(Fe.from_int(B)))
    assert r.to_int() == h("1fc80001820802c100002a05050999c06b004640a2521008608202cc24200878")
    carry = L.o_addX64(C.byref(r), C.byref(Fe.from_int(A)), C.byref(Fe.from_int(B)))
    assert carry == 1 and r.to_int() == h("1fa6fb7755a3729caef029b4c4d959906a60363ea1a608055869aecbd41fa877")
    borrow = L.o_subX64(C.byref(r), C.byref(Fe.from_int(A)), C.byref(Fe.from_int(B)))
    assert borrow == 1 and r.to_int() == h("60115893af709ae66d1acb6665ba21f793aba694a4fe60546e9ba7318953757b")


def test_addmod_submod():  # Curve64.pb:3230-3255
    assert O.fe_op("o_addModX64", A, B, mod=P) == h("1fa6fb7755a3729caef029b4c4d959906a60363ea1a608055869aeccd41fac48")
    assert O.fe_op("o_subModX64", A, B, mod=P) == h("60115893af709ae66d1acb6665ba21f793aba694a4fe60546e9ba730895371aa")


def test_squares():  # Curve64.pb:3135-3168
    v = h("342119815c0f816f31f431a9fe98a6c76d11425ecaeaecf2d0ef6def197c56b0")
    assert O.fe_op("o_squareModX64", v) == h("38f37014ce22fc29cf19f28a5ce4da091445536c3e2cff318ba07c2a3048f518")
    assert O.fe_op("o_squareModX64", A) == h("3d6c452d1c076d0425ac63c7783f563df3ec12324d0f16bf7c8335253ef4be33")
    assert O.fe_op("o_squareModX64", GY) == h("4866d6a5ab41ab2c6bcc57ccd3735da5f16f80a548e5e20a44e4e9b8118c26f2")
    assert O.fe_op("o_mulModX64", A, A) == h("3d6c452d1c076d0425ac63c7783f563df3ec12324d0f16bf7c8335253ef4be33")


def test_modinv():  # Curve64.pb:3260-3267
    assert O.fe_op("o_modInvX64", GX, mod=P) == h("237afdf1d2938d86870aaeb8ad77626a67b8e794abfb076be61d003687ca9ef6")


def test_points():  # Curve64.pb:3113-3132, 3270-3320, 3370-3392
    G = (GX, GY)
    g2 = O.pt_add(G, G)
    assert g2 == (h("c6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5"),
                  h("1ae168fea63dc339a3c58419466ceaeef7f632653266d0e1236431a950cfe52a"))
    assert O.pt_add(G, g2) == (h("f9308a019258c31049344f85f89d5229b531c845836f99b08601f113bce036f9"),
                               h("388f7b0f632de8140fe337e62a37f3566500a99934c2231b6cb9fd7584b8e672"))
    assert O.pt_add(g2, g2) == (h("e493dbf1c10d80f3581e4904930b1404cc6c13900ee0758474fa94abe8c4cd13"),
                                h("51ed993ea0d455b75642e2098ea51448d967ae33bfbdfe40cfe97bdc47739922"))
    acc = G
    for _ in range(10000):
        acc = O.pt_add(acc, G)
    assert acc == (h("db7432110ba814bfe6371ddfd03ba554b558548aa90e81b8e1421321656065a8"),
                   h("8236f24d965a900384b382e8d772d7e92dee2ce6c3cb33883ea627d54a5170c4"))
    assert O.pt_mul(10001) == acc
    assert O.pt_mul(A) == (h("510f6efbef396a1985da989104a295063606319beafa4e1fd0ebd29ace19088f"),
                           h("fcf1cb9e1a9c02fea09e983fe5fe8fb7ce74a80ed3b1783706e27bde4b2ede5e"))


def test_known_keys():  # 1_9_7File.pb:189, 191, 200-203
    assert O.pt_mul(0x1E9AD) == (h("e1e5e6f7b0b8d67604e3940c87bf06b814cedc486112b9956c68e3d78b1bd812"),
                                 h("97fe4f65fbd6e9f7eb1eea80b144d1487f2a9b0aeae5fcf6f43b41491641884e"))
    L = O.lib()
    for key, comp in ((0x16f7027bbf8454a5c, b"036d05521c67b9cc1c0ef906b42215c7120c7302c34d9316a2726199bedac50936"),
                      (0xf7051f27b09112d4, b"03100611c54dfef604163b8358f7b7fac13ce478e02cb224ae16d45526b25d9d4d")):
        q = Pt()
        assert L.o_parse_pubkey(C.byref(q), comp) == 0
        assert q.to_ints() == O.pt_mul(key)
        out = C.create_string_buffer(67)
        L.o_compress_pub(out, C.byref(q))
        assert out.value == comp


def test_random_against_python_ints():
    import random
    rnd = random.Random(1234)
    for _ in range(300):
        a, b = rnd.randrange(P), rnd.randrange(P)
        assert O.fe_op("o_mulModX64", a, b) == a * b % P
        assert O.fe_op("o_squareModX64", a) == a * a % P
        assert O.fe_op("o_addModX64", a, b, mod=P) == (a + b) % P
        assert O.fe_op("o_subModX64", a, b, mod=P) == (a - b) % P
        if a:
            assert O.fe_op("o_modInvX64", a, mod=P) == pow(a, -1, P)
    for a in (0, 1, P - 1, 2**256 - 1, 2**255, P + 5):      # unreduced operands still reduce correctly
        for b in (1, P - 1, 2**256 - 1, 0x1000003D1):
            assert O.fe_op("o_mulModX64", a, b) == a * b % P


def test_fast_cpu_keys_equal_the_literal_port():
    """oracle/cpu_fast.c (the multi-threaded CPU implementation the whole-tile GPU test takes its 2^25 keys from) lists, key by key, what the
    literal restatement of the reference kernel lists (o_tile_ref_slice_keys): a random tile and an equal-x tile (x(2P) slot)"""
    import numpy as np
    import oracle_lib as O
    t, b, p, w = 8, 4, 16, 1 << 12
    g2 = np.frombuffer(O.build_g2(t, b, p, w), dtype=np.uint8)
    for P in (O.pt_mul(987654321987654321), O.g2_unpack(g2.tobytes(), t, b, p, 77), O.pt_neg(O.g2_unpack(g2.tobytes(), t, b, p, 300))):
        a = O.tile_slice_keys(P, g2, t, b, p, 0, t * b)
        f = O.fast_tile_slice_keys(P, g2, t, b, p, 0, t * b, 4)
        assert a.shape == f.shape == (t * b, p, 2) and (a == f).all()


def test_fast_cpu_keys_of_scalars_equal_the_literal_port_and_the_table_builder():
    """o_fast_keys_of_scalars_mt (the expected keys of the GPU tests' sampled-membership check: low 64 bits of x(k*G), Jacobian additions over a table
    of 2^j*G, batched inversion, host threads) against the literal port's scalar multiplication (o_PTMULX64, Curve64.pb LSB-first double-and-add) and
    against the keys the oracle's table builder files under (o_build_baby_tables: bucket = x & mask, hash = bits 32..63)"""
    import random
    import numpy as np
    import oracle_lib as O
    rnd = random.Random(77)
    ks = [1, 2, 3, 255, 256, 2**32 - 1, 2**32, 2**32 + 1, 2**33, 2**34, 2**36 - 1, 2**63 + 12345, 2**64 - 1] + [rnd.getrandbits(rnd.randint(1, 64)) or 1 for _ in range(120)]
    got = O.fast_keys_of_scalars(ks, nthreads=3)
    for k, g in zip(ks, got):
        assert O.pt_mul(k)[0] & (2**64 - 1) == int(g), k
    # block boundaries of the batched inversion (256 keys per block) and of the thread split
    ks2 = list(range(1, 1200))
    assert [int(v) for v in O.fast_keys_of_scalars(ks2, nthreads=4)] == [int(v) for v in O.fast_keys_of_scalars(ks2, nthreads=1)]
    w, htsz = 1024, 8
    htgpu, _ = O.build_baby_tables(w, htsz)
    img = np.frombuffer(htgpu, dtype=np.uint32)
    starts, items = img[: (1 << htsz) + 1], img[(1 << htsz) + 1:]
    for k, key in zip(ks2[:w], O.fast_keys_of_scalars(ks2[:w])):
        b, h = int(key) & ((1 << htsz) - 1), int(key) >> 32
        assert h in items[starts[b]:starts[b + 1]].tolist(), k
