"""Parity at BASELINE.json's FULL sizes through size-independent properties (GPU):
  * config 2 (-t 256 -b 256 -p 256 -w 26 -htsz 25) and the metric's -w 30 -htsz 28 table, REAL tables and giants;
  * tiles whose centre is m*G for crafted m: the analytic hit list {(1,i): m-(i+1)2w in +-[1,w]} u {(2,i): m+(i+1)2w in
    +-[1,w]} u {(5,-): |m| <= w} must be reported, at the first / middle / last giant index;
  * every reported hit beyond the analytic ones must be a genuine 32-bit hash collision: the exact CSR layout and the
    bucket-line layout must return identical lists (two independent probe implementations over the same table)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def analytic_hits(m, w, maxnonce):
    hits = []
    if 1 <= abs(m) <= w:
        hits.append((5, 0xFFFFFFFF))
    for code, sgn in ((1, -1), (2, +1)):                 # P + G2[i] = (m - (i+1)2w)G ; P - G2[i] = (m + (i+1)2w)G
        # |m + sgn*(i+1)*2w| in [1, w]
        centre = -sgn * m                                # (i+1)*2w close to centre
        for i1 in {centre // (2 * w), centre // (2 * w) + 1}:
            if 1 <= i1 <= maxnonce and 1 <= abs(m + sgn * i1 * 2 * w) <= w:
                hits.append((code, i1 - 1))
    return sorted(set(hits), key=lambda h: (h[1], h[0]))


@pytest.mark.parametrize("wexp,htsz", [(26, 25), (30, 28)])
def test_fullsize_planted_and_layout_agreement(wexp, htsz):
    import pybsgs
    from pybsgs import ecpy
    t, b, p, w = 256, 256, 256, 1 << wexp
    maxnonce = t * b * p
    dev = pybsgs.Device(0)
    img = torch.empty((1 << htsz) + 1 + w, dtype=torch.int32, device="cuda:0")
    dev.build_baby_tables_device(w, htsz, img.data_ptr())
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    ms = [(0 + 1) * 2 * w + 77,                       # code 1 at the first giant
          -((maxnonce) * 2 * w) + 12345,              # code 2 at the last giant
          (maxnonce // 2) * 2 * w - w,                # edge b' = w: two adjacent giants see it (code 1)
          w // 3,                                     # code 5 (P itself is a baby point)
          -(5000001 * 2 * w) - 1,                     # code 2, b' = 1
          (maxnonce + 5) * 2 * w + 99,                # beyond the last giant: nothing
          3 * 2 * w * 1000003 + (w - 7)]
    centres = [ecpy.mul(m % N) for m in ms]
    results = {}
    for layout in (pybsgs.TABLE_LINES64, pybsgs.TABLE_CSR, pybsgs.TABLE_LINES64_LIST):
        dev.upload_htgpu_device(img.data_ptr(), 1 << htsz, w, layout)
        assert dev.table_info()[0] == layout
        hits, n, _ = dev.run(centres, 65536)
        assert n == len(hits)
        results[layout] = hits
    assert results[pybsgs.TABLE_LINES64] == results[pybsgs.TABLE_CSR]          # two probe implementations, one answer
    assert results[pybsgs.TABLE_LINES64_LIST] == results[pybsgs.TABLE_CSR]     # lines + overflow list, CSR dropped
    if wexp == 30:
        # the direct builder (atomic scatter, no sort, no CSR: the w >= 2^32 path) must give the same table semantics
        del img
        torch.cuda.empty_cache()
        dev.build_baby_table_ext(w, htsz, pybsgs.TABLE_LINES64_LIST)
        lay, nbytes, ovf = dev.table_info()
        assert lay == pybsgs.TABLE_LINES64_LIST and nbytes >= 64 << htsz and ovf > 0
        hits, n, _ = dev.run(centres, 65536)
        assert hits == results[pybsgs.TABLE_CSR]
    got = results[pybsgs.TABLE_LINES64]
    extra = 0
    for k, m in enumerate(ms):
        mine = [(c, i) for tile, c, i in got if tile == k]
        expect = analytic_hits(m, w, maxnonce)
        assert set(expect) <= set(mine), (k, expect, mine)
        extra += len(mine) - len(expect)
    # 32-bit hash collisions: ~ 2 * maxnonce * (w / 2^htsz) / 2^32 per tile  (0.016 at -w 26, 0.03 at -w 30)
    assert extra <= 3
    assert ms[5] and not analytic_hits(ms[5], w, maxnonce)
    dev.close()


def verify_table(dev, w, n_in=100000, n_out=100000, seed=1, extra_k=()):
    """The reference verifies every table it builds or loads (checkHT / checkHTpack: sampled "every k*G is found", 1_9_7File.pb:3599-3627, 3101-3134;
    ascending buckets, 1_9_7File.pb:2797-2805).  Same here, for any layout and size: (1) the census -- one streaming pass over what the device holds:
    entries in lines + keys in the overflow set - duplicates must equal w, no malformed line; (2) batched membership through the SHIPPED probe:
    n_in random k in [1, w] (with k = 1, w, the 32-bit boundaries of the reference format) must ALL be found, n_out random k in (w, 5w] must all miss
    except by a 32-bit hash collision (probability load / 2^32 each).  Expected keys: oracle/cpu_fast.c on host threads, pinned to the literal port
    (tests/test_oracle_kat.py).  Returns (census, seconds spent in census + lookups)."""
    import time
    import numpy as np
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    special = [k for k in (1, 2, w, w - 1, w // 2, 1 << 32, (1 << 32) - 1, (1 << 32) + 1, 1 << 33, (1 << 33) + 1) + tuple(extra_k) if 1 <= k <= w]
    k_in = np.concatenate([np.array(special, dtype=np.uint64), rng.integers(1, w + 1, size=n_in, dtype=np.uint64)])
    k_out = rng.integers(w + 1, 5 * w + 1, size=n_out, dtype=np.uint64)
    keys_in, keys_out = O.fast_keys_of_scalars(k_in), O.fast_keys_of_scalars(k_out)
    t0 = time.time()
    c = dev.table_census()
    found_in = dev.table_lookup(keys_in.tolist())
    found_out = dev.table_lookup(keys_out.tolist())
    dt = time.time() - t0
    assert c["w"] == w and c["total"] == w, c
    assert c["malformed_lines"] == 0, c
    missing = [int(k) for k, f in zip(k_in, found_in) if not f]
    assert not missing, ("baby points missing from the table", missing[:10], len(missing))
    fp = sum(found_out)
    return c, dt, fp


def test_census_and_sampled_membership_of_small_tables_in_every_layout():
    """census + membership (verify_table) on small tables in every device layout: the three made from a reference-format image (with and without resident
    CSR), the CSR image alone, direct-built 64- and 128-byte lines with heavy overflow, and direct-built tables with a bucket count that is NOT a power
    of two (the bucket then comes from 48 bits of the key; 64- and 128-byte lines)."""
    import pybsgs
    dev = pybsgs.Device(0)
    wexp, htsz = 20, 17                                    # load 8
    w = 1 << wexp
    img = torch.empty((1 << htsz) + 1 + w, dtype=torch.int32, device="cuda:0")
    dev.build_baby_tables_device(w, htsz, img.data_ptr())
    for layout in (pybsgs.TABLE_CSR, pybsgs.TABLE_LINES64, pybsgs.TABLE_LINES128, pybsgs.TABLE_LINES64_LIST, pybsgs.TABLE_LINES128_LIST):
        dev.upload_htgpu_device(img.data_ptr(), 1 << htsz, w, layout)
        c, _, fp = verify_table(dev, w, 20000, 20000, seed=layout)
        assert c["unsorted_lines"] == 0, (layout, c)       # the reference's files have ascending buckets, and so have the lines made from them
        assert fp <= 2, (layout, fp)                       # 2e4 * 8 / 2^32
    del img
    for w2, hb, layout in ((1 << 22, 18, pybsgs.TABLE_LINES64_LIST),          # load 16: most 64-byte lines over-full
                           (1 << 22, 18, pybsgs.TABLE_LINES128_LIST),
                           (1 << 22, 150001, pybsgs.TABLE_LINES128_LIST),      # 150001 buckets (not a power of two), load 28: many 128-byte lines over-full
                           (1 << 22, 393241, pybsgs.TABLE_LINES64_LIST),       # 393241 buckets of 64 bytes (not a power of two), load 10.67: -w 35 on 3 * 2^30 lines in small (one line in 13 over-full)
                           (3 * (1 << 20), 3 * (1 << 18) + 1, pybsgs.TABLE_LINES64_LIST),   # load 4, an odd bucket count
                           (3 * (1 << 20) + 12345, 3 * (1 << 17), pybsgs.TABLE_LINES128_LIST),   # 1.5 * 2^18 buckets, load 8: -w 35's shape in small
                           (1000003, 48611, pybsgs.TABLE_LINES128_LIST)):
        dev.build_baby_table_ext(w2, hb, layout)
        buckets = hb if hb > 31 else 1 << hb
        lay, nbytes, over = dev.table_info()
        assert lay == layout and nbytes >= buckets * (64 if layout == pybsgs.TABLE_LINES64_LIST else 128)
        c, _, fp = verify_table(dev, w2, 20000, 20000, seed=hb)
        assert c["overfull_lines"] == over and c["unsorted_lines"] == 0, (c, over)      # (round 5: the direct builder closes its lines sorted)
        assert fp <= 3
    dev.close()


def test_census_sees_a_damaged_line_header(request):
    """one bit of a line header (an entry count changes by one: a word of padding becomes an "entry", or an entry is dropped) shows in the census
    (runs in the TEST library: the hook that damages a table is not part of the shipped one)"""
    from conftest import rerun_in_test_library
    if rerun_in_test_library(request):
        return
    import pybsgs
    dev = pybsgs.Device(0)
    for w2, hb, layout, words in ((1000003, 48611, pybsgs.TABLE_LINES128_LIST, 32), (3 * (1 << 20), 3 * (1 << 18) + 1, pybsgs.TABLE_LINES64_LIST, 16)):
        dev.build_baby_table_ext(w2, hb, layout)
        c0 = dev.table_census()
        assert c0["total"] == w2 and c0["malformed_lines"] == 0
        dev.debug_corrupt_table(4 * words * 7, 1)
        c1 = dev.table_census()
        assert c1["total"] != c0["total"] or c1["malformed_lines"] > 0, (c0, c1)
    dev.close()


def test_extended_table_w34():
    """BASELINE configs 3/5 geometry: -w 34 -htsz 31 (2^34 baby points, 128 GiB of bucket lines + overflow list), built on
    the GPU by bsgs_build_baby_table_ext.  No reference format exists at this size (1_9_7File.pb:4412-4418 caps w below
    2^32), so parity is through the analytic hit list of crafted centres; everything else must be a hash collision
    (expected 2 * 2^24 * 8 / 2^32 = 0.06 per tile)."""
    import pybsgs
    from pybsgs import ecpy
    from conftest import free_hbm
    free = free_hbm(200 * 2**30)
    if free < 200 * 2**30:
        pytest.skip("needs ~150 GiB of free HBM")
    wexp, htsz = 34, 31
    t, b, p, w = 256, 256, 256, 1 << wexp
    maxnonce = t * b * p
    dev = pybsgs.Device(0)
    dev.build_baby_table_ext(w, htsz, pybsgs.TABLE_LINES64_LIST)
    lay, nbytes, ovf = dev.table_info()
    assert lay == pybsgs.TABLE_LINES64_LIST and nbytes >= 64 << htsz
    assert 0.005 * 2**31 < ovf < 0.012 * 2**31          # P(Poisson(8) > 15) = 0.82 % of the buckets overflow a 64-byte line
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    ms = [(0 + 1) * 2 * w + 77,                       # code 1 at the first giant, b' = 77
          -((maxnonce) * 2 * w) + 12345,              # code 2 at the last giant
          (maxnonce // 2) * 2 * w - w,                # edge b' = w (the LAST baby point, k = 2^34): two adjacent giants
          w // 3,                                     # code 5
          -(5000001 * 2 * w) - 1,                     # code 2, b' = 1
          (maxnonce + 5) * 2 * w + 99,                # beyond the last giant: nothing
          3 * 2 * w * 1000003 + (w - 7),
          7 * 2 * w + (1 << 32) + 5,                  # baby indices around the 32-bit boundary of the reference format
          9 * 2 * w + (1 << 33) - 1]
    centres = [ecpy.mul(m % N) for m in ms]
    got, n, _ = dev.run(centres, 65536)
    assert n == len(got)
    extra = 0
    for k, m in enumerate(ms):
        mine = [(c, i) for tile, c, i in got if tile == k]
        expect = analytic_hits(m, w, maxnonce)
        assert set(expect) <= set(mine), (k, expect, mine)
        extra += len(mine) - len(expect)
    assert extra <= 4
    # the table itself, pinned the way the reference pins its own: census (lines + set - duplicates = 2^34) and 2 x 10^5 sampled
    # keys through the shipped probe, in well under 10 s
    c, dt, fp = verify_table(dev, w, 100000, 100000, seed=34)
    assert c["overfull_lines"] == ovf and c["set_keys"] > 0 and c["duplicates"] > 0, c
    assert fp <= 3 and dt < 10.0, (fp, dt)
    print("census -w 34:", c, "census + 2e5 lookups: %.2f s" % dt)
    dev.close()


@pytest.mark.parametrize("w,buckets,layout,kernel", [(1 << 35, 3 << 30, 4, "giant_pair2_kernel<4, false, true>"), (1 << 35, 3 << 29, 5, "giant_pair2_kernel<3, false, true>"),
                                                     (36 << 30, 3 << 30, 4, "giant_pair2_kernel<4, false, true>")])
def test_extended_table_w35(w, buckets, layout, kernel):
    """-w 35, the table that fills one MI355X (Tune's choice for an 80-bit range): 2^35 baby points in 3 * 2^30 bucket lines of 64 bytes (load 10.67 of
    14, a 16 GiB overflow set) and in 1.5 * 2^30 lines of 128 bytes (load 21.3 of 30) -- bucket counts that are no power of two, so the bucket comes from 48 bits of the key.
    Pinned like -w 34: the analytic hit lists of crafted centres through the shipped tile kernel (everything else must be a hash collision), the census
    (lines + set - duplicates = 2^35 exactly) and 2 x 10^5 sampled keys through the shipped probe, incl. k = 2^32, 2^33, 2^34 +- 1 and k = w.
    Third case: 36 * 2^30 points on the same 3 * 2^30 lines of 64 bytes (load 12 of 14, 15.6 % of the lines over-full, a 32 GiB overflow set at load 0.47) -- Tune's choice for
    an 80-bit range since the headers of over-full lines carry the overflow fingerprint (profiles/r08g_*): a count that is no power of two either."""
    import pybsgs
    from pybsgs import ecpy
    from conftest import free_hbm
    free = free_hbm(250 * 2**30)
    if free < 250 * 2**30:
        pytest.skip("needs ~245 GiB of free HBM")
    t, b, p = 256, 256, 256
    maxnonce = t * b * p
    dev = pybsgs.Device(0)
    dev.build_baby_table_ext(w, buckets, layout)
    lay, nbytes, ovf = dev.table_info()
    assert lay == layout and nbytes >= (64 if layout == 4 else 128) * buckets
    # over-full lines: P(Poisson(10.67) > 14) = 11.4 % minus the lines whose 15th... entries: measured 7.6 % of 3 * 2^30; P(Poisson(21.33) > 30) = 1.84 % of 1.5 * 2^30
    # 36 * 2^30 points: P(Poisson(12) > 14) = 22.8 %, measured 15.6 %
    lo, hi = ((0.06, 0.09) if w == 1 << 35 else (0.14, 0.17)) if layout == 4 else (0.015, 0.022)
    assert lo * buckets < ovf < hi * buckets, ovf
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    ms = [(0 + 1) * 2 * w + 77,                       # code 1 at the first giant, b' = 77
          -((maxnonce) * 2 * w) + 12345,              # code 2 at the last giant
          (maxnonce // 2) * 2 * w - w,                # edge b' = w (the LAST baby point, k = 2^35): two adjacent giants
          w // 3,                                     # code 5
          -(5000001 * 2 * w) - 1,                     # code 2, b' = 1
          (maxnonce + 5) * 2 * w + 99,                # beyond the last giant: nothing
          3 * 2 * w * 1000003 + (w - 7),
          7 * 2 * w + (1 << 32) + 5,                  # baby indices around the word boundaries
          9 * 2 * w + (1 << 34) - 1,
          11 * 2 * w + (1 << 34) + (1 << 33) + 12345]
    centres = [ecpy.mul(m % N) for m in ms]
    got, n, _ = dev.run(centres, 65536)
    assert n == len(got) and dev.last_kernel() == kernel, dev.last_kernel()
    extra = 0
    for k, m in enumerate(ms):
        mine = [(c, i) for tile, c, i in got if tile == k]
        expect = analytic_hits(m, w, maxnonce)
        assert set(expect) <= set(mine), (k, expect, mine)
        extra += len(mine) - len(expect)
    assert extra <= 6                                  # expected 2 * 2^24 * (10.67 | 21.33) / 2^32 = 0.08 | 0.17 per tile
    c, dt, fp = verify_table(dev, w, 100000, 100000, seed=35, extra_k=((1 << 34) - 1, 1 << 34, (1 << 34) + 1, (1 << 35) - 1, 1 << 35, (1 << 35) + 1))
    assert c["overfull_lines"] == ovf and c["set_keys"] > 0 and c["duplicates"] > 0 and c["unsorted_lines"] == 0, c
    assert fp <= 4 and dt < 15.0, (fp, dt)
    print("census, %d points in %d lines of %d bytes:" % (w, buckets, 64 if layout == 4 else 128), c, "census + 2e5 lookups: %.2f s" % dt)
    dev.close()


def test_extended_table_built_into_caller_memory_and_installed_borrowed():
    """the RCCL-broadcast route of an extended table (bench.py, N > 1): build into caller-owned device buffers, install
    them borrowed; must answer exactly like the engine-owned build of the same table"""
    import pybsgs
    from pybsgs import ecpy
    t, b, p, wexp, htsz = 64, 16, 64, 22, 18          # load 16 per bucket: 64-byte lines overflow often, 128-byte ones rarely
    w = 1 << wexp
    dev = pybsgs.Device(0)
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    maxnonce = t * b * p
    ms = [5 * 2 * w + 77, -(maxnonce * 2 * w) + 12345, w // 3, 17 * 2 * w - w, (maxnonce + 5) * 2 * w + 99]
    centres = [ecpy.mul(m % N) for m in ms]
    for lay in (pybsgs.TABLE_LINES64_LIST, pybsgs.TABLE_LINES128_LIST):
        dev.build_baby_table_ext(w, htsz, lay)
        want, nw, _ = dev.run(centres, 65536)
        info = dev.table_info()
        cap = dev.ext_overflow_capacity(w, htsz, lay)
        lines = torch.empty((1 << htsz) * (16 if lay == pybsgs.TABLE_LINES64_LIST else 32), dtype=torch.int32, device="cuda:0")
        ovf = torch.empty(cap, dtype=torch.int64, device="cuda:0")
        n_ovf, n_over = dev.build_baby_table_ext_device(w, htsz, lay, lines.data_ptr(), ovf.data_ptr(), cap)
        dev.install_table_ext_device(lines.data_ptr(), ovf.data_ptr(), n_ovf, n_over, w, htsz, lay)
        assert dev.table_info() == info and n_ovf <= cap
        got, ng, _ = dev.run(centres, 65536)
        assert (ng, got) == (nw, want) and nw >= 4
        for k, m in enumerate(ms):
            assert set(analytic_hits(m, w, maxnonce)) <= {(c, i) for tile, c, i in got if tile == k}
    dev.close()


def test_gpu_built_tables_pass_the_reference_verifiers_at_config2_size():
    """The reference checks its tables structurally (1_9_7File.pb:2797-2805, 2911-2912): every k*G, k in
    [1, w], is found in htCPU with position k-1, and every bucket is ascending.  Same checks on the GPU-built images at
    BASELINE config 2 size (-w 26 -htsz 25), on the device with torch, plus a spot check against plain Python integers."""
    import random
    import pybsgs
    from pybsgs import ecpy
    wexp, htsz = 26, 25
    w, items = 1 << wexp, 1 << htsz
    dev = pybsgs.Device(0)
    g = torch.empty(items + 1 + w, dtype=torch.int32, device="cuda:0")
    c = torch.empty(items + 1 + 2 * w, dtype=torch.int32, device="cuda:0")
    dev.build_baby_tables_device(w, htsz, g.data_ptr(), c.data_ptr())
    u32 = lambda t: t.to(torch.int64) & 0xFFFFFFFF                                   # noqa: E731
    starts = u32(g[: items + 1])
    assert int(starts[0]) == 0 and int(starts[-1]) == w and bool((starts[1:] >= starts[:-1]).all())
    assert torch.equal(c[: items + 1], g[: items + 1])                               # same header in both files
    counts = starts[1:] - starts[:-1]
    bucket = torch.repeat_interleave(torch.arange(items, device="cuda:0"), counts)
    hashes = u32(g[items + 1:])
    key = (bucket << 32) | hashes
    assert bool((key[1:] >= key[:-1]).all())                                          # ascending inside every bucket
    pairs = c[items + 1:].view(w, 2)
    assert torch.equal(u32(pairs[:, 0]), hashes)                                      # htCPU carries the same hashes ...
    pos = u32(pairs[:, 1])
    assert torch.equal(torch.sort(pos).values, torch.arange(w, device="cuda:0"))      # ... and every position exactly once
    # load statistics of a uniform hash: mean 2, P(count > 15) negligible
    assert int(counts.max()) < 24 and abs(float(counts.float().mean()) - 2.0) < 1e-6
    rnd = random.Random(26)
    for k in [1, 2, w, w - 1, w // 2] + [rnd.randrange(1, w + 1) for _ in range(40)]:
        x = ecpy.mul(k)[0]
        b, h = x & (items - 1), (x >> 32) & 0xFFFFFFFF
        lo, hi = int(starts[b]), int(starts[b + 1])
        seg_h, seg_p = hashes[lo:hi].tolist(), pos[lo:hi].tolist()
        assert (h, k - 1) in list(zip(seg_h, seg_p)), k
    dev.close()


def test_gpu_generated_giants_at_full_geometry():
    """the reference verifies G2[i] == (i+1)*ADDPUBG after loading the giants (1_9_7File.pb:1543-1556); same check on the
    2^24 giants of -t 256 -b 256 -p 256 built by the GPU generator, read back in the reference's file layout"""
    import random
    import oracle_lib as O
    import pybsgs
    from pybsgs import ecpy
    O.lib()
    t, b, p, w = 256, 256, 256, 1 << 30
    n = t * b * p
    dev = pybsgs.Device(0)
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    img = dev.download_g2(64 * n)
    assert len(img) == 64 * n
    rnd = random.Random(24)
    for i in [0, 1, p - 1, p, n - 1, n - p, n // 2] + [rnd.randrange(n) for _ in range(13)]:
        assert O.g2_unpack(img, t, b, p, i) == ecpy.mul((i + 1) * (N - 2 * w) % N), i
    dev.close()


def test_hit_lists_of_two_independent_kernels_agree_under_load():
    """144 tiles (4.8e9 probes) at the metric's size through the default kernel (LDS-DMA staged probes behind counted
    vmcnt waits, pair-batched chain) and through the plain per-giant kernel with the exact CSR probe: the complete hit
    lists -- about 4 genuine 32-bit hash collisions plus the planted ones -- must be identical.  A probe compared before
    its line has landed, or a chain entry read before it was written, would show up as a difference."""
    import pybsgs
    from pybsgs import ecpy
    wexp, htsz = 30, 28
    t, b, p, w = 256, 256, 256, 1 << wexp
    maxnonce = t * b * p
    dev = pybsgs.Device(0)
    img = torch.empty((1 << htsz) + 1 + w, dtype=torch.int32, device="cuda:0")
    dev.build_baby_tables_device(w, htsz, img.data_ptr())
    A = ecpy.addpubg(w)
    dev.generate_g2(A[0], A[1], t, b, p)
    gstep, stride_pt = ecpy.tile_stride(t, b, p, w)
    cur = ecpy.mul(0xDEADBEEFCAFE1234567)
    centres = []
    for k in range(144):
        centres.append(cur)
        cur = ecpy.add(cur, stride_pt)
    centres[5] = ecpy.mul((777 * 2 * w + 31337) % N)            # planted: code 1 at giant 776
    centres[143] = ecpy.mul((N - (maxnonce * 2 * w) + 99) % N)  # planted: code 2 at the last giant
    # the default kernel's scratch lies in pieces of 32 tiles (8 bytes per giant: one stored product per four giants): a planted hit in the first and
    # the last tile of every piece, and around the former 16-tile boundaries
    edge_tiles = sorted({t for k in range(5) for t in (32 * k, min(32 * k + 31, 142))} | {15, 16, 47, 48, 111, 112} - {5, 143})
    for t in edge_tiles:
        centres[t] = ecpy.mul(((t + 100) * 2 * w + 1000 + t) % N)     # code 1 at giant t + 99
    res = {}
    for layout in (pybsgs.TABLE_LINES64, pybsgs.TABLE_CSR):
        dev.upload_htgpu_device(img.data_ptr(), 1 << htsz, w, layout)
        hits, n, _ = dev.run(centres, 65536)
        assert n == len(hits)
        res[layout] = hits
        if layout == pybsgs.TABLE_LINES64:
            # the default kernel's scratch (sized for a full launch of 192 tiles: 24 GiB) lies in pieces of 32 tiles graded against the 16 GiB of bucket
            # lines -- and the hit lists equal those of the one-buffer CSR kernel below
            cp = dev.chain_placement()
            assert cp["pieces"] in (5, 6) and cp["tiles_per_piece"] == 32 and cp["handed_back"] == cp["graded"] - cp["pieces"] >= 0
            assert cp["best_grade_G_per_s"] >= cp["worst_kept_grade_G_per_s"] > 0 and not cp["from_reserved_group"]
    assert dev.chain_placement()["pieces"] == 0                  # the per-giant kernel took one buffer
    # launches of 72 tiles: three pieces, the last one a quarter used, two launches for the 144 tiles -- the same hit list
    dev.upload_htgpu_device(img.data_ptr(), 1 << htsz, w, pybsgs.TABLE_LINES64)
    dev.set_tiles_per_launch(72)
    hits40, n40, _ = dev.run(centres, 65536)
    assert dev.chain_placement()["pieces"] == 3 and hits40 == res[pybsgs.TABLE_LINES64]
    dev.set_tiles_per_launch(0)
    assert res[pybsgs.TABLE_LINES64] == res[pybsgs.TABLE_CSR]
    got = res[pybsgs.TABLE_LINES64]
    assert (5, 1, 776) in got and (143, 2, maxnonce - 1) in got
    for t in edge_tiles:
        assert (t, 1, t + 99) in got, t
    assert 2 + len(edge_tiles) <= len(got) <= 30 + len(edge_tiles)
    dev.close()
