#!/usr/bin/env python3
"""tools/experiments/make_experiment_sources.py <csrc dir> -- turn a COPY of bsgs-cuda_amd/csrc into the experiment sources: the timing experiments of rounds 3 and 4 (the
*_CEILING switches -- libraries that return WRONG results and keep the timing --, BSGS_FULL_X, BSGS_INV_PER_WAVE, the slice gate) are written back into the tile kernel of
that copy.  The shipped kernel carries none of them.  Every edit is anchored on a short unique line of the shipped source, so the experiments survive
changes to the kernel that a line-numbered patch does not; `--patch` prints the result as a unified diff (tools/experiments/tile_kernel_experiments.patch is that output,
kept for reading; tests/test_abi.py checks that it applies and equals what this script produces).  tools/experiments/build_experiment.sh calls this on its copy.
"""
import difflib
import os
import sys


def edit(text, old, new, count=1, where=""):
    assert text.count(old) == count, "%s: anchor found %d times, expected %d: %r" % (where, text.count(old), count, old[:90])
    return text.replace(old, new)


def kernel(s):
    w = "giant_kernel.hip.h"
    guard_a = s.index("// ---- compile-time switches ---")
    guard_b = s.index("#define BSGS_STR2(x) #x")
    s = s[:guard_a] + '''// ---- compile-time switches --------------------------------------------------------------------------------------------------
// PATCHED COPY (tools/experiments/make_experiment_sources.py): never shipped.
// The *_CEILING switches build a library that returns WRONG results and keeps the timing ("what would it be worth if ...":
// tools/experiments/README.md); they compile only together with -DBSGS_EXPERIMENT, and bsgs_build_info() names every switch a
// library was built with (tests/test_abi.py requires the shipped one to report none).
#if (defined(BSGS_NO_OVF_CEILING) || defined(BSGS_QUAD_CEILING) || defined(BSGS_NOCHAIN_CEILING) || defined(BSGS_NOCHAIN_STORE_CEILING) || \\
     defined(BSGS_NOCHAIN_LOAD_CEILING) || defined(BSGS_OCT_CEILING) || defined(BSGS_G2_DUP_CEILING) || defined(BSGS_G2_CACHED_CEILING)) && !defined(BSGS_EXPERIMENT)
#error "a *_CEILING switch builds a library that returns wrong results: add -DBSGS_EXPERIMENT (never ship it)"
#endif
''' + s[guard_b:]
    s = edit(s, '''#define BSGS_HIT_WALK_STATUS 4''', '''#ifndef BSGS_SLICE_GATE
#define BSGS_SLICE_GATE 0                 /* experiment (exact results): rows a block may run AHEAD of the slowest running block of its (chunk, slice) group -- the 64 blocks that walk one
                                             slice of the giants for the tiles of a chunk on one XCD; 0 = no gate.  See giant_pair2_kernel and DESIGN.md 4 "Round 4" */
#endif
#define BSGS_GATE_DONE 0xFFFFFFFFu
#define BSGS_HIT_WALK_STATUS 4''', where=w)
    s = edit(s, '''    u32x4 *chain_piece[BSGS_CHAIN_PIECES_MAX];
''', '''    u32x4 *chain_piece[BSGS_CHAIN_PIECES_MAX];
    u32 *gate;             // BSGS_SLICE_GATE builds: one progress word per block, [xcd][slot], zeroed before the launch (NULL: no gate)
''', where=w)
    s = edit(s, '''    if (!A.csr) slow &= !m & (xhi >= bound) & (((hdr >> ovf_fingerprint_index(xhi)) & (BK ? hdr >> ovf_fingerprint_index2(xhi) : 1u) & 1u) != 0);
''', '''    if (!A.csr) slow &= !m & (xhi >= bound) & (((hdr >> ovf_fingerprint_index(xhi)) & (BK ? hdr >> ovf_fingerprint_index2(xhi) : 1u) & 1u) != 0);
#ifdef BSGS_NO_OVF_CEILING      /* -D switch, experiments only: never search the overflow set (results WRONG for 0.26 % of the probes): what the remaining slow path costs */
    if (!A.csr) slow = false;
#endif
''', where=w)
    a = s.index("giant_pair2_kernel(const TileArgs A)")
    head, k = s[:a], s[a:]
    k = edit(k, '''    u32 tb, tile;
    if ((nb & 7u) == 0) {''', '''    u32 tb, tile;
#if BSGS_SLICE_GATE
    u32 *gate_group = nullptr;
    u32 gate_me = 0, gate_width = 0;
#endif
    if ((nb & 7u) == 0) {''', where=w)
    k = edit(k, '''        tb = (r / width) * 8u + xcd;
''', '''        tb = (r / width) * 8u + xcd;
#if BSGS_SLICE_GATE
        if (A.gate) {
            const u32 nslots = gridDim.x >> 3;
            gate_group = A.gate + (u64)xcd * nslots + (slot - r % width);      // the progress words of this block's group: `width` consecutive words
            gate_me = r % width; gate_width = width;
        }
#endif
''', where=w)
    k = edit(k, '''    const u32 lane = threadIdx.x & 63;
''', '''    const u32 lane = threadIdx.x & 63;
#if BSGS_SLICE_GATE
    // publish this block's progress (rows of giants done, phase 1 then phase 3: 1 .. 2p) and wait while it is more than BSGS_SLICE_GATE rows ahead of the slowest
    // block of the group that is running (started, not finished, not hopelessly behind).  Nobody waits for a block that waits: the slowest never does.
    auto gate_step = [&](u32 progress) {
        if (!gate_group) return;
        if (threadIdx.x == 0) __hip_atomic_store(gate_group + gate_me, progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            u32 v = lane < gate_width ? __hip_atomic_load(gate_group + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : BSGS_GATE_DONE;
            if (v == 0u || (v < progress && progress - v > 8u * BSGS_SLICE_GATE)) v = BSGS_GATE_DONE;      // not started / out of reach: not waited for
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const u32 w = __shfl_xor(v, o); v = w < v ? w : v; }
            if (v == BSGS_GATE_DONE || progress <= v + BSGS_SLICE_GATE) break;
            __builtin_amdgcn_s_sleep(32);
        }
    };
#else
    auto gate_step = [&](u32) {};
#endif
''', where=w)
    k = edit(k, '''    const u32 TG = T;                                          // stride of the giants' [slot][4][thread] arrays, in 16-byte elements
''', '''#ifdef BSGS_G2_CACHED_CEILING     /* -D switch, experiments only (results WRONG): with bit 31 of TileArgs::debug_flags set every read of a giant hits the thread's own 16 bytes of slot 0 -- one cached
                                     KiB per wave -- so the giants cost their load instructions and nothing in HBM: what the whole G2 stream is worth (VERDICT r03 item 3) */
    const u32 TG = (A.debug_flags & 0x80000000u) ? 0u : T;
    // ... and because a thread that adds the SAME giant 1024 times probes the same two lines 1024 times (a first attempt at this ceiling did, and took the
    // probe stream out of HBM as well: +6.7 %, profiles/r06f_*), the coordinates handed to the probe arithmetic are made to differ per giant again
#define BSGS_G2_VARY(gx, gy, j) do { (gx).v[0] += (j) * 0x9E3779B9u; (gx).v[3] ^= (j) * 0x85EBCA6Bu; (gy).v[1] += (j) * 0xC2B2AE35u; (gy).v[4] ^= (j) * 0x27D4EB2Fu; } while (0)
#else
    const u32 TG = T;
#define BSGS_G2_VARY(gx, gy, j) do { } while (0)
#endif
''', where=w)
    k = edit(k, '''            const bool store_now = QUAD ? (j & 3u) == 3u : (j & 1u) != 0;
''', '''#ifdef BSGS_QUAD_CEILING     /* -D switch, experiments only: speed ceiling of "one stored product per FOUR giants" (results WRONG: the odd pairs use a stale product) */
            const bool store_now = (j & 3u) == 3u;
#elif defined(BSGS_NOCHAIN_CEILING) || defined(BSGS_NOCHAIN_STORE_CEILING)   /* -D switches, experiments only: no chain stores (and, _NOCHAIN_, no fetches): results WRONG */
            const bool store_now = false;
#else
#ifdef BSGS_OCT_CEILING
            const bool store_now = QUAD ? (j & 7u) == 7u : (j & 1u) != 0;
#else
            const bool store_now = QUAD ? (j & 3u) == 3u : (j & 1u) != 0;
#endif
#endif
''', where=w)
    k = edit(k, '''            if (store_now && j + 1 < p && live) CHAIN_STORE(chain + ((u64)((j + 1) >> GSH) * 2 + 0) * CS, chain + ((u64)((j + 1) >> GSH) * 2 + 1) * CS, acc);
''', '''            if (store_now && j + 1 < p && live) CHAIN_STORE(chain + ((u64)((j + 1) >> GSH) * 2 + 0) * CS, chain + ((u64)((j + 1) >> GSH) * 2 + 1) * CS, acc);
            if (BSGS_SLICE_GATE && (j & 15u) == 15u) gate_step(j + 1u);
''', where=w)
    k = edit(k, '''    fe inv;
    {
        const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);''', '''    fe inv;
#ifdef BSGS_INV_PER_WAVE                                           /* A-B only: one Fermat inversion per wave, as before */
    fe_inv(inv, acc);
    if (false)
#endif
    {
        const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);''', where=w)
    k = edit(k, '''        km = x_key_from_lambda(lam, nPx, gx, cad);
''', '''#ifdef BSGS_FULL_X
        { fe xm; x_from_lambda(xm, lam, nPx, gx); km = ((u64)xm.v[1] << 32) | xm.v[0]; }
#else
        km = x_key_from_lambda(lam, nPx, gx, cad);
#endif
''', where=w)
    k = edit(k, '''            kp = x_key_from_lambda(lam, nPx, gx, cad);
''', '''#ifdef BSGS_FULL_X
            { fe xp; x_from_lambda(xp, lam, nPx, gx); kp = ((u64)xp.v[1] << 32) | xp.v[0]; }
#else
            kp = x_key_from_lambda(lam, nPx, gx, cad);
#endif
''', where=w)
    # the second reads of Gx in the quad chain (giants a, b through the temporaries; c in registers) served from one cached KiB
    k = edit(k, '''        auto dma_gx = [&](u32 j, char *wave_dst) {                                      // p - Gx of giant j -> an LDS temporary (lane l: bytes [16 l, 16 l + 16) of each half)
''', '''        auto dma_gx = [&](u32 j, char *wave_dst) {                                      // p - Gx of giant j -> an LDS temporary (lane l: bytes [16 l, 16 l + 16) of each half)
#ifdef BSGS_G2_DUP_CEILING                                                              /* timing experiment only (tools/experiments/README.md): the SECOND reads of Gx (a, b here; c below) hit one cached KiB */
            j = 0;
#endif
''', where=w)
    k = edit(k, '''            if (Q > 0) stash_fetch(Q);
''', '''#ifdef BSGS_OCT_CEILING        /* -D switch, experiments only: speed ceiling of "one stored product per EIGHT giants" (results WRONG: odd quads use a stale product) */
            if (Q > 0 && !(Q & 1u)) stash_fetch(Q);
#elif defined(BSGS_NOCHAIN_CEILING) || defined(BSGS_NOCHAIN_LOAD_CEILING)   /* no chain fetches at all (results WRONG): with _NOCHAIN_ also no stores -- what the chain streams cost, and their share of FETCH_SIZE */
#else
            if (Q > 0) stash_fetch(Q);
#endif
''', where=w)
    k = edit(k, '''            fe_load2(q2, g2 + ((u64)(ja + 2) * 4 + 0) * TG, g2 + ((u64)(ja + 2) * 4 + 1) * TG);       // Gx_c
''', '''#ifdef BSGS_G2_DUP_CEILING
            fe_load2(q2, g2, g2 + T);
#else
            fe_load2(q2, g2 + ((u64)(ja + 2) * 4 + 0) * TG, g2 + ((u64)(ja + 2) * 4 + 1) * TG);       // Gx_c
#endif
''', where=w)
    k = edit(k, '''            const u32 Q = nq - 1 - QQ, ja = 4 * Q, jb = ja + 1, jc = ja + 2, jd = ja + 3;
''', '''            const u32 Q = nq - 1 - QQ, ja = 4 * Q, jb = ja + 1, jc = ja + 2, jd = ja + 3;
            if (BSGS_SLICE_GATE && (QQ & 3u) == 0u && QQ) gate_step(p + 4u * QQ);
''', where=w)
    for g in "dcba":
        k = edit(k, "                giant(gx%s, gy%s, s%s, eq%s, tid * p + j%s, [&]() {" % (g, g, g, g, g),
                 "                BSGS_G2_VARY(gx%s, gy%s, j%s);\n                giant(gx%s, gy%s, s%s, eq%s, tid * p + j%s, [&]() {" % (g, g, g, g, g, g, g, g), where=w)
    k = edit(k, '''                    if (Q2 > 0) stash_fetch(Q2);
''', '''#ifdef BSGS_OCT_CEILING
                    if (Q2 > 0 && !(Q2 & 1u)) stash_fetch(Q2);
#elif defined(BSGS_NOCHAIN_CEILING) || defined(BSGS_NOCHAIN_LOAD_CEILING)
#else
                    if (Q2 > 0) stash_fetch(Q2);
#endif
''', where=w)
    k = edit(k, '''                    fe_load2(q2, g2 + ((u64)(ja2 + 2) * 4 + 0) * TG, g2 + ((u64)(ja2 + 2) * 4 + 1) * TG);
''', '''#ifdef BSGS_G2_DUP_CEILING
                    fe_load2(q2, g2, g2 + T);
#else
                    fe_load2(q2, g2 + ((u64)(ja2 + 2) * 4 + 0) * TG, g2 + ((u64)(ja2 + 2) * 4 + 1) * TG);
#endif
''', where=w)
    # pair chain
    k = edit(k, '''        if (m > 0) stash_fetch(m);                         // older than the loads below: it has landed when they have
''', '''#ifdef BSGS_QUAD_CEILING
        if (m > 0 && !(m & 1u)) stash_fetch(m);
#elif defined(BSGS_NOCHAIN_CEILING) || defined(BSGS_NOCHAIN_LOAD_CEILING)
#else
        if (m > 0) stash_fetch(m);                         // older than the loads below: it has landed when they have
#endif
''', where=w)
    k = edit(k, '''                if (m > 1) stash_fetch(m - 1);                         // S of the pair below: first used one giant from now
''', '''#ifdef BSGS_QUAD_CEILING
                if (m > 1 && !((m - 1) & 1u)) stash_fetch(m - 1);
#elif defined(BSGS_NOCHAIN_CEILING) || defined(BSGS_NOCHAIN_LOAD_CEILING)
#else
                if (m > 1) stash_fetch(m - 1);                         // S of the pair below: first used one giant from now
#endif
''', where=w)
    for g in "ba":                                                  # (the newline in the anchor: the quad chain's calls are these very lines, indented deeper)
        k = edit(k, "\n            giant(gx%s, gy%s, s%s, eq%s, tid * p + j%s, [&]() {" % (g, g, g, g, g),
                 "\n            BSGS_G2_VARY(gx%s, gy%s, j%s);\n            giant(gx%s, gy%s, s%s, eq%s, tid * p + j%s, [&]() {" % (g, g, g, g, g, g, g, g), where=w)
    k = edit(k, '''    if (PHASE_PROBE && want_digest && live) {''', '''#if BSGS_SLICE_GATE
    if (gate_group && threadIdx.x == 0) __hip_atomic_store(gate_group + gate_me, BSGS_GATE_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (PHASE_PROBE && want_digest && live) {''', where=w)
    return head + k


def engine(s):
    w = "bsgs_hip.hip"
    s = edit(s, '''#ifdef BSGS_SLICE_GATE
        add("BSGS_SLICE_GATE=" BSGS_STR(BSGS_SLICE_GATE));
#endif
''', '''        if (BSGS_SLICE_GATE != 0) add("BSGS_SLICE_GATE=" BSGS_STR(BSGS_SLICE_GATE));
''', where=w)
    s = edit(s, '''    if (d->digest) (void)hipFree(d->digest);
''', '''    if (d->digest) (void)hipFree(d->digest);
    if (d->gate) (void)hipFree(d->gate);
''', where=w)
    s = edit(s, '''    A.debug_flags = d->debug_flags; A.bucket_mul = d->bucket_mul;
''', '''    A.debug_flags = d->debug_flags; A.bucket_mul = d->bucket_mul;
#ifdef BSGS_G2_CACHED_CEILING
    A.debug_flags |= 0x80000000u;        // experiment build only (results WRONG): every giant read hits one cached KiB per wave
#endif
    A.gate = nullptr;
#if BSGS_SLICE_GATE
    if (chain_group(d, pi) == 4 && !d->debug_flags && !d->phase_probe) {      // experiment build (exact results): progress words of the slice gate, zeroed per launch
        const size_t words = (size_t)(((Ti + tile_block(d) - 1) / tile_block(d)) * ntiles);
        if (d->gate_words < words) { if (d->gate) (void)hipFree(d->gate); d->gate = nullptr; HIPCHK(hipMalloc(&d->gate, words * 4)); d->gate_words = words; }
        HIPCHK(hipMemsetAsync(d->gate, 0, words * 4, st));
        A.gate = d->gate;
    }
#endif
''', where=w)
    return s


def internal(s):
    return edit(s, '''    u64 *digest = nullptr;                 // bsgs_run_digest: [tile][Ti][2]
''', '''    u32 *gate = nullptr;                   // BSGS_SLICE_GATE builds only: progress words of the slice gate
    size_t gate_words = 0;
    u64 *digest = nullptr;                 // bsgs_run_digest: [tile][Ti][2]
''', where="bsgs_internal.h")


FILES = (("giant_kernel.hip.h", kernel), ("bsgs_hip.hip", engine), ("bsgs_internal.h", internal))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    csrc = args[0]
    as_patch = "--patch" in sys.argv
    out = []
    for name, fn in FILES:
        path = os.path.join(csrc, name)
        old = open(path).read()
        new = fn(old)
        if as_patch:
            out += list(difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/csrc/" + name, "b/csrc/" + name))
        else:
            open(path, "w").write(new)
    if as_patch:
        sys.stdout.write("".join(out))


if __name__ == "__main__":
    main()
