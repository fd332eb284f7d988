// microbench.hip -- roofline denominators for the giant-step kernel on MI355X (gfx950).
//   (1) instruction issue rates that bound 256-bit modular multiplication:
//       v_mad_u64_u32, v_mul_lo/hi_u32, v_add_co/v_addc carry chain, v_fma_f64, v_mad_u32_u24
//   (2) random-read bandwidth of HBM at a given footprint and granule (GUPS style): the
//       denominator SURVEY.md 8(d) asks for (neither guide gives one).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 microbench.hip -o microbench
// Run  : ./microbench [footprint_MiB ...]
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#include "fp256.hip.h"       // the engine's own field arithmetic: fe_mul (Comba columns on v_mad_u64_u32 + fast fold), fe_mul512

// ---------------------------------------------------------------- instruction rates
template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(u32 *out, int iters, u32 seed)
{
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u32 a = seed * 2654435761u + t, b = a ^ 0x9E3779B9u;
    u64 acc0 = a, acc1 = b, acc2 = a + 7, acc3 = b + 9;
    u32 x0 = a, x1 = b, x2 = a + 3, x3 = b + 5;
    double d0 = (double)a, d1 = (double)b, d2 = 1.5, d3 = 2.5, dm = 1.0000001, da = 0.5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) {        // v_mad_u64_u32, 4 independent accumulators
                asm volatile("v_mad_u64_u32 %0, s[6:7], %4, %5, %0\n\t"
                             "v_mad_u64_u32 %1, s[6:7], %4, %5, %1\n\t"
                             "v_mad_u64_u32 %2, s[6:7], %4, %5, %2\n\t"
                             "v_mad_u64_u32 %3, s[6:7], %4, %5, %3"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "s6", "s7");
            } else if (OP == 1) { // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\t"
                             "v_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if (OP == 2) { // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\t"
                             "v_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if (OP == 3) { // plain v_add_u32 (full-rate reference)
                asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\t"
                             "v_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if (OP == 4) { // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\t"
                             "v_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(da));
            } else if (OP == 5) { // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n\tv_mad_u32_u24 %1, %1, %4, %5\n\t"
                             "v_mad_u32_u24 %2, %2, %4, %5\n\tv_mad_u32_u24 %3, %3, %4, %5"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            } else if (OP == 6) { // carry chain: add_co + addc with the 2 wait states gfx950 needs
                asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\t"
                             "s_nop 1\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc\n\ts_nop 1\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
            } else if (OP == 7) { // v_lshl_add_u64 (64-bit add, no carry out)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\t"
                             "v_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(acc0));
            } else if (OP == 8) { // v_mul_hi_u32_u24
                asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\t"
                             "v_mul_hi_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            } else if (OP == 9) { // mad + dependent addc with vcc, spaced by 2 independent mads (pipelined carries)
                asm volatile("v_mad_u64_u32 %0, s[6:7], %4, %5, %0\n\t"
                             "v_mad_u64_u32 %1, s[8:9], %4, %5, %1\n\t"
                             "v_mad_u64_u32 %2, s[10:11], %4, %5, %2\n\t"
                             "v_addc_co_u32 %3, s[12:13], 0, %3, s[6:7]\n\t"
                             "v_addc_co_u32 %3, s[12:13], 0, %3, s[8:9]\n\t"
                             "v_addc_co_u32 %3, s[12:13], 0, %3, s[10:11]\n\t"
                             "v_mad_u64_u32 %0, s[6:7], %4, %5, %0"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(x3) : "v"(a), "v"(b)
                             : "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13");
            }
        }
    }
    u32 r = (u32)acc0 ^ (u32)acc1 ^ (u32)acc2 ^ (u32)acc3 ^ (u32)(acc0 >> 32) ^ x0 ^ x1 ^ x2 ^ x3 ^
            (u32)d0 ^ (u32)d1 ^ (u32)d2 ^ (u32)d3;
    if (r == 0x12345678u) out[t] = r;   // keep everything live
}

template <int OP>
static double run_rate(const char *name, int per_iter, u32 *dout)
{
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, 10, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, iters, 2u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)blocks * threads * iters * 16.0 * per_iter;
    double gops = n / (ms * 1e-3) / 1e9;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: 256 CU * 4 SIMD
    double wave_instr_per_s = gops * 1e9 / 64.0;
    double cyc = 2.4e9 * 256 * 4 / wave_instr_per_s;
    printf("{\"bench\":\"rate\",\"op\":\"%s\",\"Gops_per_s\":%.1f,\"cycles_per_wave_instr_per_simd_at_2.4GHz\":%.2f,\"ms\":%.3f}\n",
           name, gops, cyc, ms);
    return gops;
}

// ---------------------------------------------------------------- random reads (GUPS)
__device__ __forceinline__ u64 splitmix(u64 &s)
{
    s += 0x9E3779B97F4A7C15ULL;
    u64 z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// each lane reads GRAN bytes (16*NV) at a random GRAN-aligned offset, UNROLL independent loads in flight
template <int NV, int UNROLL>
__global__ void __launch_bounds__(256) gups_kernel(const u32x4 *__restrict__ buf, u64 n_gran, int iters, u32 *out, u64 seed)
{
    u64 s = seed + (u64)(threadIdx.x + blockIdx.x * blockDim.x) * 0x632BE59BD9B4E019ULL;
    u32 acc = 0;
    for (int i = 0; i < iters; i++) {
        u32x4 v[UNROLL][NV];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            u64 g = splitmix(s) % n_gran;     // n_gran is a power of two in practice
            const u32x4 *p = buf + g * NV;
#pragma unroll
            for (int k = 0; k < NV; k++) v[u][k] = p[k];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int k = 0; k < NV; k++) acc ^= v[u][k].x ^ v[u][k].y ^ v[u][k].z ^ v[u][k].w;
    }
    if (acc == 0x9abcdef1u) out[0] = acc;
}

// cooperative: a group of (GRAN/16) lanes reads one granule, each lane 16 B (coalesced inside the granule)
template <int LANES_PER, int UNROLL>
__global__ void __launch_bounds__(256) gups_coop_kernel(const u32x4 *__restrict__ buf, u64 n_gran, int iters, u32 *out, u64 seed)
{
    u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
    u64 s = seed + (u64)(tid / LANES_PER) * 0x632BE59BD9B4E019ULL;
    u32 sub = tid % LANES_PER;
    u32 acc = 0;
    for (int i = 0; i < iters; i++) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            u64 g = splitmix(s) % n_gran;
            v[u] = buf[g * LANES_PER + sub];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9abcdef1u) out[0] = acc;
}

template <int NV, int UNROLL>
static void run_gups(const u32x4 *buf, size_t bytes, u32 *dout, int waves_per_simd)
{
    const int threads = 256, blocks = 256 * waves_per_simd;   // waves/SIMD = blocks*4/(256*4)
    const int iters = 256;
    u64 n_gran = bytes / (16 * NV);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gups_kernel<NV, UNROLL>), dim3(blocks), dim3(threads), 0, 0, buf, n_gran, 4, dout, 1ULL);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gups_kernel<NV, UNROLL>), dim3(blocks), dim3(threads), 0, 0, buf, n_gran, iters, dout, 99ULL);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double reads = (double)blocks * threads * iters * UNROLL;
    printf("{\"bench\":\"gups\",\"mode\":\"lane\",\"footprint_MiB\":%zu,\"granule_B\":%d,\"in_flight_per_lane\":%d,\"waves_per_simd\":%d,"
           "\"Greads_per_s\":%.2f,\"GBps\":%.1f,\"ms\":%.3f}\n",
           bytes >> 20, 16 * NV, UNROLL, waves_per_simd, reads / (ms * 1e-3) / 1e9, reads * 16 * NV / (ms * 1e-3) / 1e9, ms);
}

template <int LANES_PER, int UNROLL>
static void run_gups_coop(const u32x4 *buf, size_t bytes, u32 *dout, int waves_per_simd)
{
    const int threads = 256, blocks = 256 * waves_per_simd;
    const int iters = 256;
    u64 n_gran = bytes / (16 * LANES_PER);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gups_coop_kernel<LANES_PER, UNROLL>), dim3(blocks), dim3(threads), 0, 0, buf, n_gran, 4, dout, 1ULL);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gups_coop_kernel<LANES_PER, UNROLL>), dim3(blocks), dim3(threads), 0, 0, buf, n_gran, iters, dout, 99ULL);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double reads = (double)blocks * threads * iters * UNROLL / LANES_PER;
    printf("{\"bench\":\"gups\",\"mode\":\"coop\",\"footprint_MiB\":%zu,\"granule_B\":%d,\"in_flight_per_lane\":%d,\"waves_per_simd\":%d,"
           "\"Greads_per_s\":%.2f,\"GBps\":%.1f,\"ms\":%.3f}\n",
           bytes >> 20, 16 * LANES_PER, UNROLL, waves_per_simd, reads / (ms * 1e-3) / 1e9,
           reads * 16 * LANES_PER / (ms * 1e-3) / 1e9, ms);
}

__global__ void fill_kernel(uint4 *buf, size_t n)
{
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) buf[i] = make_uint4((u32)i, (u32)(i >> 32), (u32)i * 2654435761u, 7u);
}

// "power" mode: keep ONE instruction mix running for `secs` seconds so that tools/power_ops.sh can sample rocm-smi next
// to it; prints the sustained rate (the clock the chip settles at under that mix is part of the answer)
template <int OP>
static void sustain(const char *name, int per_iter, double secs, u32 *dout)
{
    const int blocks = 256 * 8, threads = 256, iters = 40000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_ms = 0, n = 0;
    while (total_ms < secs * 1e3) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, iters, 2u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms; n += (double)blocks * threads * iters * 16.0 * per_iter;
    }
    printf("{\"bench\":\"sustain\",\"op\":\"%s\",\"seconds\":%.2f,\"Gops_per_s\":%.1f}\n", name, total_ms / 1e3, n / (total_ms * 1e-3) / 1e9);
}

// sustained cooperative random reads (4 lanes x 16 B = one 64-byte line, or 8 x 16 = 128 B) over `mib` MiB
template <int LANES_PER>
static void sustain_gups(double secs, size_t mib, u32 *dout)
{
    const size_t bytes = mib << 20;
    uint4 *buf; CK(hipMalloc(&buf, bytes));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 16);
    CK(hipDeviceSynchronize());
    const int threads = 256, blocks = 256 * 8, iters = 4096;
    const u64 n_gran = bytes / (16 * LANES_PER);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_ms = 0, reads = 0;
    u64 seed = 7;
    while (total_ms < secs * 1e3) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gups_coop_kernel<LANES_PER, 8>), dim3(blocks), dim3(threads), 0, 0, (const u32x4 *)buf, n_gran, iters, dout, seed++);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms; reads += (double)blocks * threads * iters * 8 / LANES_PER;
    }
    printf("{\"bench\":\"sustain\",\"op\":\"random %d-byte reads over %zu MiB\",\"seconds\":%.2f,\"Greads_per_s\":%.2f}\n", 16 * LANES_PER, mib,
           total_ms / 1e3, reads / (total_ms * 1e-3) / 1e9);
}

// sustained coalesced 16-byte-per-lane reads (a wave reads 1 KiB contiguous) over `mib` MiB
__global__ void __launch_bounds__(256) stream_kernel(const u32x4 *__restrict__ buf, u64 n16, int iters, u32 *out)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const u32x4 v = buf[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
            i += stride;
            if (i >= n16) i -= n16;
        }
    }
    if (acc == 0x9abcdef1u) out[0] = acc;
}
static void sustain_stream(double secs, size_t mib, u32 *dout)
{
    const size_t bytes = mib << 20;
    uint4 *buf; CK(hipMalloc(&buf, bytes));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 16);
    CK(hipDeviceSynchronize());
    const int threads = 256, blocks = 256 * 8, iters = 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_ms = 0, rd = 0;
    while (total_ms < secs * 1e3) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(threads), 0, 0, (const u32x4 *)buf, (u64)(bytes / 16), iters, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms; rd += (double)blocks * threads * iters * 8 * 16;
    }
    printf("{\"bench\":\"sustain\",\"op\":\"coalesced reads over %zu MiB\",\"seconds\":%.2f,\"GB_per_s\":%.1f}\n", mib, total_ms / 1e3, rd / (total_ms * 1e-3) / 1e9);
}

// ---------------------------------------------------------------- a different multiplier: FP64 fused multiply-adds
// 256-bit operands as six 48-bit limbs held as doubles; every limb product a_i*b_j (96 bits) is split EXACTLY into its part above 2^48
// and its part below by two fused multiply-adds (default rounding; nothing depends on the rounding mode):
//     hi chain : H <- fma(a_i, b_j, H), H starting at 2^100 (ulp 2^48): H - 2^100 = 2^48 * sum round(a_i b_j / 2^48), exactly -- free accumulation
//     each hi  : nhi = H_before - H_after (exact) ; lo = fma(a_i, b_j, nhi) (exact, |lo| <= 2^47: a SIGNED low part) ; L += lo (exact, |L| <= 6 * 2^47)
// 4 FP64 operations per limb product, 36 products, 3 per column to hand the hi sum to the next column: 177 FP64 instructions for the
// 512-bit product in redundant form (12 column values < 2^52) -- BEFORE any reduction mod p, carry normalisation or conversion from / to the
// 32-bit words the giants are stored in.  (5 x 52-bit limbs, the Emmart-Weems form, needs integer 64-bit adds of the bit patterns -- two
// carry steps each on this ISA -- because five 52-bit halves do not sum exactly in a double: 7 instructions per product x 25 = 175, the same.)
// The integer product it competes with (fe_mul512) is 64 multiply-adds + 56 carry counts + 15 moves = 135 instructions, and v_fma_f64 issues
// at the rate of v_mad_u64_u32 (4 cycles per wave) at 29 pJ against 25 (profiles/r01h_power_ops.jsonl).
struct fd6 { double l[6]; };
__device__ __forceinline__ void dpf_mul512(double (&V)[12], const fd6 &a, const fd6 &b)
{
    const double C = 0x1p100;
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        double H = C, L = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int j = k - i;
            if (j < 0 || j > 5) continue;
            const double before = H;
            H = __builtin_fma(a.l[i], b.l[j], H);               // + round(ab / 2^48) * 2^48 (H's ulp is 2^48)
            const double nhi = before - H;
            L += __builtin_fma(a.l[i], b.l[j], nhi);            // ab - round(ab / 2^48) * 2^48: the signed low part, exactly
        }
        V[k] = L + carry;
        carry = (H - C) * 0x1p-48;
    }
    V[11] = carry;
}

// ---------------------------------------------------------------- unsaturated limbs: N limbs of BITS bits, 64-bit column accumulators, NO carry counts
// The engine's multiplier (fe_mul: 8 x 32-bit words) pays one carry count per product because v_mad_u64_u32 has no carry-in: 73 multiply-adds + 72 carry steps + 25
// others, and a carry step costs the issue time of a multiply-add on this chip.  With limbs narrow enough that a whole column of products fits 64 bits -- 9 x 29 bits:
// 9 * 2^58 < 2^61.2; 10 x 26 bits: 10 * 2^52 < 2^55.4 -- the carry counts disappear; what comes instead is MORE products (81 / 100 for 64), a shift + mask + move per
// column to cut the sums back to limbs, and a fold mod p whose constant no longer sits on a limb boundary: 2^(N*BITS) = 2^HI * 2^256 = 2^HI * (2^32 + 977) (mod p) lands
// as  high_limb * (977 << HI)  on limb j and  high_limb << (32 + HI - BITS)  on limb j + 1: two multiply-adds per high limb (a 64-bit shift-add issues no faster).
// unsat<29, 9>: 81 + 18 + 2 multiply-adds, ~90 32-bit operations; unsat<26, 10>: 100 + 20 + 2 and ~100.  Inputs and outputs fully normalised (limbs < 2^BITS, value < 2^256,
// congruent mod p): a lazy form would make field additions cheap but breaks the 9 * 2^58 column bound.  `microbench unsatcheck` compares both with fe_mul on random and
// extreme operands; `microbench power 203|204 <s>` is the sustained rate for tools/power_ops.sh.
template <int BITS, int N>
struct unsat {
    static constexpr int HI = N * BITS - 256;                       // bits of the limb vector beyond 2^256
    static constexpr int TOPBITS = 256 - (N - 1) * BITS;            // width of the top limb of a normalised value
    static constexpr u32 M = (1u << BITS) - 1u, MTOP = (1u << TOPBITS) - 1u;
    static constexpr u32 C1 = 977u << HI;                           // 2^(N BITS) = C1 + 2^BITS * 2^S1  (mod p)
    static constexpr int S1 = 32 + HI - BITS;
    static constexpr int S2 = 32 - BITS;                            // 2^256 = 977 + 2^BITS * 2^S2  (mod p)
    u32 l[N];

    __device__ __forceinline__ void from_fe(const fe &a)
    {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = i * BITS, w = bit >> 5, sh = bit & 31;
            u64 v = a.v[w];
            if (w + 1 < 8) v |= (u64)a.v[w + 1] << 32;
            l[i] = (u32)(v >> sh) & (i == N - 1 ? MTOP : M);
        }
    }
    __device__ __forceinline__ void to_fe(fe &r) const
    {
#pragma unroll
        for (int w = 0; w < 8; w++) r.v[w] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = i * BITS, w = bit >> 5, sh = bit & 31;
            const u64 v = (u64)l[i] << sh;
            r.v[w] |= (u32)v;
            if (w + 1 < 8) r.v[w + 1] |= (u32)(v >> 32);
        }
    }
    // the same product written for the ISA: carries by v_alignbit_b32 (a column sum stays below 2^(32 + BITS), so its carry fits 32 bits), every 64-bit addition as a
    // multiply-add by an opaque 1 (v_mad_u64_u32: 4.2 cycles; a 64-bit add is two carry steps: 8.2), the shifted share of the fold as a multiply-add by an opaque 2^S1
    static __device__ __forceinline__ void mul_tuned(unsat &r, const unsat &a, const unsat &b)
    {
        u32 one = 1u, sh1 = 1u << S1, c1 = C1;
        asm volatile("" : "+v"(one), "+v"(sh1), "+v"(c1));           // opaque: the compiler must not turn the multiply-adds into shifts and 64-bit adds
        u32 c[2 * N];
        u32 cy = 0;
#pragma unroll
        for (int k = 0; k < 2 * N - 1; k++) {
            u64 acc = cy;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) acc = (u64)a.l[i] * b.l[j] + acc;
            }
            c[k] = (u32)acc & M;
            cy = __builtin_amdgcn_alignbit((u32)(acc >> 32), (u32)acc, BITS);
        }
        c[2 * N - 1] = cy;
        u64 t[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            t[j] = (u64)c[N + j] * c1 + c[j];
            if (j > 0) t[j] = (u64)c[N + j - 1] * sh1 + t[j];
        }
        const u32 t9 = c[2 * N - 1] << S1;
        t[0] = (u64)t9 * c1 + t[0];
        t[1] = (u64)t9 * sh1 + t[1];
        u32 cr = 0;
#pragma unroll
        for (int j = 0; j < N - 1; j++) {
            const u64 v = (u64)cr * one + t[j];                      // t[j] < 2^46, cr < 2^17
            r.l[j] = (u32)v & M;
            cr = __builtin_amdgcn_alignbit((u32)(v >> 32), (u32)v, BITS);
        }
        const u64 v = (u64)cr * one + t[N - 1];
        r.l[N - 1] = (u32)v & MTOP;
        u32 top = __builtin_amdgcn_alignbit((u32)(v >> 32), (u32)v, TOPBITS);
        while (__builtin_expect(top != 0, 1)) {                      // top * (2^32 + 977) onto limbs 0 and 1, rippling on (rarely beyond limb 2)
            const u64 w0 = (u64)top * 977u + r.l[0];
            r.l[0] = (u32)w0 & M;
            const u32 w1 = r.l[1] + (top << S2) + (u32)(w0 >> BITS);  // < 2^BITS + 2^(24 + S2) + 2^6: fits 32 bits (BITS + S2 = 32, top < 2^24 only on the first round)
            r.l[1] = w1 & M;
            u32 c2 = w1 >> BITS;
#pragma unroll
            for (int j = 2; j < N - 1; j++) { if (__builtin_expect(c2 == 0, 1)) break; const u32 x = r.l[j] + c2; r.l[j] = x & M; c2 = x >> BITS; }
            const u32 x = r.l[N - 1] + c2;
            r.l[N - 1] = x & MTOP;
            top = x >> TOPBITS;
        }
    }
    // r = a * b mod p, normalised
    static __device__ __forceinline__ void mul(unsat &r, const unsat &a, const unsat &b)
    {
        u32 c[2 * N];                                               // the 2N limbs of the product
        u64 acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) acc += (u64)a.l[i] * b.l[j];    // one v_mad_u64_u32 each: the column (plus the carry it started from) fits 64 bits
            }
            c[k] = (u32)acc & M;
            acc >>= BITS;
        }
        c[2 * N - 1] = (u32)acc;                                    // < 2^(2 TOPBITS - BITS + 1)
        // fold the N high limbs: t[j] = c[j] + C1 h[j] + (h[j-1] << S1), h = c[N ..]; the last high limb's shifted share lands on limb N: t9
        u64 t[N + 1];
#pragma unroll
        for (int j = 0; j < N; j++) {
            t[j] = (u64)c[N + j] * C1 + c[j];
            if (j > 0) t[j] += (u64)c[N + j - 1] << S1;
        }
        const u32 t9 = c[2 * N - 1] << S1;                          // limb N once more (tiny): folded below
        t[0] += (u64)t9 * C1;
        t[1] += (u64)t9 << S1;
        // cut back to limbs; what leaves the top limb is a multiple of 2^256 = 977 + (2^S2 << BITS)
        u64 cr = 0;
#pragma unroll
        for (int j = 0; j < N - 1; j++) { const u64 v = t[j] + cr; r.l[j] = (u32)v & M; cr = v >> BITS; }
        u64 v = t[N - 1] + cr;
        r.l[N - 1] = (u32)v & MTOP;
        u64 top = v >> TOPBITS;                                     // < 2^24
        // top * (2^32 + 977) onto limbs 0 and 1, rippling on (each round's `top` is a single bit at most after the first)
        while (__builtin_expect(top != 0, 1)) {
            u64 w0 = (u64)r.l[0] + top * 977u, w1;
            r.l[0] = (u32)w0 & M;
            w1 = (u64)r.l[1] + (top << S2) + (w0 >> BITS);
            r.l[1] = (u32)w1 & M;
            u64 c2 = w1 >> BITS;
#pragma unroll
            for (int j = 2; j < N - 1; j++) { if (__builtin_expect(c2 == 0, 1)) break; const u64 x = (u64)r.l[j] + c2; r.l[j] = (u32)x & M; c2 = x >> BITS; }
            const u64 x = (u64)r.l[N - 1] + c2;
            r.l[N - 1] = (u32)x & MTOP;
            top = x >> TOPBITS;
        }
    }
};

// OP 200: fe_mul (integer, with the fold) ; 201: fe_mul512 only (integer product, no fold) ; 202: dpf_mul512 (FP64 product, no fold)
template <int OP>
__global__ void __launch_bounds__(256) mulrate_kernel(u32 *out, int iters, u32 seed)
{
    const u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u32 r = 0;
    if (OP == 202) {
        fd6 a, b;
#pragma unroll
        for (int i = 0; i < 6; i++) { a.l[i] = (double)(((u64)(seed * 2654435761u + t * 40503u + i) << 16) | 0x1234u); b.l[i] = a.l[i] + 97.0; }
        for (int it = 0; it < iters; it++) {
            double V[12];
            dpf_mul512(V, a, b);
#pragma unroll
            for (int i = 0; i < 6; i++) a.l[i] = V[i] + V[i + 6];        // (6 extra additions per product: a dependency, not a reduction)
            dpf_mul512(V, b, a);
#pragma unroll
            for (int i = 0; i < 6; i++) b.l[i] = V[i] + V[i + 6];
        }
        r = (u32)a.l[0] ^ (u32)b.l[3];
    } else if (OP == 203 || OP == 204 || OP == 205) {
        fe a0, b0;
#pragma unroll
        for (int i = 0; i < 8; i++) { a0.v[i] = seed * 2654435761u + t * 40503u + i; b0.v[i] = a0.v[i] ^ 0x9E3779B9u; }
        if (OP == 203) {
            unsat<29, 9> a, b; a.from_fe(a0); b.from_fe(b0);
            for (int it = 0; it < iters; it++) { unsat<29, 9>::mul_tuned(a, a, b); unsat<29, 9>::mul_tuned(b, b, a); }
            r = (a.l[0] ^ b.l[3]) + 0x12000000u;              // (the sink compares with 0x12345678: it must stay REACHABLE for limbs below 2^29 / 2^26, or the loop is dead code)
        } else if (OP == 205) {
            unsat<29, 9> a, b; a.from_fe(a0); b.from_fe(b0);
            for (int it = 0; it < iters; it++) { unsat<29, 9>::mul(a, a, b); unsat<29, 9>::mul(b, b, a); }
            r = (a.l[0] ^ b.l[3]) + 0x12000000u;
        } else {
            unsat<26, 10> a, b; a.from_fe(a0); b.from_fe(b0);
            for (int it = 0; it < iters; it++) { unsat<26, 10>::mul_tuned(a, a, b); unsat<26, 10>::mul_tuned(b, b, a); }
            r = (a.l[0] ^ b.l[3]) + 0x12000000u;
        }
    } else {
        fe a, b;
#pragma unroll
        for (int i = 0; i < 8; i++) { a.v[i] = seed * 2654435761u + t * 40503u + i; b.v[i] = a.v[i] ^ 0x9E3779B9u; }
        for (int it = 0; it < iters; it++) {
            if (OP == 200) { fe_mul(a, a, b); fe_mul(b, b, a); }
            else if (OP == 207) {                                          // the engine's multiplier with the carries of the product counted on the scalar unit
                u32 w[16];
                fe_mul512_s(w, a.v, b.v); fe_reduce512(a, w);
                fe_mul512_s(w, b.v, a.v); fe_reduce512(b, w);
            } else if (OP == 206) {
                u32 w[16];
                fe_mul512_s(w, a.v, b.v);
#pragma unroll
                for (int i = 0; i < 8; i++) a.v[i] = w[i] ^ w[i + 8];
                fe_mul512_s(w, b.v, a.v);
#pragma unroll
                for (int i = 0; i < 8; i++) b.v[i] = w[i] ^ w[i + 8];
            } else {
                u32 w[16];
                fe_mul512(w, a.v, b.v);
#pragma unroll
                for (int i = 0; i < 8; i++) a.v[i] = w[i] ^ w[i + 8];     // (8 extra xors per product)
                fe_mul512(w, b.v, a.v);
#pragma unroll
                for (int i = 0; i < 8; i++) b.v[i] = w[i] ^ w[i + 8];
            }
        }
        r = a.v[0] ^ b.v[3];
    }
    if (r == 0x12345678u) out[t] = r;
}
template <int OP>
static void sustain_mul(const char *name, double secs, u32 *dout)
{
    const int blocks = 256 * 8, threads = 256, iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_ms = 0, n = 0;
    hipLaunchKernelGGL(mulrate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, 10, 1u);
    CK(hipDeviceSynchronize());
    while (total_ms < secs * 1e3) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mulrate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, iters, 2u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms; n += (double)blocks * threads * iters * 2.0;
    }
    printf("{\"bench\":\"sustain\",\"op\":\"%s\",\"seconds\":%.2f,\"Gmul_per_s\":%.1f}\n", name, total_ms / 1e3, n / (total_ms * 1e-3) / 1e9);
}

// correctness of fe_mul512_s (carry counts on the scalar unit) against fe_mul512: all 16 words of the 512-bit product, random operands and operands made of
// 0xFFFFFFFF / 0 / 1 words (every column then carries as often as it can)
__global__ void salu_check_kernel(const fe *a_in, const fe *b_in, unsigned long long *bad, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const fe a = a_in[t], b = b_in[t];
    u32 w0[16], w1[16];
    fe_mul512(w0, a.v, b.v);
    fe_mul512_s(w1, a.v, b.v);
    unsigned long long mine = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) mine += w0[i] != w1[i];
    if (t & 1) {                                                        // half of the threads of every wave stop here: the masks of inactive lanes must not count
        if (mine) atomicAdd(bad, mine);
        return;
    }
    fe_mul512(w0, b.v, a.v);
    fe_mul512_s(w1, b.v, a.v);
#pragma unroll
    for (int i = 0; i < 16; i++) mine += w0[i] != w1[i];
    if (mine) atomicAdd(bad, mine);
}
static int salu_check()
{
    const int n = 1 << 18;
    std::vector<fe> a(n), b(n);
    u64 s = 0x0123456789ABCDEFull;
    auto rnd = [&]() { s += 0x9E3779B97F4A7C15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (u32)((z ^ (z >> 31)) >> 16); };
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) {
            a[i].v[k] = rnd(); b[i].v[k] = rnd();
            if (i < (1 << 16)) {                                          // words from {0xFFFFFFFF, 0xFFFFFFFE, 0, 1, 0x80000000, random}
                static const u32 pick[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0u, 1u, 0x80000000u, 0xFFFFFFFFu, 0x7FFFFFFFu};
                const u32 ra = rnd(), rb = rnd();
                if ((ra >> 8) & 3) a[i].v[k] = pick[ra & 7];
                if ((rb >> 8) & 3) b[i].v[k] = pick[rb & 7];
            }
        }
    for (int k = 0; k < 8; k++) { a[0].v[k] = b[0].v[k] = 0xFFFFFFFFu; a[1].v[k] = 0xFFFFFFFFu; b[1].v[k] = 0xFFFFFFFEu; }
    fe *da, *db; unsigned long long *dbad, h = 0;
    CK(hipMalloc(&da, n * sizeof(fe))); CK(hipMalloc(&db, n * sizeof(fe))); CK(hipMalloc(&dbad, 8));
    CK(hipMemcpy(da, a.data(), n * sizeof(fe), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * sizeof(fe), hipMemcpyHostToDevice));
    CK(hipMemset(dbad, 0, 8));
    hipLaunchKernelGGL(salu_check_kernel, dim3(n / 256), dim3(256), 0, 0, da, db, dbad, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost));
    printf("{\"bench\":\"salucheck\",\"operand_pairs\":%d,\"mismatching_words\":%llu}\n", n, h);
    return h ? 1 : 0;
}

// correctness of the unsaturated multipliers against the engine's fe_mul: chains of dependent products from random and extreme operands, compared canonically
template <int BITS, int N>
__global__ void unsat_check_kernel(const fe *a_in, const fe *b_in, unsigned long long *bad, int n, int chain)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    fe a = a_in[t], b = b_in[t];
    unsat<BITS, N> ua, ub;
    ua.from_fe(a); ub.from_fe(b);
    unsigned long long mine = 0;
    for (int k = 0; k < chain; k++) {
        fe_mul(a, a, b); unsat<BITS, N>::mul(ua, ua, ub);                  // the plain form ...
        fe_mul(b, b, a); unsat<BITS, N>::mul_tuned(ub, ub, ua);            // ... and the one written for the ISA, alternately
        fe x = a, y; ua.to_fe(y);
        fe_canon(x); fe_canon(y);
        mine += !fe_eq(x, y);
        x = b; ub.to_fe(y);
        fe_canon(x); fe_canon(y);
        mine += !fe_eq(x, y);
    }
    if (mine) atomicAdd(bad, mine);
}
static int unsat_check()
{
    const int n = 1 << 16, chain = 16;
    std::vector<fe> a(n), b(n);
    u64 s = 0x123456789ABCDEFull;
    auto rnd = [&]() { s += 0x9E3779B97F4A7C15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (u32)((z ^ (z >> 31)) >> 16); };
    const u32 P[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) {
            a[i].v[k] = rnd(); b[i].v[k] = rnd();
            if (i < 64) { a[i].v[k] = (i & 1) ? P[k] - (k == 0 ? (u32)(i >> 1) : 0u) : 0xFFFFFFFFu; if (i & 2) b[i].v[k] = 0xFFFFFFFFu; }      // p - small, 2^256 - 1: the extremes of the representation
            if (i >= 64 && i < 128) { a[i].v[k] = k == 0 ? (u32)(i - 63) : 0u; }
        }
    fe *da, *db; unsigned long long *dbad, h[2] = {0, 0};
    CK(hipMalloc(&da, n * sizeof(fe))); CK(hipMalloc(&db, n * sizeof(fe))); CK(hipMalloc(&dbad, 16));
    CK(hipMemcpy(da, a.data(), n * sizeof(fe), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * sizeof(fe), hipMemcpyHostToDevice));
    CK(hipMemset(dbad, 0, 16));
    hipLaunchKernelGGL((unsat_check_kernel<29, 9>), dim3(n / 256), dim3(256), 0, 0, da, db, dbad, n, chain);
    hipLaunchKernelGGL((unsat_check_kernel<26, 10>), dim3(n / 256), dim3(256), 0, 0, da, db, dbad + 1, n, chain);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, dbad, 16, hipMemcpyDeviceToHost));
    printf("{\"bench\":\"unsatcheck\",\"products_each\":%d,\"mismatches_9x29\":%llu,\"mismatches_10x26\":%llu}\n", n * chain * 2, h[0], h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}

// correctness of dpf_mul512 against exact integer arithmetic on the host: 2^16 random operand pairs
__global__ void dpf_check_kernel(const double *a_in, const double *b_in, double *v_out, int n)
{
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t >= n) return;
    fd6 a, b;
    for (int i = 0; i < 6; i++) { a.l[i] = a_in[t * 6 + i]; b.l[i] = b_in[t * 6 + i]; }
    double V[12];
    dpf_mul512(V, a, b);
    for (int k = 0; k < 12; k++) v_out[t * 12 + k] = V[k];
}
static int dpf_check()
{
    const int n = 1 << 16;
    std::vector<double> a(n * 6), b(n * 6), v(n * 12);
    u64 s = 0x1234567;
    auto rnd48 = [&]() { s += 0x9E3779B97F4A7C15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return (z ^ (z >> 31)) & 0xFFFFFFFFFFFFULL; };
    for (int t = 0; t < n; t++)
        for (int i = 0; i < 6; i++) {
            u64 x = rnd48(), y = rnd48();
            if (t < 64) { x = (t & 1) ? 0xFFFFFFFFFFFFULL : (t & 2 ? 0 : x); y = (t & 4) ? 0xFFFFFFFFFFFFULL : (t & 8 ? 1 : y); }   // extremes first
            a[t * 6 + i] = (double)x; b[t * 6 + i] = (double)y;
        }
    double *da, *db, *dv;
    CK(hipMalloc(&da, a.size() * 8)); CK(hipMalloc(&db, b.size() * 8)); CK(hipMalloc(&dv, v.size() * 8));
    CK(hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dpf_check_kernel, dim3(n / 256), dim3(256), 0, 0, da, db, dv, n);
    CK(hipMemcpy(v.data(), dv, v.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    long long negative = 0;
    for (int t = 0; t < n; t++) {
        __int128 col[12] = {0}, got[12] = {0};
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) col[i + j] += (__int128)((unsigned __int128)(u64)a[t * 6 + i] * (u64)b[t * 6 + j]);
        for (int k = 0; k < 12; k++) { got[k] = (__int128)(long long)v[t * 12 + k]; negative += v[t * 12 + k] < 0; }   // column values are exact integers, |V| < 2^52
        __int128 c1 = 0, c2 = 0;                   // normalise both to 48-bit limbs (arithmetic shifts: floor) and compare
        for (int k = 0; k < 12; k++) {
            c1 += col[k]; c2 += got[k];
            if ((u64)(c1 & 0xFFFFFFFFFFFFULL) != (u64)(c2 & 0xFFFFFFFFFFFFULL)) { bad++; break; }
            c1 >>= 48; c2 >>= 48;
        }
        if (c1 != c2) bad++;
    }
    printf("{\"bench\":\"dpf_check\",\"cases\":%d,\"mismatches\":%d,\"negative_column_values\":%lld,"
           "\"note\":\"sum V_k 2^(48k) == a*b exactly (column values are signed)\"}\n", n, bad, negative);
    return bad;
}

// ---------------------------------------------------------------- scatter into bucket lines (round 4: what bounds the baby-table builder)
// Every key claims a slot of a random 64-byte line with an atomic add on the line's word 0 and stores its hash into the slot it got (the
// dependent store of ext_scatter_kernel).  MODE 0: device-scope atomic (what hipcc emits for atomicAdd); 1: workgroup-scope atomic, and every
// block only touches the lines of ITS XCD (line index mod 8 == XCC_ID: the atomics then execute in that XCD's L2 and no other XCD ever sees
// the line during the kernel); 2: no atomic at all, a plain load of word 0 and two stores (wrong counts; the memory system's own ceiling for
// a random read-modify-write of a line); 3: MODE 1's atomics without the XCD discipline (wrong in general; isolates the cost of the scope)
template <int MODE>
__global__ void __launch_bounds__(256) scatter_kernel(u32 *__restrict__ lines, u64 line_mask, int iters, u64 seed)
{
    const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
    u64 s = seed + (u64)tid * 0x632BE59BD9B4E019ULL;
    const u32 xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;             // HW_REG_XCC_ID[3:0]
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s += 0x9E3779B97F4A7C15ULL;
            u64 z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            u64 line = z & line_mask;
            if (MODE == 1) line = (line & ~7ull) | xcc;
            u32 *L = lines + line * 16;
            u32 slot;
            if (MODE == 0) slot = atomicAdd(L, 1u);
            else if (MODE == 1 || MODE == 3) slot = __hip_atomic_fetch_add(L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else { slot = L[0]; L[0] = slot + 1; }
            L[1 + (slot % 15u)] = (u32)(z >> 32);
        }
    }
}
template <int MODE>
static void run_scatter(const char *name, u32 *lines, size_t bytes)
{
    const u64 nlines = bytes / 64;
    u64 mask = 1; while (mask * 2 <= nlines) mask *= 2; mask -= 1;
    const int blocks = 256 * 16, iters = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemsetAsync(lines, 0, bytes));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, lines, mask, iters, 1234567ull + rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double keys = (double)blocks * 256 * iters * 4;
    printf("{\"bench\":\"scatter\",\"mode\":\"%s\",\"footprint_GiB\":%.1f,\"Gkeys_per_s\":%.2f,\"ms\":%.2f}\n", name, bytes / 1073741824.0, keys / (best * 1e-3) / 1e9, best);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "scatter")) {
        for (int i = 2; i < argc; i++) {
            const size_t bytes = (size_t)atoll(argv[i]) << 20;
            u32 *lines; CK(hipMalloc(&lines, bytes));
            run_scatter<0>("device-scope atomicAdd + dependent store", lines, bytes);
            run_scatter<1>("workgroup-scope atomic, lines partitioned by XCD + dependent store", lines, bytes);
            run_scatter<3>("workgroup-scope atomic, NOT partitioned (wrong in general)", lines, bytes);
            run_scatter<2>("plain load + two stores (no atomic: wrong counts)", lines, bytes);
            CK(hipFree(lines));
        }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "dpfcheck")) return dpf_check() ? 1 : 0;
    if (argc >= 2 && !strcmp(argv[1], "unsatcheck")) return unsat_check();
    if (argc >= 2 && !strcmp(argv[1], "salucheck")) return salu_check();
    if (argc >= 4 && !strcmp(argv[1], "power")) {
        u32 *dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
        const int op = atoi(argv[2]); const double secs = atof(argv[3]);
        switch (op) {
        case 200: sustain_mul<200>("fe_mul: integer 256x256 product + fold mod p (the engine's multiplier)", secs, dout); break;
        case 201: sustain_mul<201>("fe_mul512: integer 256x256->512 product only", secs, dout); break;
        case 206: sustain_mul<206>("fe_mul512_s: the same product, carry-outs counted on the scalar unit (bit-sliced pairs, 43 vector carry steps instead of 62)", secs, dout); break;
        case 207: sustain_mul<207>("fe_mul512_s + fold mod p", secs, dout); break;
        case 203: sustain_mul<203>("unsat 9x29 (written for the ISA: alignbit carries, multiply-adds for every 64-bit addition): 81 products in 64-bit columns, no carry counts, + fold mod p, normalised in and out", secs, dout); break;
        case 204: sustain_mul<204>("unsat 10x26 (written for the ISA): 100 products in 64-bit columns, no carry counts, + fold mod p, normalised in and out", secs, dout); break;
        case 205: sustain_mul<205>("unsat 9x29 (plain C++: 64-bit shifts and additions as the compiler lowers them)", secs, dout); break;
        case 202: sustain_mul<202>("dpf_mul512: FP64 6x48-bit-limb 288x288->576 product only (no fold, no normalisation, no conversion)", secs, dout); break;
        case 105: sustain_gups<2>(secs, 16384, dout); break;       // 32-byte half lines: 2 lanes x 16 B
        case 100: sustain_gups<4>(secs, 16384, dout); break;
        case 101: sustain_gups<8>(secs, 16384, dout); break;
        case 102: sustain_gups<4>(secs, 16, dout); break;          // L2-resident footprint
        case 110: sustain_gups<4>(secs, 64, dout); break;          // footprints between the L2s (8 x 4 MiB) and HBM: what the 256 MB memory-side cache serves
        case 111: sustain_gups<4>(secs, 128, dout); break;
        case 112: sustain_gups<4>(secs, 192, dout); break;
        case 113: sustain_gups<4>(secs, 256, dout); break;
        case 114: sustain_gups<4>(secs, 512, dout); break;
        case 115: sustain_gups<4>(secs, 1024, dout); break;
        case 116: sustain_gups<4>(secs, 4096, dout); break;
        case 103: sustain_stream(secs, 16384, dout); break;
        case 104: sustain_stream(secs, 16, dout); break;
        case 3: sustain<3>("v_add_u32", 4, secs, dout); break;
        case 0: sustain<0>("v_mad_u64_u32", 4, secs, dout); break;
        case 1: sustain<1>("v_mul_lo_u32", 4, secs, dout); break;
        case 4: sustain<4>("v_fma_f64", 4, secs, dout); break;
        case 6: sustain<6>("carry_chain_add_co+3addc", 4, secs, dout); break;
        case 9: sustain<9>("4mad+3addc_pipelined", 7, secs, dout); break;
        default: printf("unknown op\n"); return 1;
        }
        return 0;
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\":\"%s\",\"cus\":%d,\"clock_MHz\":%d,\"mem_GiB\":%.1f}\n", prop.name, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.totalGlobalMem / 1073741824.0);
    u32 *dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    run_rate<3>("v_add_u32", 4, dout);
    run_rate<0>("v_mad_u64_u32", 4, dout);
    run_rate<1>("v_mul_lo_u32", 4, dout);
    run_rate<2>("v_mul_hi_u32", 4, dout);
    run_rate<4>("v_fma_f64", 4, dout);
    run_rate<5>("v_mad_u32_u24", 4, dout);
    run_rate<8>("v_mul_hi_u32_u24", 4, dout);
    run_rate<6>("carry_chain_add_co+3addc(with s_nop 1)", 4, dout);
    run_rate<7>("v_lshl_add_u64", 4, dout);
    run_rate<9>("4mad+3addc_pipelined", 7, dout);

    std::vector<size_t> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back((size_t)atoll(argv[i]) << 20);
    if (sizes.empty()) { sizes.push_back((size_t)384 << 20); sizes.push_back((size_t)5120 << 20); sizes.push_back((size_t)16384 << 20); }
    for (size_t bytes : sizes) {
        uint4 *buf; CK(hipMalloc(&buf, bytes));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 16);
        CK(hipDeviceSynchronize());
        for (int wps : {4, 8}) {
            run_gups<1, 8>((const u32x4*)buf, bytes, dout, wps);    // 16 B
            run_gups<2, 8>((const u32x4*)buf, bytes, dout, wps);    // 32 B
            run_gups<4, 4>((const u32x4*)buf, bytes, dout, wps);    // 64 B
            run_gups<4, 8>((const u32x4*)buf, bytes, dout, wps);    // 64 B, deeper
            run_gups<8, 4>((const u32x4*)buf, bytes, dout, wps);    // 128 B
            run_gups_coop<2, 8>((const u32x4*)buf, bytes, dout, wps);   // 32 B by 2 lanes
            run_gups_coop<4, 8>((const u32x4*)buf, bytes, dout, wps);   // 64 B by 4 lanes
            run_gups_coop<8, 8>((const u32x4*)buf, bytes, dout, wps);   // 128 B by 8 lanes
        }
        CK(hipFree(buf));
    }
    return 0;
}
