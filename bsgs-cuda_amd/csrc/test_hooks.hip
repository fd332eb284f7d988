// test_hooks.hip -- TEST BUILD ONLY: linked into build/libbsgs_hip_test.so, never into the shipped libbsgs_hip.so (Makefile).  The corrupted replica that the
// verification must catch, and buffer re-allocation for the placement experiments (tools/placement_probe.py).
#define BSGS_TEST_HOOKS 1
#include "bsgs_internal.h"
#include "support_kernels.hip.h"
#include "host_secp.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// test hook: flip bits of ONE byte of the installed table (bucket lines if the layout has them, else the CSR image) -- the corrupted replica
// the verification must catch (tests/test_gpu_round4.py, bench.py BENCH_CORRUPT_RANK)
extern "C" int bsgs_debug_corrupt_table(bsgs_dev *d, uint64_t byte_offset, uint32_t xor_mask)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout) return fail(BSGS_ERR_STATE, "no table on device");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    uint8_t *base = d->lines ? (uint8_t *)d->lines : (uint8_t *)d->csr;
    const uint64_t bytes = d->lines ? d->lines_bytes : 4 * (d->ht_items + 1) + 4 * d->w;
    if (byte_offset >= bytes) return fail(BSGS_ERR_ARG, "offset %llu beyond the %llu bytes of the table", (unsigned long long)byte_offset, (unsigned long long)bytes);
    HIPCHK(hipSetDevice(d->id));
    uint8_t v = 0;
    HIPCHK(hipMemcpy(&v, base + byte_offset, 1, hipMemcpyDeviceToHost));
    v ^= (uint8_t)xor_mask;
    HIPCHK(hipMemcpy(base + byte_offset, &v, 1, hipMemcpyHostToDevice));
    return BSGS_OK;
}

// diagnostics: give ONE of the engine's buffers a new allocation with the same contents (0 = bucket lines, 1 = chain scratch,
// 2 = giants), optionally after a `spacer_bytes` allocation that is released again (so the new one lands elsewhere)
extern "C" int bsgs_debug_realloc(bsgs_dev *d, int which, uint64_t spacer_bytes)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued");
    HIPCHK(hipSetDevice(d->id));
    HIPCHK(hipStreamSynchronize(d->stream));
    void *spacer = nullptr;
    if (spacer_bytes) HIPCHK(hipMalloc(&spacer, spacer_bytes));
    hipError_t e = hipSuccess;
    if (which == 0 && d->lines && d->lines_owned) {
        void *n = nullptr;
        e = bsgs_big_malloc(&n, d->lines_bytes);
        if (e == hipSuccess) e = hipMemcpy(n, d->lines, d->lines_bytes, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) { (void)bsgs_big_free(d->lines); d->lines = (u32x4 *)n; }
    } else if (which == 1 && (d->chain || !d->chain_pieces.empty())) {
        if (d->chain) (void)hipFree(d->chain);
        free_chain_pieces(d);
        d->chain = nullptr; d->chain_bytes = 0;                                             // scratch: the next enqueue allocates it again
    } else if (which == 3) {                                     // a new HIP stream (= possibly another hardware queue)
        hipStream_t ns = nullptr;
        e = hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
        if (e == hipSuccess) { (void)hipStreamDestroy(d->stream); d->stream = ns; }
    } else if (which == 4 && d->hitbuf) {                        // hit buffer + centres
        u32 *nh = nullptr;
        e = hipMalloc(&nh, bsgs_hitbuf_bytes(d));
        if (e == hipSuccess) e = hipMemset(nh, 0, 64);
        if (e == hipSuccess) { (void)hipFree(d->hitbuf); d->hitbuf = nh; }
        if (d->cen_dev) { (void)hipFree(d->cen_dev); d->cen_dev = nullptr; }
        if (d->cen_pin) { (void)hipHostFree(d->cen_pin); d->cen_pin = nullptr; }
        d->cen_cap = 0;
    } else if (which == 2 && d->g2) {
        void *n = nullptr;
        e = hipMalloc(&n, d->maxnonce * 64);
        if (e == hipSuccess) e = hipMemcpy(n, d->g2, d->maxnonce * 64, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) { (void)hipFree(d->g2); d->g2 = (u32x4 *)n; }
    }
    if (spacer) (void)hipFree(spacer);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "debug_realloc: %s", hipGetErrorString(e));
    return BSGS_OK;
}
