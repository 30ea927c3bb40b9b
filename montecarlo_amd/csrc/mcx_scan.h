/*
 * mcx_scan.h - exclusive prefix sums of 32-bit counts on the device, shared by
 * the translation units that turn per-host or per-device counts into offsets.
 */
#ifndef MCX_SCAN_H
#define MCX_SCAN_H

#include "mcx_internal.h"

#define FL_THREADS 256

/* ---- exclusive scan of H counts into H + 1 offsets ------------------------ */
/* three small kernels: per-block sums, one block over the sums, per-block scan */

#define SCAN_ITEMS 1024u /* per block: 256 threads x 4 */

__device__ __forceinline__ uint32_t fl_block_exclusive(uint32_t mine, uint32_t *total_out)
{
    __shared__ uint32_t wsum[FL_THREADS / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d)
            incl += o;
    }
    __syncthreads(); /* wsum may still be read by a previous call */
    if (lane == 63u)
        wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < FL_THREADS / 64; ++k) {
        if (k < wave)
            base += wsum[k];
        total += wsum[k];
    }
    *total_out = total;
    return base + incl - mine;
}

static __global__ void __launch_bounds__(FL_THREADS)
fl_scan_sums(const uint32_t *__restrict__ cnt, uint32_t H, uint32_t *__restrict__ bsum)
{
    const uint32_t i0 = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < 4u; ++k)
        if (i0 + k < H)
            mine += cnt[i0 + k];
    uint32_t total;
    (void)fl_block_exclusive(mine, &total);
    if (threadIdx.x == 0)
        bsum[blockIdx.x] = total;
}

/* single block: bsum[b] becomes the sum of the blocks before b; 64-bit grand total */
static __global__ void __launch_bounds__(FL_THREADS)
fl_scan_top(uint32_t *bsum, uint32_t nb, unsigned long long *grand)
{
    unsigned long long carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += FL_THREADS) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t mine = b < nb ? bsum[b] : 0u;
        uint32_t total;
        const uint32_t ex = fl_block_exclusive(mine, &total);
        if (b < nb)
            bsum[b] = (uint32_t)carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0)
        *grand = carry;
}

static __global__ void __launch_bounds__(FL_THREADS)
fl_scan_apply(const uint32_t *__restrict__ cnt, uint32_t H, const uint32_t *__restrict__ bsum,
              uint32_t *__restrict__ off)
{
    const uint32_t i0 = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4u;
    uint32_t c[4], mine = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        c[k] = i0 + k < H ? cnt[i0 + k] : 0u;
        mine += c[k];
    }
    uint32_t total;
    uint32_t at = bsum[blockIdx.x] + fl_block_exclusive(mine, &total);
    for (uint32_t k = 0; k < 4u; ++k) {
        if (i0 + k < H)
            off[i0 + k] = at;
        at += c[k];
    }
    /* off[H]: the thread that owns the last count also writes the end */
    if (H - 1u >= i0 && H - 1u < i0 + 4u)
        off[H] = at;
}


/* cnt[0..n) -> off[0..n], grand total (64-bit) -> *d_total; bsum: ceil(n / SCAN_ITEMS) words of scratch */
static inline void mcx_scan_enqueue(hipStream_t stream, const uint32_t *cnt, uint32_t n, uint32_t *bsum,
                                    unsigned long long *d_total, uint32_t *off)
{
    const uint32_t nb = (n + SCAN_ITEMS - 1u) / SCAN_ITEMS;
    fl_scan_sums<<<nb, FL_THREADS, 0, stream>>>(cnt, n, bsum);
    fl_scan_top<<<1, FL_THREADS, 0, stream>>>(bsum, nb, d_total);
    fl_scan_apply<<<nb, FL_THREADS, 0, stream>>>(cnt, n, bsum, off);
}

#endif /* MCX_SCAN_H */
