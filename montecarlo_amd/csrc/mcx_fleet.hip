/*
 * mcx_fleet.hip - the dashboard's module list for a whole fleet of hosts in one
 * call: alias match -> used drivers -> loaded set (closed under "needs") ->
 * holders -> root filter, everything on the device.  Restates per host the loop
 * of reference desktop/ui.py:1565-1599 (see include/mcx.h for the exact rule).
 *
 * Sets of hosts are bit columns.  Two matrices of V rows (modules) by
 * HW = ceil(H/32) words live in HBM, row-major, bit h%32 of word h/32 = host h:
 *   U  used drivers of each host, later overwritten with the result R
 *   L  extra modules of each host, later the closed loaded set
 * "host h has loaded everything its loaded modules need" is then one OR of rows
 * per edge, for all hosts at once, and because the modules are visited in
 * holder-level order (every holder of v before v) each row is read once per edge
 * and written once: E x HW word reads in total, no iteration to a fixpoint and no
 * atomics after the scatter.
 */
#include <stdlib.h>
#include <string.h>

#include "mcx_internal.h"
#include "mcx_scan.h"


/* largest h with off[h] <= i, for off[0] = 0 <= i < off[H] */
__device__ __forceinline__ uint32_t fl_owner(const uint32_t *__restrict__ off, uint32_t H, uint32_t i)
{
    uint32_t lo = 0, hi = H;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void fl_set(uint32_t *M, uint32_t HW, uint32_t m, uint32_t h)
{
    /* no look before the OR: a host rarely names a module twice, so the read
     * would only add a dependent round trip in front of nearly every atomic */
    atomicOr(&M[(size_t)m * HW + (h >> 5)], 1u << (h & 31u));
}

/*
 * One thread per device: U[driver module][host].  The devices of a block are
 * consecutive, so they belong to a short run of hosts: the block finds the
 * host of its first device once, copies that run of dev_off into LDS and every
 * thread searches there.
 */
__global__ void __launch_bounds__(FL_THREADS)
fl_scatter_used(const uint32_t *__restrict__ first_idx, const uint32_t *__restrict__ a_mod,
                const uint32_t *__restrict__ bound_mod, uint32_t N,
                const uint32_t *__restrict__ dev_off, uint32_t H, uint32_t V, uint32_t HW, uint32_t *U, uint32_t *L)
{
    __shared__ uint32_t run[FL_THREADS + 1]; /* dev_off[h0 .. h0 + FL_THREADS] */
    const uint32_t i0 = blockIdx.x * FL_THREADS;
    const uint32_t i = i0 + threadIdx.x;
    const uint32_t h0 = fl_owner(dev_off, H, i0);
    for (uint32_t k = threadIdx.x; k <= FL_THREADS; k += FL_THREADS)
        run[k] = h0 + k <= H ? dev_off[h0 + k] : 0xFFFFFFFFu;
    __syncthreads();
    if (i >= N)
        return;
    /* the module the device's modalias resolves to, and - when the caller says which
     * driver is bound to the device - the module of that name */
    const uint32_t j = first_idx[i];
    const uint32_t m = j == MCX_NO_MATCH ? MCX_NO_MATCH : a_mod[j];
    const uint32_t b = bound_mod ? bound_mod[i] : MCX_NO_MATCH;
    if (m >= V && b >= V)
        return;
    /* largest k with run[k] <= i; beyond the copied run (256 hosts without a
     * device in a row) the search goes back to the whole array */
    uint32_t h;
    if (run[FL_THREADS] <= i) {
        h = fl_owner(dev_off, H, i);
    } else {
        uint32_t lo = 0, hi = FL_THREADS;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (run[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        h = h0 + lo;
    }
    if (!bound_mod) {
        fl_set(U, HW, m, h); /* used = resolved */
    } else {
        if (m < V)
            fl_set(L, HW, m, h); /* resolved: loaded, like a module loaded by hand */
        if (b < V)
            fl_set(U, HW, b, h); /* used = bound */
    }
}

/*
 * The same result by word column: a block owns the 32 hosts of one word of every row, takes
 * their devices (consecutive in the arrays) and ORs the host bits into a column of V words in
 * LDS; then it stores the words that are not 0 (the matrices start out zeroed).  The devices of
 * 32 hosts name a few hundred modules between them, so a column costs a few hundred plain
 * stores instead of one atomic per device on a row 4 * HW bytes from the next.
 * Dynamic LDS: V words, twice that with bound_mod (a column of U and one of L).
 */
#define FL_COL_LDS_MAX 65536u   /* what a block may ask for without raising the kernel's limit */
#define FL_COL_LDS_STATIC 256u  /* the column kernels' own static part (hoff), rounded up */
__global__ void __launch_bounds__(FL_THREADS)
fl_scatter_cols(const uint32_t *__restrict__ first_idx, const uint32_t *__restrict__ a_mod,
                const uint32_t *__restrict__ bound_mod, uint32_t N, const uint32_t *__restrict__ dev_off,
                uint32_t H, uint32_t V, uint32_t HW, uint32_t *__restrict__ U, uint32_t *__restrict__ L)
{
    extern __shared__ uint32_t fl_col[];
    __shared__ uint32_t hoff[33];
    uint32_t *ucol = fl_col, *lcol = fl_col + V;
    const uint32_t w = blockIdx.x, h0 = 32u * w, nh = min(32u, H - h0);
    if (threadIdx.x <= nh)
        hoff[threadIdx.x] = dev_off[h0 + threadIdx.x];
    for (uint32_t v = threadIdx.x; v < (bound_mod ? 2u * V : V); v += FL_THREADS)
        fl_col[v] = 0u;
    __syncthreads();
    const uint32_t i0 = hoff[0], i1 = min(hoff[nh], N); /* the offsets are the caller's: stay inside the arrays */
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += FL_THREADS) {
        const uint32_t j = first_idx[i];
        const uint32_t m = j == MCX_NO_MATCH ? MCX_NO_MATCH : a_mod[j];
        const uint32_t b = bound_mod ? bound_mod[i] : MCX_NO_MATCH;
        if (m >= V && b >= V)
            continue;
        /* largest k < nh with hoff[k] <= i */
        uint32_t lo = 0, hi = nh;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (hoff[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t bit = 1u << lo;
        if (!bound_mod) {
            atomicOr(&ucol[m], bit); /* used = resolved */
        } else {
            if (m < V)
                atomicOr(&lcol[m], bit); /* resolved: loaded, like a module loaded by hand */
            if (b < V)
                atomicOr(&ucol[b], bit); /* used = bound */
        }
    }
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < V; v += FL_THREADS) {
        const uint32_t u = ucol[v];
        if (u != 0u)
            U[(size_t)v * HW + w] = u;
        if (bound_mod) {
            const uint32_t l = lcol[v];
            if (l != 0u)
                L[(size_t)v * HW + w] = l;
        }
    }
}

/* one thread per entry of the hosts' extra-module lists: L[module][host] */
__global__ void __launch_bounds__(FL_THREADS)
fl_scatter_extra(const uint32_t *__restrict__ extra_off, const uint32_t *__restrict__ extra_ids, uint32_t X,
                 uint32_t H, uint32_t V, uint32_t HW, uint32_t *L)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= X)
        return;
    const uint32_t m = extra_ids[k];
    if (m < V)
        fl_set(L, HW, m, fl_owner(extra_off, H, k));
}

/*
 * "(In Use)" labels (reference desktop/ui.py:1579-1590 asks mc_driver_is_in_use for every
 * module it lists; montecarlo/libmontecarlo.c:573-725 answers from the device links under
 * /sys/bus/{pci,usb,pcmcia}/drivers/<name> for the module's name and its name variant).
 * Over records: every device of a host on one of those three buses - listed or not, inside
 * the list's limit or beyond - that has a driver bound marks the modules that driver's name
 * stands for (mark_off / mark_ids by driver number, mcx_driver_marks).  I[module][host], same
 * layout as U and L.  By word column like fl_scatter_cols; fl_mark_rec is the one-thread-per-
 * record form for module counts whose column does not fit LDS.
 */
__device__ __forceinline__ bool fl_bus_has_driver_links(uint32_t subsystem)
{
    return subsystem == MCX_SUB_PCI || subsystem == MCX_SUB_USB || subsystem == MCX_SUB_PCMCIA;
}

__global__ void __launch_bounds__(FL_THREADS)
fl_mark_cols(const uint4 *__restrict__ rec, uint32_t N, const uint32_t *__restrict__ dev_off, uint32_t H,
             const uint32_t *__restrict__ mark_off, const uint32_t *__restrict__ mark_ids, uint32_t D, uint32_t V,
             uint32_t HW, uint32_t *__restrict__ I)
{
    extern __shared__ uint32_t fl_col[];
    __shared__ uint32_t hoff[33];
    const uint32_t w = blockIdx.x, h0 = 32u * w, nh = min(32u, H - h0);
    if (threadIdx.x <= nh)
        hoff[threadIdx.x] = dev_off[h0 + threadIdx.x];
    for (uint32_t v = threadIdx.x; v < V; v += FL_THREADS)
        fl_col[v] = 0u;
    __syncthreads();
    const uint32_t i0 = hoff[0], i1 = min(hoff[nh], N);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += FL_THREADS) {
        const uint4 r = rec[i]; /* mcx_dev_rec_t: subsystem in the low byte of x, driver number in w */
        const uint32_t d = r.w;
        if (!fl_bus_has_driver_links(r.x & 0xFFu) || d >= D)
            continue;
        const uint32_t e1 = mark_off[d + 1];
        uint32_t e = mark_off[d];
        if (e >= e1)
            continue;
        uint32_t lo = 0, hi = nh;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (hoff[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        for (; e < e1; ++e) {
            const uint32_t m = mark_ids[e];
            if (m < V)
                atomicOr(&fl_col[m], 1u << lo);
        }
    }
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < V; v += FL_THREADS) {
        const uint32_t u = fl_col[v];
        if (u != 0u)
            I[(size_t)v * HW + w] = u;
    }
}

__global__ void __launch_bounds__(FL_THREADS)
fl_mark_rec(const uint4 *__restrict__ rec, uint32_t N, const uint32_t *__restrict__ dev_off, uint32_t H,
            const uint32_t *__restrict__ mark_off, const uint32_t *__restrict__ mark_ids, uint32_t D, uint32_t V,
            uint32_t HW, uint32_t *I)
{
    const uint32_t i = blockIdx.x * FL_THREADS + threadIdx.x;
    if (i >= N)
        return;
    const uint4 r = rec[i];
    const uint32_t d = r.w;
    if (!fl_bus_has_driver_links(r.x & 0xFFu) || d >= D)
        return;
    const uint32_t e1 = mark_off[d + 1];
    uint32_t e = mark_off[d];
    if (e >= e1)
        return;
    const uint32_t h = fl_owner(dev_off, H, i);
    for (; e < e1; ++e) {
        const uint32_t m = mark_ids[e];
        if (m < V)
            fl_set(I, HW, m, h);
    }
}

/*
 * One holder level.  Thread (v, w): hold = OR over the holders u of v of L[u][w]
 * (final: u sits in an earlier level), loaded = U | L | hold,
 *   L[v][w] = loaded
 *   U[v][w] = eligible[v] ? loaded & ~used & ~hold : 0     (the root bits)
 */
__global__ void __launch_bounds__(FL_THREADS)
fl_level(const uint32_t *__restrict__ order, uint32_t n_rows, const uint32_t *__restrict__ rrowptr,
         const uint32_t *__restrict__ rcol, const uint8_t *__restrict__ safe_name,
         const uint8_t *__restrict__ hw_alias, uint32_t HW, uint32_t *U, uint32_t *L)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_rows * HW)
        return;
    const uint32_t v = order[(uint32_t)(t / HW)];
    const uint32_t w = (uint32_t)(t % HW);
    uint32_t hold = 0;
    const uint32_t e1 = rrowptr[v + 1];
    uint32_t e = rrowptr[v];
    /* eight holders at a time: their ids, then their rows, each requested together
     * (a core module has thousands of holders) */
    for (; e + 8u <= e1; e += 8u) {
        uint32_t u[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            u[k] = rcol[e + k];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            hold |= L[(size_t)u[k] * HW + w];
    }
    for (; e < e1; ++e)
        hold |= L[(size_t)rcol[e] * HW + w];
    const size_t at = (size_t)v * HW + w;
    const uint32_t used = U[at];
    const uint32_t loaded = used | L[at] | hold;
    L[at] = loaded;
    U[at] = (safe_name[v] && hw_alias[v]) ? (loaded & ~used & ~hold) : 0u;
}

/*
 * The same for a level of few modules with many holders each (the core modules at the end of the
 * order: one row, thousands of holders): a block per module and 64 word columns, wave s of the
 * block ORs the s-th part of the holder list, the parts meet in LDS and wave 0 writes the row.
 * The one-thread-per-word kernel would walk such a list with 3000 threads on the whole chip.
 */
#define FLS_MAX 16u
__global__ void __launch_bounds__(64 * FLS_MAX)
fl_level_sliced(const uint32_t *__restrict__ order, uint32_t n_rows, const uint32_t *__restrict__ rrowptr,
                const uint32_t *__restrict__ rcol, const uint8_t *__restrict__ safe_name,
                const uint8_t *__restrict__ hw_alias, uint32_t HW, uint32_t chunks, uint32_t *U, uint32_t *L)
{
    __shared__ uint32_t part[FLS_MAX][64];
    const uint32_t c = threadIdx.x & 63u, s = threadIdx.x >> 6, slices = blockDim.x >> 6;
    const uint32_t row = blockIdx.x / chunks, w = (blockIdx.x - row * chunks) * 64u + c;
    const uint32_t v = order[row];
    const uint32_t e0 = rrowptr[v], n = rrowptr[v + 1] - e0, per = (n + slices - 1u) / slices;
    uint32_t e = min(n, s * per) + e0;
    const uint32_t e1 = min(n, s * per + per) + e0;
    uint32_t hold = 0;
    if (w < HW) {
        for (; e + 8u <= e1; e += 8u) {
            uint32_t u[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k)
                u[k] = rcol[e + k];
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k)
                hold |= L[(size_t)u[k] * HW + w];
        }
        for (; e < e1; ++e)
            hold |= L[(size_t)rcol[e] * HW + w];
    }
    part[s][c] = hold;
    __syncthreads();
    if (s != 0u || w >= HW)
        return;
    for (uint32_t k = 1; k < slices; ++k)
        hold |= part[k][c];
    const size_t at = (size_t)v * HW + w;
    const uint32_t used = U[at];
    const uint32_t loaded = used | L[at] | hold;
    L[at] = loaded;
    U[at] = (safe_name[v] && hw_alias[v]) ? (loaded & ~used & ~hold) : 0u;
}

/*
 * R -> per-host lists.  One thread per word column of R (32 hosts) and row
 * segment (FLC_SEGS segments of consecutive modules): a column alone would be
 * H / 32 threads for the whole matrix.  A thread walks its rows in ascending
 * order, FLC_ROWS of them under way at a time, reading R coalesced (the set
 * bits are sparse: the inner loop rarely runs), with its 32 list cursors in
 * LDS.  FILL = false counts: seg_cnt[seg][h] = entries of host h in segment
 * seg.  fl_seg_prefix turns these into where each segment starts inside its
 * host's list and the host totals; after the scan of the totals FILL = true
 * writes the module ids from root_off[h] + seg_cnt[seg][h] on, so every host's
 * list comes out sorted.  LABEL: next to every id the "(In Use)" byte from the matrix I.
 */
#define FLC_THREADS 64
#define FLC_ROWS 32u
#define FLC_SEGS 16u
template <bool FILL, bool LABEL = false>
__global__ void __launch_bounds__(FLC_THREADS)
fl_columns(const uint32_t *__restrict__ R, uint32_t V, uint32_t H, uint32_t HW, uint32_t *__restrict__ seg_cnt,
           const uint32_t *__restrict__ root_off, uint32_t *__restrict__ root_ids,
           const uint32_t *__restrict__ I = nullptr, uint8_t *__restrict__ root_in_use = nullptr)
{
    __shared__ uint32_t cur[32][FLC_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint32_t w = blockIdx.x * FLC_THREADS + tid;
    const uint32_t seg = blockIdx.y;
    if (w >= HW)
        return;
    const uint32_t rows = (V + FLC_SEGS - 1u) / FLC_SEGS;
    const uint32_t v_lo = seg * rows, v_hi = min(V, v_lo + rows);
    uint32_t *mine = seg_cnt + (size_t)seg * H;
#pragma unroll
    for (uint32_t b = 0; b < 32u; ++b) {
        const uint32_t h = 32u * w + b;
        cur[b][tid] = (FILL && h < H) ? root_off[h] + mine[h] : 0u;
    }
    const uint32_t *col = R + w;
    for (uint32_t v0 = v_lo; v0 < v_hi; v0 += FLC_ROWS) {
        uint32_t r[FLC_ROWS];
#pragma unroll
        for (uint32_t k = 0; k < FLC_ROWS; ++k)
            r[k] = v0 + k < v_hi ? col[(size_t)(v0 + k) * HW] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < FLC_ROWS; ++k) {
            uint32_t bits = r[k];
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
                const uint32_t at = cur[b][tid]++;
                if (FILL)
                    root_ids[at] = v0 + k;
                if (FILL && LABEL) /* one byte per entry: the host's bit of the module's row of I */
                    root_in_use[at] = (uint8_t)((I[(size_t)(v0 + k) * HW + w] >> b) & 1u);
            }
        }
    }
    if (!FILL) {
#pragma unroll
        for (uint32_t b = 0; b < 32u; ++b) {
            const uint32_t h = 32u * w + b;
            if (h < H)
                mine[h] = cur[b][tid];
        }
    }
}

/* per host: seg_cnt[seg][h] -> entries of h in the segments before seg; cnt[h] = all of them */
__global__ void __launch_bounds__(FL_THREADS)
fl_seg_prefix(uint32_t *__restrict__ seg_cnt, uint32_t H, uint32_t *__restrict__ cnt)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H)
        return;
    uint32_t c[FLC_SEGS], run = 0;
#pragma unroll
    for (uint32_t s = 0; s < FLC_SEGS; ++s)
        c[s] = seg_cnt[(size_t)s * H + h];
#pragma unroll
    for (uint32_t s = 0; s < FLC_SEGS; ++s) {
        seg_cnt[(size_t)s * H + h] = run;
        run += c[s];
    }
    cnt[h] = run;
}

/* ------------------------------------------------------------------------- */
/* C ABI                                                                      */
/* ------------------------------------------------------------------------- */

/* d_bound_mod == nullptr: used = the modules the modaliases resolve to; otherwise used = d_bound_mod[i].
 * lab (may be nullptr): where the "(In Use)" byte of every listed module comes from and goes to. */
int mcx_fleet_roots_enqueue(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, const void *d_dev_off,
                            uint32_t H, const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                            const void *d_safe_name, const void *d_bound_mod, const mcx_fleet_labels *lab,
                            void *d_root_off, void *d_root_ids, uint64_t cap, uint64_t *total_out)
{
    if (lab && !lab->root_in_use)
        lab = nullptr;
    if (lab && ((lab->Y && (!lab->inuse_off || !lab->inuse_ids)) ||
                (lab->n_rec && (!lab->rec || !lab->rec_off)) || (lab->D && lab->n_rec && (!lab->mark_off || !lab->mark_ids))))
        return MCX_E_BAD_ARG;
    if (!ctx || !d_s_off || !d_dev_off || !d_extra_off || !d_root_off || !total_out || (X && !d_extra_ids) ||
        (cap && !d_root_ids))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded || !ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    *total_out = 0;
    const uint32_t V = ctx->V;
    if (V && !d_safe_name)
        return MCX_E_BAD_ARG;
    if (ctx->lvl_ptr.empty()) {
        ctx->err = "the dependency graph has a cycle: no load order, no fleet pipeline";
        return MCX_E_BAD_ARG;
    }
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    if (H == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_root_off, 0, 4, ctx->stream));
        return MCX_OK;
    }
    const uint32_t HW = (H + 31u) / 32u;
    const uint32_t nb = (H + SCAN_ITEMS - 1u) / SCAN_ITEMS;
    /* scratch: U, L, first_idx, count, per-host counts, block sums, hw flags, total */
    const size_t mat = up256(4 * (size_t)(V ? V : 1) * HW);
    const size_t o_first = 2 * mat, o_count = o_first + up256(4 * (size_t)(N ? N : 1));
    const size_t o_cnt = o_count + up256(4 * (size_t)(N ? N : 1)), o_bsum = o_cnt + up256(4 * (size_t)H);
    const size_t o_hw = o_bsum + up256(4 * (size_t)nb), o_seg = o_hw + up256(V ? V : 1);
    const size_t o_total = o_seg + up256(4 * (size_t)FLC_SEGS * H);
    void *base;
    int rc = mcx_scratch(ctx, o_total + 256, &base);
    if (rc != MCX_OK)
        return rc;
    uint32_t *U = (uint32_t *)base, *L = (uint32_t *)((char *)base + mat);
    uint32_t *d_first = (uint32_t *)((char *)base + o_first), *d_count = (uint32_t *)((char *)base + o_count);
    uint32_t *d_cnt = (uint32_t *)((char *)base + o_cnt), *d_bsum = (uint32_t *)((char *)base + o_bsum);
    uint32_t *d_seg = (uint32_t *)((char *)base + o_seg);
    const dim3 col_grid((HW + FLC_THREADS - 1) / FLC_THREADS, FLC_SEGS);
    uint8_t *d_hw = (uint8_t *)base + o_hw;
    unsigned long long *d_total = (unsigned long long *)((char *)base + o_total);

    HIPCHK(ctx, hipMemsetAsync(base, 0, 2 * mat, ctx->stream));
    if ((rc = mcx_k1_hw_flags_dev(ctx, V, d_hw)) != MCX_OK)
        return rc;
    if (N) {
        if ((rc = mcx_alias_match_dev(ctx, d_S, d_s_off, N, d_first, d_count)) != MCX_OK)
            return rc;
        const size_t col_lds = 4 * (size_t)V * (d_bound_mod ? 2 : 1);
        if (col_lds + FL_COL_LDS_STATIC <= FL_COL_LDS_MAX)
            fl_scatter_cols<<<HW, FL_THREADS, col_lds, ctx->stream>>>(d_first, ctx->d_amod,
                                                                      (const uint32_t *)d_bound_mod, N,
                                                                      (const uint32_t *)d_dev_off, H, V, HW, U, L);
        else
            fl_scatter_used<<<(N + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(
                d_first, ctx->d_amod, (const uint32_t *)d_bound_mod, N, (const uint32_t *)d_dev_off, H, V, HW, U, L);
        HIPCHK(ctx, hipGetLastError());
    }
    if (X)
        fl_scatter_extra<<<(X + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(
            (const uint32_t *)d_extra_off, (const uint32_t *)d_extra_ids, X, H, V, HW, L);
    for (size_t k = 0; k + 1 < ctx->lvl_ptr.size(); ++k) {
        const uint32_t r0 = ctx->lvl_ptr[k], n_rows = ctx->lvl_ptr[k + 1] - r0;
        const uint64_t threads = (uint64_t)n_rows * HW;
        const uint64_t blocks = (threads + FL_THREADS - 1) / FL_THREADS;
        if (blocks > 0x7FFFFFFFull) {
            ctx->err = "fleet too large for one launch: split the fleet into groups of hosts";
            return MCX_E_TOO_LARGE;
        }
        /* few threads and long holder lists: the lists are cut into parts of 32 holders or more */
        uint32_t slices = 1;
        while (slices < FLS_MAX && threads * slices < 400000ull && ctx->lvl_maxdeg[k] >= 64u * slices)
            slices *= 2u;
        const uint32_t chunks = (HW + 63u) / 64u;
        if (slices > 1u && (uint64_t)n_rows * chunks <= 0x7FFFFFFFull)
            fl_level_sliced<<<n_rows * chunks, 64u * slices, 0, ctx->stream>>>(
                ctx->d_lvl_order + r0, n_rows, ctx->d_rrowptr, ctx->d_rcol, (const uint8_t *)d_safe_name, d_hw, HW,
                chunks, U, L);
        else
            fl_level<<<(uint32_t)blocks, FL_THREADS, 0, ctx->stream>>>(ctx->d_lvl_order + r0, n_rows, ctx->d_rrowptr,
                                                                      ctx->d_rcol, (const uint8_t *)d_safe_name, d_hw,
                                                                      HW, U, L);
    }
    fl_columns<false><<<col_grid, FLC_THREADS, 0, ctx->stream>>>(U, V, H, HW, d_seg, nullptr, nullptr);
    fl_seg_prefix<<<(H + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(d_seg, H, d_cnt);
    fl_scan_sums<<<nb, FL_THREADS, 0, ctx->stream>>>(d_cnt, H, d_bsum);
    fl_scan_top<<<1, FL_THREADS, 0, ctx->stream>>>(d_bsum, nb, d_total);
    fl_scan_apply<<<nb, FL_THREADS, 0, ctx->stream>>>(d_cnt, H, d_bsum, (uint32_t *)d_root_off);
    HIPCHK(ctx, hipGetLastError());
    if (lab) {
        /* the loaded sets are not needed any more: L becomes I, the hosts on which a bound
         * device marks the module */
        HIPCHK(ctx, hipMemsetAsync(L, 0, mat, ctx->stream));
        if (lab->Y)
            fl_scatter_extra<<<(lab->Y + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(
                lab->inuse_off, lab->inuse_ids, lab->Y, H, V, HW, L);
        if (lab->n_rec && lab->D) {
            if (4 * (size_t)V + FL_COL_LDS_STATIC <= FL_COL_LDS_MAX)
                fl_mark_cols<<<HW, FL_THREADS, 4 * (size_t)V, ctx->stream>>>((const uint4 *)lab->rec, lab->n_rec,
                                                                           lab->rec_off, H, lab->mark_off,
                                                                           lab->mark_ids, lab->D, V, HW, L);
            else
                fl_mark_rec<<<(lab->n_rec + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(
                    (const uint4 *)lab->rec, lab->n_rec, lab->rec_off, H, lab->mark_off, lab->mark_ids, lab->D, V, HW,
                    L);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    unsigned long long *h_total = (unsigned long long *)ctx->h_pinned;
    HIPCHK(ctx, hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *total_out = *h_total;
    if (*h_total > 0xFFFFFFFFull) {
        ctx->err = "more than 2^32 - 1 listed modules: split the fleet into groups of hosts";
        return MCX_E_TOO_LARGE;
    }
    if (*h_total > cap) {
        ctx->err = "root_ids too small: " + std::to_string(*h_total) + " entries needed";
        return MCX_E_TOO_LARGE;
    }
    if (*h_total && lab)
        fl_columns<true, true><<<col_grid, FLC_THREADS, 0, ctx->stream>>>(U, V, H, HW, d_seg,
                                                                          (const uint32_t *)d_root_off,
                                                                          (uint32_t *)d_root_ids, L, lab->root_in_use);
    else if (*h_total)
        fl_columns<true><<<col_grid, FLC_THREADS, 0, ctx->stream>>>(U, V, H, HW, d_seg,
                                                                    (const uint32_t *)d_root_off,
                                                                    (uint32_t *)d_root_ids);
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}

extern "C" {

int mcx_fleet_roots_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, const void *d_dev_off,
                        uint32_t H, const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                        const void *d_safe_name, void *d_root_off, void *d_root_ids, uint64_t cap,
                        uint64_t *total_out)
{
    return mcx_fleet_roots_enqueue(ctx, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X, d_safe_name,
                                   nullptr, nullptr, d_root_off, d_root_ids, cap, total_out);
}

int mcx_fleet_roots_bound_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, const void *d_dev_off,
                              uint32_t H, const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                              const void *d_safe_name, const void *d_bound_mod, void *d_root_off, void *d_root_ids,
                              uint64_t cap, uint64_t *total_out)
{
    if (N && !d_bound_mod)
        return MCX_E_BAD_ARG;
    return mcx_fleet_roots_enqueue(ctx, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X, d_safe_name,
                                   N ? d_bound_mod : nullptr, nullptr, d_root_off, d_root_ids, cap, total_out);
}

int mcx_fleet_roots_labeled_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N,
                                const void *d_dev_off, uint32_t H, const void *d_extra_off, const void *d_extra_ids,
                                uint32_t X, const void *d_safe_name, const void *d_bound_mod,
                                const void *d_inuse_off, const void *d_inuse_ids, uint32_t Y, void *d_root_off,
                                void *d_root_ids, void *d_root_in_use, uint64_t cap, uint64_t *total_out)
{
    if ((N && !d_bound_mod) || !d_inuse_off || (cap && !d_root_in_use))
        return MCX_E_BAD_ARG;
    mcx_fleet_labels lab;
    lab.inuse_off = (const uint32_t *)d_inuse_off;
    lab.inuse_ids = (const uint32_t *)d_inuse_ids;
    lab.Y = Y;
    lab.root_in_use = (uint8_t *)d_root_in_use;
    return mcx_fleet_roots_enqueue(ctx, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X, d_safe_name,
                                   N ? d_bound_mod : nullptr, &lab, d_root_off, d_root_ids, cap, total_out);
}

/* inuse_off != nullptr: with "(In Use)" labels from the hosts' lists of marked modules */
static int fleet_roots_host(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, const uint32_t *dev_off,
                            uint32_t H, const uint32_t *extra_off, const uint32_t *extra_ids,
                            const uint8_t *safe_name, const uint32_t *bound_mod, bool with_bound,
                            const uint32_t *inuse_off, const uint32_t *inuse_ids, uint32_t *root_off,
                            uint32_t *root_ids, uint8_t *root_in_use, uint64_t cap, uint64_t *total_out)
{
    if (!ctx || !s_off || !dev_off || !extra_off || !root_off || !total_out || (with_bound && N && !bound_mod))
        return MCX_E_BAD_ARG;
    const uint32_t Y = inuse_off ? inuse_off[H] : 0u;
    if (inuse_off) {
        if (inuse_off[0] != 0 || (Y && !inuse_ids) || (cap && !root_in_use))
            return MCX_E_BAD_ARG;
        for (uint32_t h = 0; h < H; ++h)
            if (inuse_off[h + 1] < inuse_off[h]) {
                ctx->err = "host offsets not monotonic";
                return MCX_E_BAD_ARG;
            }
    }
    if (!ctx->table_loaded || !ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    if (dev_off[0] != 0 || dev_off[H] != N || extra_off[0] != 0) {
        ctx->err = "dev_off must run from 0 to N and extra_off from 0";
        return MCX_E_BAD_ARG;
    }
    for (uint32_t h = 0; h < H; ++h)
        if (dev_off[h + 1] < dev_off[h] || extra_off[h + 1] < extra_off[h]) {
            ctx->err = "host offsets not monotonic";
            return MCX_E_BAD_ARG;
        }
    for (uint32_t i = 0; i < N; ++i)
        if (s_off[i + 1] < s_off[i]) {
            ctx->err = "s_off not monotonic";
            return MCX_E_BAD_ARG;
        }
    const uint32_t X = extra_off[H];
    if ((X && !extra_ids) || (N && s_off[N] && !S) || (ctx->V && !safe_name))
        return MCX_E_BAD_ARG;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    const uint32_t V = ctx->V;
    /* one staging allocation for the inputs and outputs */
    const size_t sb = s_off[N];
    const size_t o_soff = up256(sb ? sb : 1), o_dev = o_soff + up256(4 * ((size_t)N + 1));
    const size_t o_xoff = o_dev + up256(4 * ((size_t)H + 1)), o_xid = o_xoff + up256(4 * ((size_t)H + 1));
    const size_t o_safe = o_xid + up256(4 * (size_t)(X ? X : 1)), o_roff = o_safe + up256(V ? V : 1);
    const size_t o_rid = o_roff + up256(4 * ((size_t)H + 1));
    const size_t o_bound = o_rid + up256(4 * (size_t)(cap ? cap : 1));
    const size_t o_yoff = o_bound + up256(4 * (size_t)(N ? N : 1)), o_yid = o_yoff + up256(4 * ((size_t)H + 1));
    const size_t o_use = o_yid + up256(4 * (size_t)(Y ? Y : 1));
    const size_t bytes = o_use + up256(cap ? cap : 1);
    char *d = nullptr;
    if (hipMalloc((void **)&d, bytes) != hipSuccess) {
        ctx->err = "hipMalloc of the staging buffers failed";
        return MCX_E_NOMEM;
    }
    hipError_t e = hipSuccess;
    if (sb)
        e = hipMemcpyAsync(d, S, sb, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + o_soff, s_off, 4 * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + o_dev, dev_off, 4 * ((size_t)H + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + o_xoff, extra_off, 4 * ((size_t)H + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && X)
        e = hipMemcpyAsync(d + o_xid, extra_ids, 4 * (size_t)X, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && V)
        e = hipMemcpyAsync(d + o_safe, safe_name, V, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && with_bound && N)
        e = hipMemcpyAsync(d + o_bound, bound_mod, 4 * (size_t)N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && inuse_off)
        e = hipMemcpyAsync(d + o_yoff, inuse_off, 4 * ((size_t)H + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && Y)
        e = hipMemcpyAsync(d + o_yid, inuse_ids, 4 * (size_t)Y, hipMemcpyHostToDevice, ctx->stream);
    mcx_fleet_labels lab;
    if (inuse_off) {
        lab.inuse_off = (const uint32_t *)(d + o_yoff);
        lab.inuse_ids = (const uint32_t *)(d + o_yid);
        lab.Y = Y;
        lab.root_in_use = (uint8_t *)(d + o_use);
    }
    int rc = MCX_OK;
    if (e != hipSuccess) {
        ctx->err = std::string("copy in: ") + hipGetErrorString(e);
        rc = MCX_E_HIP;
    }
    if (rc == MCX_OK)
        rc = mcx_fleet_roots_enqueue(ctx, d, d + o_soff, N, d + o_dev, H, d + o_xoff, d + o_xid, X, d + o_safe,
                                     with_bound && N ? d + o_bound : nullptr, inuse_off && cap ? &lab : nullptr,
                                     d + o_roff, cap ? d + o_rid : nullptr, cap, total_out);
    /* the offsets are valid even when root_ids was too small: a caller may
     * size the list from them and call again */
    if (rc == MCX_OK || (rc == MCX_E_TOO_LARGE && *total_out <= 0xFFFFFFFFull && *total_out > cap)) {
        e = hipMemcpyAsync(root_off, d + o_roff, 4 * ((size_t)H + 1), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == MCX_OK && *total_out)
            e = hipMemcpyAsync(root_ids, d + o_rid, 4 * (size_t)*total_out, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == MCX_OK && *total_out && inuse_off)
            e = hipMemcpyAsync(root_in_use, d + o_use, (size_t)*total_out, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->err = std::string("copy back: ") + hipGetErrorString(e);
            rc = MCX_E_HIP;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

int mcx_fleet_roots(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, const uint32_t *dev_off,
                    uint32_t H, const uint32_t *extra_off, const uint32_t *extra_ids, const uint8_t *safe_name,
                    uint32_t *root_off, uint32_t *root_ids, uint64_t cap, uint64_t *total_out)
{
    return fleet_roots_host(ctx, S, s_off, N, dev_off, H, extra_off, extra_ids, safe_name, nullptr, false, nullptr,
                            nullptr, root_off, root_ids, nullptr, cap, total_out);
}

int mcx_fleet_roots_bound(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, const uint32_t *dev_off,
                          uint32_t H, const uint32_t *extra_off, const uint32_t *extra_ids, const uint8_t *safe_name,
                          const uint32_t *bound_mod, uint32_t *root_off, uint32_t *root_ids, uint64_t cap,
                          uint64_t *total_out)
{
    return fleet_roots_host(ctx, S, s_off, N, dev_off, H, extra_off, extra_ids, safe_name, bound_mod, true, nullptr,
                            nullptr, root_off, root_ids, nullptr, cap, total_out);
}

int mcx_fleet_roots_labeled(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, const uint32_t *dev_off,
                            uint32_t H, const uint32_t *extra_off, const uint32_t *extra_ids, const uint8_t *safe_name,
                            const uint32_t *bound_mod, const uint32_t *inuse_off, const uint32_t *inuse_ids,
                            uint32_t *root_off, uint32_t *root_ids, uint8_t *root_in_use, uint64_t cap,
                            uint64_t *total_out)
{
    if (!inuse_off)
        return MCX_E_BAD_ARG;
    return fleet_roots_host(ctx, S, s_off, N, dev_off, H, extra_off, extra_ids, safe_name, bound_mod, true, inuse_off,
                            inuse_ids, root_off, root_ids, root_in_use, cap, total_out);
}

} /* extern "C" */
