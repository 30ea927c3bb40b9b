/*
 * mcx_k2.hip - K2: the module dependency graph.
 *   K2a dep_flags   has-holders / in-use / root bits and holder counts,
 *   K2b dep_closure reachability among loaded modules (frontier BFS),
 * HIP kernels for gfx950 and their C ABI entry points (include/mcx.h).
 * There is no CPU path here: every entry point needs a HIP device.
 *
 * Layout in HBM (built once by mcx_dep_graph_load):
 *   rowptr/col    forward CSR, u -> col[k]: "u needs col[k]"
 *   rrowptr/rcol  reverse CSR, v -> rcol[k]: "rcol[k] holds v" (the content of
 *                 /sys/module/<v>/holders)
 *   part          row blocks of the reverse CSR for K2a: consecutive rows, at
 *                 most K2_ROWS rows and about K2_EDGES edges each (a row with
 *                 more edges than that is a block of its own)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <vector>

#include "mcx_internal.h"

#define K2_THREADS 256
#define K2_ROWS 256u
#define K2_EDGES 4096u
#ifndef K2B_BLOCKS
#define K2B_BLOCKS 1024u  /* grid of one BFS level; strides over the frontier */
#endif
#define K2B_PINNED_END 44u /* six words of the pinned read-back area: see k2b_small's only_the_end */
#define K2B_BATCH 16u     /* most levels enqueued between two looks at the queue sizes */
#define K2B_MAX_LEVELS 64u   /* window of per-level counters, reused by deeper searches */

/* ------------------------------------------------------------------------- */
/* K2a: one block = one row block of the reverse CSR.                         */
/* Every edge (holder u of row v) is read once, coalesced, and answers "is u  */
/* loaded" from a packed bitmap; the answers of 64 edges are one ballot, kept */
/* in LDS as one bit per edge; a row's loaded holders are then the set bits   */
/* of its edge range.  The block writes the flags and counts of its own rows. */
/* No atomics, no search for the row of an edge, no zero-filled scratch.      */
/* What bounds it is the per-edge gather: 10^7 random 4-byte reads go through */
/* the vector cache at about one lane address per cycle and CU.               */
/* ------------------------------------------------------------------------- */

/* LOADED bits of in_flags packed 32 modules to a word: the per-edge gather of
 * K2a then reads a 125 KB array (V = 10^6) that stays in cache, not V bytes */
__global__ void __launch_bounds__(K2_THREADS)
k2_loaded_bitmap(const uint8_t *__restrict__ in_flags, uint32_t V, uint32_t aligned16,
                 uint32_t *__restrict__ bm)
{
    /* one thread per bitmap word: 32 flag bytes by two 16-byte loads.  The
     * readers copy whole 16-byte pieces: the words between the last one in use
     * and that rounding are written as zero, so that a bitmap buffer that held a
     * larger graph's bits before shows nothing of them. */
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v0 = 32u * w;
    if (v0 >= V) {
        if (w < ((((V + 31u) / 32u) + 3u) & ~3u))
            bm[w] = 0u;
        return;
    }
    uint32_t bits = 0;
    if (aligned16 && v0 + 32u <= V) {
        const uint4 *src = (const uint4 *)(in_flags + v0);
        const uint4 lo = src[0], hi = src[1];
        const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            /* LOADED is bit 0 of every byte: gather the four of them */
            const uint32_t m = q[k] & 0x01010101u;
            bits |= (((m * 0x10204080u) >> 28) & 0xFu) << (4u * k);
        }
    } else {
        for (uint32_t k = 0; k < 32u && v0 + k < V; ++k)
            bits |= (uint32_t)(in_flags[v0 + k] & MCX_F_LOADED) << k;
    }
    bm[w] = bits;
}

__global__ void __launch_bounds__(K2_THREADS)
k2_flags_rows(const uint32_t *__restrict__ rrowptr, const uint32_t *__restrict__ rcol,
              const uint32_t *__restrict__ part, const uint8_t *__restrict__ in_flags,
              const uint32_t *__restrict__ loaded_bm, uint8_t *__restrict__ out_flags,
              uint32_t *__restrict__ refcnt)
{
    __shared__ uint32_t rp[K2_ROWS + 1];
    __shared__ uint32_t cnt[K2_ROWS];
    __shared__ uint32_t wsum[K2_THREADS / 64];
    __shared__ uint32_t ebits[K2_EDGES / 32u + 8u];

    const uint32_t tid = threadIdx.x;
    const uint32_t r0 = part[blockIdx.x], r1 = part[blockIdx.x + 1];
    const uint32_t R = r1 - r0;
    for (uint32_t i = tid; i <= R; i += K2_THREADS)
        rp[i] = rrowptr[r0 + i];
    __syncthreads();
    const uint32_t e0 = rp[0], e1 = rp[R];

    if (R == 1u) {
        /* one (possibly very long) row: count in registers, reduce once */
        uint32_t mine = 0;
        for (uint32_t e = e0 + tid; e < e1; e += K2_THREADS) {
            const uint32_t u = rcol[e];
            mine += (loaded_bm[u >> 5] >> (u & 31u)) & 1u;
        }
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1)
            mine += __shfl_down(mine, d, 64);
        if ((tid & 63u) == 0)
            wsum[tid >> 6] = mine;
        __syncthreads();
        if (tid == 0)
            cnt[0] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    } else {
        /* edge-parallel and coalesced: "is the holder of edge e loaded", 64
         * edges per wave step, kept as one bit per edge in LDS ... */
        const uint32_t ne = e1 - e0, lane = tid & 63u;
        for (uint32_t base = tid - lane; base < ne; base += K2_THREADS) {
            uint32_t bit = 0;
            if (base + lane < ne) {
                const uint32_t u = rcol[e0 + base + lane];
                bit = (loaded_bm[u >> 5] >> (u & 31u)) & 1u;
            }
            const unsigned long long m = __ballot(bit != 0u);
            if (lane == 0u) {
                ebits[base >> 5] = (uint32_t)m;
                ebits[(base >> 5) + 1u] = (uint32_t)(m >> 32);
            }
        }
        __syncthreads();
        /* ... then row-parallel: a row's loaded holders = the set bits of its
         * edge range (no search for the row of an edge, no LDS atomics) */
        for (uint32_t i = tid; i < R; i += K2_THREADS) {
            const uint32_t a = rp[i] - e0, b = rp[i + 1u] - e0;
            uint32_t h = 0;
            for (uint32_t w = a >> 5; 32u * w < b; ++w) {
                uint32_t bits = ebits[w];
                if (32u * w < a)
                    bits &= 0xFFFFFFFFu << (a & 31u);
                if (32u * w + 32u > b)
                    bits &= 0xFFFFFFFFu >> ((32u - (b & 31u)) & 31u);
                h += (uint32_t)__popc(bits);
            }
            cnt[i] = h;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < R; i += K2_THREADS) {
        const uint32_t v = r0 + i;
        const uint32_t f = in_flags[v], h = cnt[i];
        uint32_t o = 0;
        if (f & MCX_F_LOADED) {
            if (h)
                o |= MCX_O_HAS_HOLDERS;
            if ((f & MCX_F_BOUND) || h)
                o |= MCX_O_IN_USE;
            if ((f & MCX_F_SAFE) && !h)
                o |= MCX_O_ROOT;
        }
        out_flags[v] = (uint8_t)o;
        if (refcnt)
            refcnt[v] = (f & MCX_F_LOADED) ? h + ((f & MCX_F_BOUND) ? 1u : 0u) : 0xFFFFFFFFu;
    }
}

/*
 * K2a with the LOADED bitmap in LDS: one block of 1024 threads per CU owns a
 * contiguous slice of rows with about E / #CU edges.  The holder ids of a
 * slice are kept a second time in a form made for this kernel (built at graph
 * load): 20 bits per id when V <= 2^20 (25 MB instead of 40 MB at 10^7 edges),
 * 32 otherwise, cut into chunks of 32 ids and laid out so that piece q of the
 * chunks of 64 consecutive lanes is one contiguous kilobyte:
 *     group of 64 chunks (2048 ids)  ->  Q pieces x 64 lanes x 16 bytes.
 * Everything the block needs is requested before anything is waited for: the
 * whole bitmap (V/8 bytes, from L2), the row pointers and flag bytes of its
 * rows, and the slice's ids, which stay in registers (two groups per wave: 65536
 * ids per block without a second round; a slice has about 39000 at 10^7 edges
 * and 256 CUs).  Then: bitmap and row pointers into LDS, one barrier, every lane
 * answers "is this holder loaded" for its 32 ids from LDS and leaves the answers
 * as one word of the edge-bit array, a second barrier, and a row's count is the
 * number of set bits in its edge range.  Used when bitmap + slice fit the
 * 160 KB of one workgroup.
 */
#ifndef K2P_EXP
#define K2P_EXP 0u /* timing experiments (diagnostic builds only): 1 no bitmap copy, 2 no id loads, 4 no lookups, 8 no row pass */
#endif
#define K2P_THREADS 1024u
#define K2P_WAVES (K2P_THREADS / 64u)
#ifndef K2P_ROWS_PRE
#define K2P_ROWS_PRE 8u /* rows per thread whose pointers and flag bytes are requested at the start */
#endif
#ifndef K2P_ROW_WEIGHT
#define K2P_ROW_WEIGHT 6u /* a row costs about as much as this many holder entries */
#endif
#define K2P_GROUP_IDS 2048u /* 64 lanes x 32 ids */

struct k2p_slice_t {
    uint32_t r0, r1; /* rows [r0, r1) */
    uint32_t e0, ne; /* first entry of the reverse CSR, number of entries */
    uint32_t g0;     /* first 2048-id group of the slice in the packed array */
    uint32_t pad[3];
};

template <uint32_t BITS> struct k2p_fmt;
template <> struct k2p_fmt<20u> {
    static constexpr uint32_t Q = 5u; /* 16-byte pieces per chunk of 32 ids */
    static constexpr uint32_t G = 2u; /* groups per wave held in registers */
};
template <> struct k2p_fmt<32u> {
    static constexpr uint32_t Q = 8u;
    static constexpr uint32_t G = 1u;
};

template <uint32_t BITS>
__global__ void __launch_bounds__(K2P_THREADS)
k2_flags_packed(const k2p_slice_t *__restrict__ slices, const uint32_t *__restrict__ rrowptr,
                const uint4 *__restrict__ rpk, const uint8_t *__restrict__ in_flags,
                const uint32_t *__restrict__ loaded_bm, uint32_t W, uint32_t ebits_words,
                uint8_t *__restrict__ out_flags, uint32_t *__restrict__ refcnt)
{
    constexpr uint32_t Q = k2p_fmt<BITS>::Q, G = k2p_fmt<BITS>::G;
    extern __shared__ __attribute__((aligned(16))) uint32_t k2_lds[];
    const uint32_t W4 = (W + 3u) & ~3u;
    uint32_t *bm = k2_lds;      /* W words, rounded up to whole 16-byte pieces */
    uint32_t *ebits = bm + W4;  /* one word per chunk of 32 entries, and one more */
    uint32_t *pre = ebits + ebits_words + 1u; /* set bits in all words before this one; one more entry as well */
    uint32_t *wtot = pre + ebits_words + 1u;  /* per wave and group of the round: set bits of its 64 words */
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const k2p_slice_t sl = slices[blockIdx.x];
    const uint32_t R = sl.r1 - sl.r0, ne = sl.ne;
    const uint32_t ngroups = (ne + K2P_GROUP_IDS - 1u) / K2P_GROUP_IDS;

    /* the ids of round `round`: groups (round * G + g) * 16 + wave */
    auto load_round = [&](uint32_t round, uint4 (&ev)[G][Q]) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t grp = (round * G + g) * K2P_WAVES + wave;
            if (grp < ngroups && !(K2P_EXP & 2u)) { /* the packed array holds whole groups: every lane may load */
                const uint4 *src = rpk + ((size_t)sl.g0 + grp) * (Q * 64u) + lane;
#pragma unroll
                for (uint32_t q = 0; q < Q; ++q)
                    ev[g][q] = src[q * 64u];
            }
        }
    };

    /* 1. everything is requested before anything is waited for.  The bitmap and the row
     * pointers go straight into LDS (loads that write LDS at a wave-wide base + lane x
     * size: no registers, no store instruction), then the flag bytes and the ids. */
    for (uint32_t i0 = wave * 64u; 4u * i0 < W4 && !(K2P_EXP & 1u); i0 += K2P_THREADS)
        if (4u * (i0 + lane) < W4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(loaded_bm + 4u * (i0 + lane)),
                                             (__attribute__((address_space(3))) void *)(bm + 4u * i0), 16, 0, 0);
    /* a thread's rows: tid, tid + 1024, ...; the first K2P_ROWS_PRE of them are asked for now */
    uint32_t fl[K2P_ROWS_PRE], ra[K2P_ROWS_PRE], rb[K2P_ROWS_PRE];
#pragma unroll
    for (uint32_t k = 0; k < K2P_ROWS_PRE; ++k) {
        const uint32_t i = tid + K2P_THREADS * k;
        const uint32_t v = sl.r0 + (i < R ? i : R - 1u); /* a slice has at least one row */
        fl[k] = in_flags[v];
        ra[k] = rrowptr[v];
        rb[k] = rrowptr[v + 1u];
    }
    uint4 ev[G][Q];
    load_round(0u, ev);
    __syncthreads(); /* waits for the loads into LDS as well */

    /* 3. a lane's 32 ids -> 32 answers -> one word of the edge-bit array, and beside it
     * the number of set bits in all words before it (block-wide scan of the words'
     * popcounts): a row's count is then a difference of two prefix values, whatever its
     * length - a module held by ten thousand others costs what a leaf costs */
    const uint32_t nchunks = (ne + 31u) >> 5;
    uint32_t round_base = 0;
    for (uint32_t round = 0;;) {
        uint32_t cnt[G], incl[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t grp = (round * G + g) * K2P_WAVES + wave;
            const uint32_t chunk = grp * 64u + lane;
            uint32_t bits = 0;
            if (grp < ngroups && chunk < nchunks) {
                uint32_t w[Q * 4u];
#pragma unroll
                for (uint32_t q = 0; q < Q; ++q) {
                    w[4u * q] = ev[g][q].x;
                    w[4u * q + 1u] = ev[g][q].y;
                    w[4u * q + 2u] = ev[g][q].z;
                    w[4u * q + 3u] = ev[g][q].w;
                }
#pragma unroll
                for (uint32_t k0 = 0; k0 < 32u; k0 += 8u) {
                    /* eight ids, eight LDS reads under way together, eight answers */
                    uint32_t id[8], word[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j) {
                        const uint32_t k = k0 + j;
                        if constexpr (BITS == 32u) {
                            id[j] = w[k];
                        } else {
                            const uint32_t pos = BITS * k, wi = pos >> 5, sh = pos & 31u;
                            id[j] = (sh + BITS <= 32u ? w[wi] >> sh : __builtin_amdgcn_alignbit(w[wi + (sh + BITS > 32u ? 1u : 0u)], w[wi], sh)) &
                                    ((1u << (BITS & 31u)) - 1u);
                        }
                        word[j] = (K2P_EXP & 4u) ? id[j] : bm[id[j] >> 5];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j)
                        bits |= ((word[j] >> (id[j] & 31u)) & 1u) << (k0 + j);
                }
                const uint32_t left = ne - 32u * chunk; /* entries of this chunk that exist */
                if (left < 32u)
                    bits &= (1u << left) - 1u;
                ebits[chunk] = bits;
            }
            cnt[g] = (uint32_t)__popc(bits);
            uint32_t x = cnt[g];
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t o = __shfl_up(x, d, 64);
                if (lane >= d)
                    x += o;
            }
            incl[g] = x;
            if (lane == 63u)
                wtot[g * K2P_WAVES + wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < K2P_WAVES; ++k) {
                const uint32_t t = wtot[g * K2P_WAVES + k];
                total += t;
                before += k < wave ? t : 0u;
            }
            const uint32_t chunk = ((round * G + g) * K2P_WAVES + wave) * 64u + lane;
            if (chunk < nchunks)
                pre[chunk] = round_base + before + incl[g] - cnt[g];
            round_base += total;
        }
        ++round;
        if (round * G * K2P_WAVES >= ngroups)
            break;
        __syncthreads(); /* the wave totals are rewritten */
        load_round(round, ev); /* slices of more than 65536 entries: a hub's row */
    }
    if (tid == 0u) {
        pre[nchunks] = round_base; /* a row may end where the last word ends */
        ebits[nchunks] = 0u;
    }
    __syncthreads();

    /* 4. rows: loaded holders of a row = set bits before its end - set bits before its start */
    auto before = [&](uint32_t x) __attribute__((always_inline)) {
        return pre[x >> 5] + (uint32_t)__popc(ebits[x >> 5] & ((1u << (x & 31u)) - 1u));
    };
    auto emit = [&](uint32_t v, uint32_t f, uint32_t h) __attribute__((always_inline)) {
        uint32_t o = 0;
        if (f & MCX_F_LOADED) {
            if (h)
                o |= MCX_O_HAS_HOLDERS;
            if ((f & MCX_F_BOUND) || h)
                o |= MCX_O_IN_USE;
            if ((f & MCX_F_SAFE) && !h)
                o |= MCX_O_ROOT;
        }
        out_flags[v] = (uint8_t)o;
        if (refcnt)
            refcnt[v] = (f & MCX_F_LOADED) ? h + ((f & MCX_F_BOUND) ? 1u : 0u) : 0xFFFFFFFFu;
    };
    if (K2P_EXP & 8u)
        return;
    uint32_t h[K2P_ROWS_PRE];
#pragma unroll
    for (uint32_t k = 0; k < K2P_ROWS_PRE; ++k)
        h[k] = tid + K2P_THREADS * k < R ? before(rb[k] - sl.e0) - before(ra[k] - sl.e0) : 0u;
#pragma unroll
    for (uint32_t k = 0; k < K2P_ROWS_PRE; ++k)
        if (tid + K2P_THREADS * k < R)
            emit(sl.r0 + tid + K2P_THREADS * k, fl[k], h[k]);
    /* slices of more rows than that (modules nobody needs, in long runs) */
    for (uint32_t i = tid + K2P_THREADS * K2P_ROWS_PRE; i < R; i += K2P_THREADS) {
        const uint32_t v = sl.r0 + i;
        const uint32_t f = in_flags[v], a = rrowptr[v], b = rrowptr[v + 1u];
        emit(v, f, before(b - sl.e0) - before(a - sl.e0));
    }
}

/* ------------------------------------------------------------------------- */
/* K2b: level-synchronous frontier BFS.                                       */
/* ------------------------------------------------------------------------- */

/*
 * A frontier entry is a module's adjacency list, (first entry, end), not its
 * number: whoever finds a module asks for its two row pointers in the same
 * round trip as the claim, so that expanding it later starts at the list.
 * `avail` has a bit per module: loaded and not reached yet (built by the seed
 * kernel from the flag bytes); a claim is one atomic AND on it, the result
 * bitmap gets its bit by an atomic OR nobody waits for.
 *
 * Counters of a search: cnt[2 * level] = entries in the frontier of that
 * level, cnt[2 * level + 1] = hubs of that level (lists of more than K2B_HUB
 * entries, kept in a list of their own), then one 64-bit count of adjacency
 * entries scanned, the levels the one-workgroup kernel ran, and whether it
 * handed over.  All zero between calls: the kernel that reports the end of a
 * search clears them.
 */
#define K2B_HUB 256u   /* longer adjacency lists are cut into pieces ...      */
#define K2B_PIECE 512u /* ... of this many entries, dealt out over the grid  */

struct k2b_state_t {
    uint32_t *cnt;
    unsigned long long *edges;
    uint32_t *avail;   /* bit v = v is loaded and not yet in the result */
    uint2 *hubq[2];    /* hubs of the even / odd levels */
};

/* words of BFS state: the per-level counters, the edge count, and what the one-workgroup
 * kernel leaves for the host: levels it ran, 1 if it handed over */
#define K2B_STATE_WORDS (2u * (K2B_MAX_LEVELS + 1u) + 4u)
#define K2B_W_EDGES (2u * (K2B_MAX_LEVELS + 1u))
#define K2B_W_SMALL_LEVELS (K2B_W_EDGES + 2u)
#define K2B_W_HANDOVER (K2B_W_EDGES + 3u)

/* what a kernel that reports to the host writes, straight into pinned host memory */
struct k2b_report_t {
    uint32_t *sizes;            /* queue and hub list sizes of a run of levels */
    unsigned long long *edges;
    uint32_t *small;            /* levels of the one-workgroup kernel, handed over */
    uint32_t *seq;              /* written last */
    uint32_t *hist;             /* K2B_BATCH words (may be null): the queue sizes of the batch's levels in front of
                                   the launch that ended the search */
};

__global__ void __launch_bounds__(K2_THREADS)
k2b_seed(const uint32_t *__restrict__ seed, const uint8_t *__restrict__ in_flags, uint32_t aligned16, uint32_t V,
         const uint32_t *__restrict__ rowptr, uint32_t *__restrict__ out, uint2 *__restrict__ queue, k2b_state_t st)
{
    /* one thread per bitmap word: LOADED bits of its 32 modules, the result word = the seed word
     * restricted to loaded modules, written whole, so neither bitmap needs clearing */
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t W = (V + 31u) / 32u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t keep = 0;
    if (w < W) {
        const uint32_t v0 = 32u * w;
        uint32_t loaded = 0;
        if (aligned16 && v0 + 32u <= V) {
            const uint4 *src = (const uint4 *)(in_flags + v0);
            const uint4 lo = src[0], hi = src[1];
            const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                const uint32_t m = q[k] & 0x01010101u;
                loaded |= (((m * 0x10204080u) >> 28) & 0xFu) << (4u * k);
            }
        } else {
            for (uint32_t k = 0; k < 32u && v0 + k < V; ++k)
                loaded |= (uint32_t)(in_flags[v0 + k] & MCX_F_LOADED) << k;
        }
        keep = seed[w] & loaded;
        out[w] = keep;
        st.avail[w] = loaded & ~keep;
    }
    /* seeds with long lists go to the hub list (rare), the others to the queue */
    uint32_t bits = keep;
    while (bits) {
        const uint32_t b = (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1u;
        const uint32_t k0 = rowptr[32u * w + b], k1 = rowptr[32u * w + b + 1u];
        if (k1 - k0 > K2B_HUB) {
            keep &= ~(1u << b);
            st.hubq[0][atomicAdd(&st.cnt[1], 1u)] = make_uint2(k0, k1);
        }
    }
    /* one queue reservation per wave: exclusive scan of the per-lane counts */
    const uint32_t mine = (uint32_t)__popc(keep);
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d)
            incl += o;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0u)
        return;
    uint32_t at = 0;
    if (lane == 0)
        at = atomicAdd(&st.cnt[0], total);
    at = __shfl(at, 0, 64) + incl - mine;
    while (keep) {
        const uint32_t b = (uint32_t)__builtin_ctz(keep);
        keep &= keep - 1u;
        queue[at++] = make_uint2(rowptr[32u * w + b], rowptr[32u * w + b + 1u]);
    }
}

/*
 * One level, edge-parallel.  A wave takes up to 64 frontier entries, lays their
 * adjacency lists end to end (exclusive scan of the lengths, kept in LDS) and
 * walks that concatenation 256 entries a step, four per lane: the owner of
 * entry j is found by bisection in the 64 scan values; the four neighbour ids
 * of a lane, then a look at their `avail` words, then the claims and row
 * pointers of the open ones, are each requested together, so a step costs three
 * memory round trips whatever the lengths are.  New
 * entries wait in a per-wave LDS buffer and are appended to the next frontier
 * with one queue reservation per buffer.
 *
 * LOOK = false leaves the look out: a level of a few thousand entries is nothing but
 * round trips, and the look is one of them (a claim that fails costs an atomic, the row
 * pointers asked for in vain two loads).  The host chooses by the size the level had in
 * the last search of this direction, and only towards the holders: towards the
 * dependencies the look is what keeps a thousand blocks from queueing up at the words of
 * the modules everything needs.
 *
 * A hub (10^4 holders and more are common: the core modules everything needs)
 * would keep its wave busy long after the rest of the level is done.  A newly
 * found hub goes to the hub list of the next level instead of its frontier,
 * and every level first walks its frontier and then the lists of its hubs cut
 * into pieces of K2B_PIECE entries, piece g going to wave g of the grid.
 */
#define K2B_PER_LANE 4u
#define K2B_BUF 512u
#ifndef K2B_LOOK_MAX
#define K2B_LOOK_MAX 8192u /* frontier entries up to which a level towards the holders claims without a look (0: never) */
#define K2B_LOOK_PER 128u  /* ... and no more than one entry per so many modules: claims on one cache line (512 modules)
                              are served one after the other, and the smaller the graph the more of a level's claims
                              share a line (6000 modules, 4 seeds: 79 us without the look, 61 with) */
#endif
#ifndef K2B_GMIN
#define K2B_GMIN 16u
#endif
#ifndef K2B_SPREAD
#define K2B_SPREAD 256u /* blocks a short frontier is spread over before a block's share drops below 64 */
#endif

/* running sums over the 64 lanes of a wave by the data-parallel primitives: four steps inside
 * the rows of sixteen lanes (lanes without a lane that far to their left add 0), then the last
 * lane of row 0 to row 1 and of row 2 to row 3, then the last lane of row 1 to rows 2 and 3 */
__device__ __forceinline__ uint32_t k2_wave_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

/* append a wave's buffered entries to the next frontier: one reservation */
__device__ __forceinline__ void k2b_flush(const uint2 *buf, uint32_t n, uint2 *__restrict__ next, uint32_t *nsize,
                                          uint32_t lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t at = 0;
    if (lane == 0)
        at = atomicAdd(nsize, n);
    at = __shfl(at, 0, 64);
    for (uint32_t i = lane; i < n; i += 64u)
        next[at + i] = buf[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* Diagnostic build (-DK2B_STAMPS, scripts/k2stamps.py; never shipped): where the time of a level
 * goes.  Thread 0 of one block of k2b_expand (by counter-window level) and thread 0 of k2b_small
 * (by level run in that launch; the last launch that ran levels stays) leave 100 MHz clock values
 * at the places marked below, each behind a wait for the thread's outstanding memory operations. */
#ifdef K2B_STAMPS
#ifndef K2B_STAMP_BLOCK
#define K2B_STAMP_BLOCK 0
#endif
__device__ unsigned long long k2b_stamp[64][16];
__device__ unsigned long long k2s_stamp[32][8];
#define K2_STAMP_NOW(cell)                                                                 \
    do {                                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        (cell) = wall_clock64();                                                           \
    } while (0)
#define K2B_STAMP(i)                                                                       \
    do {                                                                                   \
        if (blockIdx.x == K2B_STAMP_BLOCK && threadIdx.x == 0 && level < 64u)              \
            K2_STAMP_NOW(k2b_stamp[level][i]);                                             \
    } while (0)
/* in k2b_small: `at` = the level's number within the launch */
#define K2S_STAMP(at, i)                                                                   \
    do {                                                                                   \
        if (threadIdx.x == 0 && (at) < 32u)                                                \
            K2_STAMP_NOW(k2s_stamp[at][i]);                                                \
    } while (0)
#define K2S_NOTE(at, i, value)                                                             \
    do {                                                                                   \
        if (threadIdx.x == 0 && (at) < 32u)                                                \
            k2s_stamp[at][i] = (value);                                                    \
    } while (0)
#else
#define K2B_STAMP(i)
#define K2S_STAMP(at, i)
#define K2S_NOTE(at, i, value)
#endif

template <bool LOOK>
__global__ void __launch_bounds__(K2_THREADS)
k2b_expand(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
           const uint2 *__restrict__ frontier, uint32_t *__restrict__ out, uint2 *__restrict__ next,
           k2b_state_t st, uint32_t level, uint32_t sweep_waves, uint32_t qcap, uint32_t dedup)
{
    __shared__ uint32_t pre[K2_THREADS / 64][65];
    __shared__ uint32_t beg[K2_THREADS / 64][64];
    __shared__ uint2 buf[K2_THREADS / 64][K2B_BUF];
    __shared__ uint16_t hid[K2_THREADS / 64][256];
    __shared__ uint32_t hmod[K2_THREADS / 64][256];
    __shared__ uint32_t wn[K2_THREADS / 64 + 1];
    __shared__ unsigned long long wscan[K2_THREADS / 64];
    uint32_t nbuf = 0; /* the same in all lanes of a wave */
    K2B_STAMP(0);
    /* the entries this block takes if the level turns out to be a short one (K2B_GMIN a block),
     * asked for together with the level's size instead of behind it: one round trip less in
     * the levels that are nothing but round trips */
    const uint32_t lane_early = threadIdx.x & 63u;
    uint2 ent_early = make_uint2(0u, 0u), ent_early64 = make_uint2(0u, 0u);
    if (lane_early < K2B_GMIN && blockIdx.x * K2B_GMIN + lane_early < qcap)
        ent_early = frontier[blockIdx.x * K2B_GMIN + lane_early];
    if (blockIdx.x * 64u + lane_early < qcap) /* and if it is a long one */
        ent_early64 = frontier[blockIdx.x * 64u + lane_early];
    const uint32_t n = st.cnt[2u * level], nh = st.cnt[2u * level + 1u];
    if (n == 0u && nh == 0u)
        return;
    K2B_STAMP(1);
    uint32_t *nsize = &st.cnt[2u * level + 2u], *nhub = &st.cnt[2u * level + 3u];
    uint32_t *avail = st.avail;
    uint2 *nexthubs = st.hubq[(level + 1u) & 1u];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long scanned = 0;

    /* the neighbours u[] of one step: claim, row pointers, result bit, buffer */
    auto visit = [&](const uint32_t (&u)[K2B_PER_LANE], const bool (&have)[K2B_PER_LANE])
                     __attribute__((always_inline)) {
        bool cand[K2B_PER_LANE];
        uint32_t old[K2B_PER_LANE], rb[K2B_PER_LANE], re[K2B_PER_LANE];
        /* a look first: most neighbours of a level have been reached already, and the few
         * modules everything needs would otherwise take an atomic from every lane that meets
         * them.  A plain load: a word this CU's L1 still holds as it was shows bits that are
         * gone, which costs a claim that fails, never a module. */
        uint32_t w[K2B_PER_LANE];
        K2B_STAMP(3);
        if (LOOK) {
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q)
                w[q] = have[q] ? avail[u[q] >> 5] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q)
                cand[q] = ((w[q] >> (u[q] & 31u)) & 1u) != 0u;
        } else {
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q)
                cand[q] = have[q];
        }
        /* At the start of a level every wave that meets one of the modules everything needs
         * still sees it open, and their claims on one word are served one after the other.  So
         * a wave claims a module once: its open entries meet in a small hash table in LDS (the
         * last writer of a slot owns it and leaves its module there; an entry that finds its
         * own module under another owner is dropped).  Only towards the dependencies (`dedup`):
         * towards the holders no module is everybody's neighbour, and the three wave barriers
         * of the table are half a microsecond of a level that lasts five. */
        K2B_STAMP(4);
        if (dedup && __ballot(cand[0] | cand[1] | cand[2] | cand[3])) {
            uint32_t slot[K2B_PER_LANE];
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q) {
                slot[q] = (u[q] * 0x9E3779B1u) >> 24;
                if (cand[q])
                    hid[wave][slot[q]] = (uint16_t)(64u * q + lane);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            bool owner[K2B_PER_LANE];
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q) {
                owner[q] = cand[q] && hid[wave][slot[q]] == (uint16_t)(64u * q + lane);
                if (owner[q])
                    hmod[wave][slot[q]] = u[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q)
                if (cand[q] && !owner[q] && hmod[wave][slot[q]] == u[q])
                    cand[q] = false;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        K2B_STAMP(5);
#pragma unroll
        for (uint32_t q = 0; q < K2B_PER_LANE; ++q) {
            old[q] = cand[q] ? atomicAnd(&avail[u[q] >> 5], ~(1u << (u[q] & 31u))) : 0u;
            rb[q] = cand[q] ? rowptr[u[q]] : 0u;
            re[q] = cand[q] ? rowptr[u[q] + 1u] : 0u;
        }
        K2B_STAMP(6);
#pragma unroll
        for (uint32_t q = 0; q < K2B_PER_LANE; ++q) {
            const uint32_t bit = 1u << (u[q] & 31u);
            bool got = (old[q] & bit) != 0u;
            if (got)
                atomicOr(&out[u[q] >> 5], bit); /* nobody waits for it */
            if (got && re[q] - rb[q] > K2B_HUB) { /* rare */
                nexthubs[atomicAdd(nhub, 1u)] = make_uint2(rb[q], re[q]);
                got = false;
            }
            const unsigned long long m = __ballot(got);
            if (m) {
                if (got)
                    buf[wave][nbuf + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                        make_uint2(rb[q], re[q]);
                nbuf += (uint32_t)__popcll(m);
            }
        }
        if (nbuf > K2B_BUF - 64u * K2B_PER_LANE) {
            k2b_flush(buf[wave], nbuf, next, nsize, lane);
            nbuf = 0;
        }
    };

    /* The frontier, `grp` entries per BLOCK and round: 64 when there is enough of it for
     * K2B_SPREAD blocks, fewer (down to K2B_GMIN) when there is not, so that a short frontier
     * still spreads over that many blocks (every block with work costs the level a queue
     * reservation, served one after the other: thinner than this, a level got slower).  The four waves of a block all take the same entries (the
     * second to fourth read them from L1), lay their lists end to end and share the steps of
     * that concatenation among them: the lists' lengths differ by two orders of magnitude, and
     * a level ends with its slowest wave - a wave alone with sixteen long lists took eight
     * steps where its neighbours took one. */
    const uint32_t nblocks_all = min(gridDim.x, max(sweep_waves / (K2_THREADS / 64u), 1u));
    uint32_t grp = 64u;
    while (grp > K2B_GMIN && (grp >> 1) * min(nblocks_all, K2B_SPREAD) >= n)
        grp >>= 1;
    for (uint32_t base = blockIdx.x < nblocks_all ? blockIdx.x * grp : n; base < n; base += nblocks_all * grp) {
        const uint32_t t = base + lane;
        uint32_t k0 = 0, k1 = 0;
        if (lane < grp && t < n) {
            const uint2 ent = (grp == K2B_GMIN && base == blockIdx.x * K2B_GMIN) ? ent_early
                              : (grp == 64u && base == blockIdx.x * 64u)         ? ent_early64
                                                                                 : frontier[t];
            k0 = ent.x;
            k1 = ent.y;
        }
        K2B_STAMP(2);
        const uint32_t deg = k1 - k0;
        if (wave == 0u)
            scanned += deg; /* once per block */
        const uint32_t incl = k2_wave_incl(deg);
        const uint32_t total = __shfl(incl, 63, 64);
        pre[wave][lane] = incl - deg;
        beg[wave][lane] = k0;
        if (lane == 63u)
            pre[wave][64] = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t j0 = 64u * K2B_PER_LANE * wave; j0 < total; j0 += K2_THREADS * K2B_PER_LANE) {
            uint32_t u[K2B_PER_LANE];
            bool have[K2B_PER_LANE];
#pragma unroll
            for (uint32_t q = 0; q < K2B_PER_LANE; ++q) {
                const uint32_t j = j0 + 64u * q + lane;
                have[q] = j < total;
                uint32_t lo = 0, hi = 64; /* largest i with pre[i] <= j */
#pragma unroll
                for (uint32_t step = 0; step < 6u; ++step) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pre[wave][mid] <= j)
                        lo = mid;
                    else
                        hi = mid;
                }
                u[q] = have[q] ? col[beg[wave][lo] + (j - pre[wave][lo])] : 0u;
            }
            visit(u, have);
        }
        /* pre / beg are rewritten by the next round */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    /* the hubs of this level, 64 at a time: piece g of their lists laid end to
     * end goes to wave g, counted from the far end of the grid (the near end
     * holds the frontier when that is short) */
    if (nh) {
        const uint2 *hubs = st.hubq[level & 1u];
        const uint32_t nwaves = gridDim.x * (K2_THREADS / 64u);
        const uint32_t gwave = nwaves - 1u - (blockIdx.x * (K2_THREADS / 64u) + wave);
        for (uint32_t h0 = 0; h0 < nh; h0 += 64u) {
            uint32_t k0 = 0, k1 = 0;
            if (h0 + lane < nh) {
                const uint2 ent = hubs[h0 + lane];
                k0 = ent.x;
                k1 = ent.y;
            }
            if (gwave == 0u)
                scanned += k1 - k0;
            const uint32_t np = (k1 - k0 + K2B_PIECE - 1u) / K2B_PIECE;
            uint32_t incl = np;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d)
                    incl += o;
            }
            const uint32_t total = __shfl(incl, 63, 64);
            const uint32_t excl = incl - np;
            for (uint32_t g = gwave; g < total; g += nwaves) {
                /* lanes without a hub have excl = total > g */
                const uint32_t idx = (uint32_t)__popcll(__ballot(np != 0u && excl <= g)) - 1u;
                const uint32_t ks = __shfl(k0, idx, 64) + (g - __shfl(excl, idx, 64)) * K2B_PIECE;
                const uint32_t ke = min(ks + K2B_PIECE, __shfl(k1, idx, 64));
                for (uint32_t j0 = ks; j0 < ke; j0 += 64u * K2B_PER_LANE) {
                    uint32_t u[K2B_PER_LANE];
                    bool have[K2B_PER_LANE];
#pragma unroll
                    for (uint32_t q = 0; q < K2B_PER_LANE; ++q) {
                        const uint32_t j = j0 + 64u * q + lane;
                        have[q] = j < ke;
                        u[q] = have[q] ? col[j] : 0u;
                    }
                    visit(u, have);
                }
            }
        }
    }
    /* what the waves still hold goes out with one queue reservation and one add to the entry
     * count per block: a few thousand atomics on one word would take longer than the level
     * (they are served one after the other, some 12 ns each) */
    K2B_STAMP(7);
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1)
        scanned += __shfl_down(scanned, d, 64);
    if (lane == 0u) {
        wn[wave] = nbuf;
        wscan[wave] = scanned;
    }
    __syncthreads();
    K2B_STAMP(8);
    if (threadIdx.x == 0u) {
        uint32_t tot = 0;
        unsigned long long sc = 0;
#pragma unroll
        for (uint32_t k = 0; k < K2_THREADS / 64u; ++k) {
            const uint32_t c = wn[k];
            wn[k] = tot;
            tot += c;
            sc += wscan[k];
        }
        wn[K2_THREADS / 64u] = tot ? atomicAdd(nsize, tot) : 0u;
        if (sc)
            atomicAdd(st.edges, sc);
    }
    __syncthreads();
    K2B_STAMP(9);
    const uint32_t at = wn[K2_THREADS / 64u] + wn[wave];
    for (uint32_t i = lane; i < nbuf; i += 64u)
        next[at + i] = buf[wave][i];
    K2B_STAMP(10);
}

/*
 * Closures of a few modules have frontiers of a few hundred entries for most of
 * their levels: a launch per level is then all boundary and round-trip latency.
 * One workgroup of 1024 threads runs such levels back to back: the frontier
 * lives in LDS (two queues of K2S_CAP entries), levels are separated by
 * __syncthreads().  A level is the concatenation of its adjacency lists, walked
 * K2S_THREADS x K2S_PER entries a step; per entry the claim and the neighbour's
 * row pointers are requested together.  The kernel is enqueued at level `level`
 * of the counter window, in front of that level's own launch, and takes over
 * when the frontier there is short and has no hub; it stops in front of a level
 * that outgrows the workgroup (more than K2S_CAP entries, or more than
 * K2S_LEVEL_MAX adjacency entries: faster on the whole chip) and leaves that
 * frontier as level `level` again, queue and hub list in the form the per-level
 * kernel expects.  So a search whose levels are small, then large, then small
 * runs as: this kernel, a few per-level launches, this kernel.
 * With `rep.seq` set it reports to the host when it returns (and clears the
 * state if the search is over).
 */
#ifndef K2B_TAIL
#define K2B_TAIL 2u        /* levels a batch is longer than the last search in that direction */
#endif
#ifndef K2B_SMALL_EVERY
#define K2B_SMALL_EVERY 2u /* the one-workgroup kernel goes in front of every so many levels */
#endif
#define K2S_THREADS 1024u
#ifndef K2S_TRY_N
#define K2S_TRY_N 600u /* frontier entries up to which a level is worth a try of the one-workgroup kernel */
#endif
#define K2S_HASH_BITS 12u
#define K2S_HASH (1u << K2S_HASH_BITS) /* slots of the one-workgroup kernel's claim table: one per entry of a step */
#define K2S_CAP 2048u
#define K2S_PER 4u
#ifndef K2S_LEVEL_MAX
#define K2S_LEVEL_MAX 4096u
#endif
#ifndef K2S_LOOK_MAX
#define K2S_LOOK_MAX 1024u /* adjacency entries of a level up to which it claims without a look (0: never) */
#endif

__global__ void __launch_bounds__(K2S_THREADS)
k2b_small(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col, uint32_t *__restrict__ out,
          uint2 *__restrict__ queue, uint2 *__restrict__ spill, k2b_state_t st, uint32_t level, k2b_report_t rep,
          uint32_t seq, uint32_t only_the_end, uint32_t batch_base, uint32_t dedup)
{
    __shared__ uint16_t hid[K2S_HASH];  /* dedup: who asked last for the module that hashes here */
    __shared__ uint32_t hmod[K2S_HASH]; /* ... and which module that was */
    __shared__ uint32_t qb[2][K2S_CAP], qe[2][K2S_CAP]; /* list start, list end */
    __shared__ uint32_t pre[K2S_CAP + 1u];               /* exclusive sums of the list lengths */
    __shared__ uint32_t wsum[K2S_THREADS / 64u];
    __shared__ uint32_t nnext_sh, nq_sh, nh_sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t n = st.cnt[2u * level];
    const uint32_t nh0 = st.cnt[2u * level + 1u];
    uint32_t *avail = st.avail;
    unsigned long long scanned = 0; /* thread 0 keeps the sum */
    uint32_t levels = 0, cur = 0;
    uint32_t left_q = n, left_h = nh0; /* what this level holds when the kernel returns (thread 0) */

    /* leave the frontier held in LDS queue `c` (first `nl` entries) plus `ns` spilled entries
     * as this level's: hubs to the hub list, the rest to the queue */
    auto hand_over = [&](uint32_t c, uint32_t nl, uint32_t ns) {
        if (tid == 0) {
            nq_sh = 0;
            nh_sh = 0;
        }
        __syncthreads();
        for (uint32_t i = tid; i < nl + ns; i += K2S_THREADS) {
            const uint2 ent = i < nl ? make_uint2(qb[c][i], qe[c][i]) : spill[i - nl];
            if (ent.y - ent.x > K2B_HUB)
                st.hubq[level & 1u][atomicAdd(&nh_sh, 1u)] = ent;
            else
                queue[atomicAdd(&nq_sh, 1u)] = ent;
        }
        __syncthreads();
        left_q = nq_sh;
        left_h = nh_sh;
    };

    const bool mine = nh0 == 0u && n != 0u && n <= K2S_CAP;
    if (mine) {
        for (uint32_t i = tid; i < n; i += K2S_THREADS) {
            const uint2 ent = queue[i];
            qb[0][i] = ent.x;
            qe[0][i] = ent.y;
        }
        __syncthreads();
        while (n != 0u) {
            K2S_STAMP(levels, 0);
            /* exclusive sums of the list lengths: K2S_CAP / K2S_THREADS = 2 entries per thread */
            uint32_t d[K2S_CAP / K2S_THREADS], sum = 0;
#pragma unroll
            for (uint32_t k = 0; k < K2S_CAP / K2S_THREADS; ++k) {
                const uint32_t i = (K2S_CAP / K2S_THREADS) * tid + k;
                d[k] = i < n ? qe[cur][i] - qb[cur][i] : 0u;
                sum += d[k];
            }
            const uint32_t incl = k2_wave_incl(sum);
            if (lane == 63u)
                wsum[wave] = incl;
            if (tid == 0)
                nnext_sh = 0;
            __syncthreads();
            uint32_t at = incl - sum, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < K2S_THREADS / 64u; ++k) {
                if (k < wave)
                    at += wsum[k];
                total += wsum[k];
            }
#pragma unroll
            for (uint32_t k = 0; k < K2S_CAP / K2S_THREADS; ++k) {
                const uint32_t i = (K2S_CAP / K2S_THREADS) * tid + k;
                if (i <= n)
                    pre[i] = at;
                at += d[k];
            }
            __syncthreads();
            if (total > K2S_LEVEL_MAX) {
                hand_over(cur, n, 0u);
                break;
            }
            K2S_STAMP(levels, 1);
            K2S_NOTE(levels, 6, n);
            K2S_NOTE(levels, 7, total);
            ++levels;
            scanned += total;
            const uint32_t nxt = cur ^ 1u;
            for (uint32_t j0 = 0; j0 < total; j0 += K2S_THREADS * K2S_PER) {
                uint32_t u[K2S_PER], old[K2S_PER], b[K2S_PER], e[K2S_PER];
                bool have[K2S_PER];
                /* largest i with pre[i] <= j, for the thread's four entries side by side: the same
                 * number of halvings for all (from n), so that the four chains of dependent LDS
                 * reads overlap instead of following each other (eleven reads each at n = 2048) */
                uint32_t lo[K2S_PER], hi[K2S_PER];
#pragma unroll
                for (uint32_t q = 0; q < K2S_PER; ++q) {
                    have[q] = j0 + K2S_THREADS * q + tid < total;
                    lo[q] = 0u;
                    hi[q] = n;
                }
                for (uint32_t span = n; span > 1u; span = (span + 1u) >> 1) {
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q) {
                        const uint32_t mid = (lo[q] + hi[q]) >> 1;
                        const bool go = hi[q] - lo[q] > 1u;
                        const bool up = pre[go ? mid : lo[q]] <= j0 + K2S_THREADS * q + tid;
                        lo[q] = go && up ? mid : lo[q];
                        hi[q] = go && !up ? mid : hi[q];
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < K2S_PER; ++q) {
                    const uint32_t j = j0 + K2S_THREADS * q + tid;
                    u[q] = have[q] ? col[qb[cur][lo[q]] + (j - pre[lo[q]])] : 0u;
                }
                K2S_STAMP(levels - 1u, 2);
                bool cand[K2S_PER];
                if (dedup != 0u || K2S_LOOK_MAX == 0u || total > K2S_LOOK_MAX) {
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q) /* a look first (see k2b_expand) */
                        old[q] = have[q] ? __hip_atomic_load(&avail[u[q] >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    K2S_STAMP(levels - 1u, 3);
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q)
                        cand[q] = ((old[q] >> (u[q] & 31u)) & 1u) != 0u;
                } else { /* a short level towards the holders: no look (see k2b_expand) */
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q)
                        cand[q] = have[q];
                }
                if (dedup) {
                    /* towards the dependencies the lists of a level name the same few modules over
                     * and over; all of a step's looks see such a module open, and their claims on
                     * its word would be served one after the other (3000 entries: 6 us).  One
                     * entry per module claims: the last writer of a hash slot owns it and leaves
                     * its module there, an entry that finds its own module under another owner is
                     * dropped (k2b_expand does the same per wave). */
                    uint32_t slot[K2S_PER];
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q) {
                        slot[q] = (u[q] * 0x9E3779B1u) >> (32u - K2S_HASH_BITS);
                        if (cand[q])
                            hid[slot[q]] = (uint16_t)(K2S_PER * tid + q);
                    }
                    __syncthreads();
                    bool owner[K2S_PER];
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q) {
                        owner[q] = cand[q] && hid[slot[q]] == (uint16_t)(K2S_PER * tid + q);
                        if (owner[q])
                            hmod[slot[q]] = u[q];
                    }
                    __syncthreads();
#pragma unroll
                    for (uint32_t q = 0; q < K2S_PER; ++q)
                        if (cand[q] && !owner[q] && hmod[slot[q]] == u[q])
                            cand[q] = false;
                }
#pragma unroll
                for (uint32_t q = 0; q < K2S_PER; ++q) {
                    const bool open = cand[q];
                    old[q] = open ? atomicAnd(&avail[u[q] >> 5], ~(1u << (u[q] & 31u))) : 0u;
                    b[q] = open ? rowptr[u[q]] : 0u;
                    e[q] = open ? rowptr[u[q] + 1u] : 0u;
                }
                K2S_STAMP(levels - 1u, 4);
#pragma unroll
                for (uint32_t q = 0; q < K2S_PER; ++q) {
                    /* the wave's new entries numbered by ballot: one add to the level's count per wave
                     * (an add per lane to one LDS word is served lane after lane) */
                    const uint32_t bit = 1u << (u[q] & 31u);
                    const bool got = (old[q] & bit) != 0u;
                    const unsigned long long m = __ballot(got);
                    if (m == 0ull)
                        continue;
                    uint32_t first = 0;
                    if (lane == 0u)
                        first = atomicAdd(&nnext_sh, (uint32_t)__popcll(m));
                    first = __shfl(first, 0, 64);
                    if (!got)
                        continue;
                    atomicOr(&out[u[q] >> 5], bit);
                    const uint32_t at2 = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (at2 < K2S_CAP) {
                        qb[nxt][at2] = b[q];
                        qe[nxt][at2] = e[q];
                    } else {
                        spill[at2 - K2S_CAP] = make_uint2(b[q], e[q]); /* the other level's queue: idle now */
                    }
                }
            }
            __syncthreads();
            K2S_STAMP(levels - 1u, 5);
            const uint32_t nn = nnext_sh;
            if (nn > K2S_CAP) {
                /* the spilled entries were written by this block: make them visible to its own reads */
                __threadfence();
                __syncthreads();
                hand_over(nxt, K2S_CAP, nn - K2S_CAP);
                n = nn;
                break;
            }
            n = nn;
            cur = nxt;
            left_q = nn; /* zero when the search ends here */
            left_h = 0;
            __syncthreads(); /* nnext_sh is reset at the top of the next level */
        }
    }
    /* record: this level's sizes as they stand now, the levels run here, the entries scanned */
    __syncthreads();
    if (tid == 0) {
        st.cnt[2u * level] = left_q;
        st.cnt[2u * level + 1u] = left_h;
        const uint32_t small_levels = st.cnt[K2B_W_SMALL_LEVELS] + levels;
        st.cnt[K2B_W_SMALL_LEVELS] = small_levels;
        const uint32_t handed = (left_q != 0u || left_h != 0u) ? 1u : 0u;
        st.cnt[K2B_W_HANDOVER] = handed;
        unsigned long long edges = scanned ? atomicAdd(st.edges, scanned) + scanned : *st.edges;
        /* only_the_end: one of a batch of launches - it reports (into words of its own) if the
         * search ends here, with the level it was launched for and the sequence number twice, so
         * that a host that looks later, when the batch's own report has come in, still sees
         * who ended the search */
        /* (a launch that found the search over already leaves the word to the batch's report) */
        const bool tell = seq != 0u && (only_the_end == 0u || (handed == 0u && mine));
        if (tell) {
            if (only_the_end && rep.hist) /* how long the frontiers of this batch were: the host places the
                                             next search's one-workgroup launches by them */
                for (uint32_t l = batch_base; l < level && l - batch_base < K2B_BATCH; ++l)
                    rep.hist[l - batch_base] = st.cnt[2u * l];
            rep.sizes[0] = only_the_end ? level : left_q;
            rep.sizes[1] = only_the_end ? seq : left_h;
            *rep.edges = edges;
            rep.small[0] = small_levels;
            rep.small[1] = handed;
            __threadfence_system();
            __hip_atomic_store(rep.seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        wsum[0] = (tell && !handed) ? 1u : 0u; /* the search is over and the host has been told: clear the state */
    }
    __syncthreads();
    if (wsum[0])
        for (uint32_t i = tid; i < K2B_STATE_WORDS; i += K2S_THREADS)
            st.cnt[i] = 0u;
}

/* ------------------------------------------------------------------------- */
/* LOADED bitmap of mcx_dep_flags_noalloc_dev: a device array of the code      */
/* object, so that the call needs no allocation.  Every GPU the library runs   */
/* on holds its own copy at its own address; in host code the array's name is  */
/* only a handle for hipGetSymbolAddress, never a device address.              */
/* ------------------------------------------------------------------------- */
static __device__ __align__(256) uint32_t scratch[MCX_NOALLOC_MAX_MODULES / 32u];

#define K2_MAX_DEVICES 64
static std::mutex g_scratch_lock;
static uint32_t *g_scratch_ptr[K2_MAX_DEVICES]; /* device address of `scratch`, per device ordinal */

/* looked up once per device, with that device current (mcx_ctx_create calls this;
 * the entry point only reads the table) */
int mcx_k2_static_init(mcx_ctx *ctx)
{
    if (ctx->device < 0 || ctx->device >= K2_MAX_DEVICES)
        return MCX_OK; /* no static bitmap there: the entry point says so */
    std::lock_guard<std::mutex> guard(g_scratch_lock);
    if (g_scratch_ptr[ctx->device])
        return MCX_OK;
    uint32_t *p = nullptr;
    HIPCHK(ctx, hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[ctx->device] = p;
    return MCX_OK;
}

static uint32_t *k2_static_bitmap(mcx_ctx *ctx)
{
    uint32_t *p = nullptr;
    if (ctx->device >= 0 && ctx->device < K2_MAX_DEVICES) {
        std::lock_guard<std::mutex> guard(g_scratch_lock);
        p = g_scratch_ptr[ctx->device];
    }
    if (!p)
        ctx->err = "mcx_dep_flags_noalloc_dev: no static bitmap on device " + std::to_string(ctx->device);
    return p;
}

/* ------------------------------------------------------------------------- */
/* C ABI                                                                      */
/* ------------------------------------------------------------------------- */

/* dynamic LDS of k2_flags_packed: bitmap (whole 16-byte pieces), row pointers, edge bits */
static size_t k2p_lds_bytes(const mcx_ctx *ctx)
{
    const size_t W4 = (((size_t)ctx->V + 31) / 32 + 3) & ~(size_t)3;
    const size_t ebits_words = ((size_t)ctx->cu_edges_max + 31) / 32;
    return 4 * (W4 + 2 * (ebits_words + 1) + 2 * K2P_WAVES + 8);
}

extern "C" {

int mcx_dep_graph_load(mcx_ctx *ctx, const uint32_t *rowptr, const uint32_t *col, uint32_t V, uint32_t E)
{
    if (!ctx || !rowptr || (E && !col))
        return MCX_E_BAD_ARG;
    if (rowptr[0] != 0 || rowptr[V] != E) {
        ctx->err = "rowptr[0] must be 0 and rowptr[V] must be E";
        return MCX_E_BAD_ARG;
    }
    for (uint32_t u = 0; u < V; ++u)
        if (rowptr[u + 1] < rowptr[u]) {
            ctx->err = "rowptr not monotonic";
            return MCX_E_BAD_ARG;
        }
    for (uint32_t k = 0; k < E; ++k)
        if (col[k] >= V) {
            ctx->err = "col entry out of range";
            return MCX_E_BAD_ARG;
        }
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    /* reverse CSR (holders of each module) by counting sort */
    std::vector<uint32_t> rrow((size_t)V + 2, 0), rcol(E ? E : 1);
    for (uint32_t k = 0; k < E; ++k)
        rrow[(size_t)col[k] + 2]++;
    for (uint32_t v = 0; v < V; ++v)
        rrow[(size_t)v + 2] += rrow[(size_t)v + 1];
    for (uint32_t u = 0; u < V; ++u)
        for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; ++k)
            rcol[rrow[(size_t)col[k] + 1]++] = u;
    /* rrow[0..V] is now the reverse row pointer; cut it into row blocks */
    std::vector<uint32_t> part;
    part.push_back(0);
    for (uint32_t r = 0; r < V;) {
        uint32_t end = r + 1;
        while (end < V && end - r < K2_ROWS && rrow[end + 1] - rrow[r] <= K2_EDGES)
            ++end;
        part.push_back(end);
        r = end;
    }
    /* holder levels (Kahn over "u needs v": v follows all of its holders);
     * a graph with a cycle has none and only the fleet pipeline minds */
    std::vector<uint32_t> order, lvl_ptr, lvl_maxdeg;
    {
        std::vector<uint32_t> waiting((size_t)V), level((size_t)V, 0);
        order.reserve(V);
        for (uint32_t v = 0; v < V; ++v) {
            waiting[v] = rrow[(size_t)v + 1] - rrow[v];
            if (waiting[v] == 0)
                order.push_back(v);
        }
        for (size_t q = 0; q < order.size(); ++q) {
            const uint32_t u = order[q];
            for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
                const uint32_t v = col[k];
                if (level[v] < level[u] + 1u)
                    level[v] = level[u] + 1u;
                if (--waiting[v] == 0)
                    order.push_back(v);
            }
        }
        if (order.size() == (size_t)V) {
            /* Kahn's queue is not sorted by level: bucket it */
            uint32_t nl = 0;
            for (uint32_t v = 0; v < V; ++v)
                nl = level[v] + 1u > nl ? level[v] + 1u : nl;
            lvl_ptr.assign((size_t)nl + 1, 0);
            for (uint32_t v = 0; v < V; ++v)
                lvl_ptr[(size_t)level[v] + 1]++;
            for (uint32_t k = 0; k < nl; ++k)
                lvl_ptr[(size_t)k + 1] += lvl_ptr[k];
            std::vector<uint32_t> at(lvl_ptr.begin(), lvl_ptr.end() - 1);
            for (uint32_t v = 0; v < V; ++v)
                order[at[level[v]]++] = v;
            lvl_maxdeg.assign(nl, 0);
            for (uint32_t v = 0; v < V; ++v) {
                const uint32_t deg = rrow[(size_t)v + 1] - rrow[v];
                if (deg > lvl_maxdeg[level[v]])
                    lvl_maxdeg[level[v]] = deg;
            }
        } else {
            order.clear();
        }
    }
    /* slices for the LDS variant of K2a: one per CU, about E / #CU edges each, and their
     * holder ids once more, packed (see k2_flags_packed) */
    std::vector<k2p_slice_t> slices;
    std::vector<uint32_t> rpk;
    uint32_t cu_rows_max = 0, cu_edges_max = 0;
    const uint32_t pk_bits = V <= (1u << 20) ? 20u : 32u;
    {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n_cu <= 0)
            n_cu = 256;
        /* consecutive rows, cut so that no slice weighs more than `cap` (a row = its entries +
         * K2P_ROW_WEIGHT; a row heavier than that is a slice of its own): the smallest cap
         * that needs no more slices than there are CUs, so every block is resident at once */
        auto cut = [&](uint64_t cap, std::vector<k2p_slice_t> *out) {
            uint32_t n = 0;
            uint64_t groups = 0;
            for (uint32_t r = 0; r < V;) {
                uint32_t end = r;
                uint64_t load = 0;
                while (end < V) {
                    const uint64_t w = (uint64_t)(rrow[end + 1] - rrow[end]) + K2P_ROW_WEIGHT;
                    if (end != r && load + w > cap)
                        break;
                    load += w;
                    ++end;
                }
                if (out) {
                    k2p_slice_t sl = {};
                    sl.r0 = r;
                    sl.r1 = end;
                    sl.e0 = rrow[r];
                    sl.ne = rrow[end] - rrow[r];
                    sl.g0 = (uint32_t)groups;
                    groups += (sl.ne + K2P_GROUP_IDS - 1u) / K2P_GROUP_IDS;
                    out->push_back(sl);
                }
                ++n;
                r = end;
            }
            return n;
        };
        uint64_t lo = 1, hi = (uint64_t)E + (uint64_t)K2P_ROW_WEIGHT * V + 1;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (cut(mid, nullptr) <= (uint32_t)n_cu)
                hi = mid;
            else
                lo = mid + 1;
        }
        cut(lo, &slices);
        uint64_t groups = 0;
        for (const k2p_slice_t &sl : slices) {
            groups += (sl.ne + K2P_GROUP_IDS - 1u) / K2P_GROUP_IDS;
            cu_rows_max = std::max(cu_rows_max, sl.r1 - sl.r0);
            cu_edges_max = std::max(cu_edges_max, sl.ne);
        }
        const size_t lds = 4 * ((size_t)(V + 31) / 32 + 4 + 2 * ((cu_edges_max + 31) / 32 + 1) + 2 * K2P_WAVES + 8);
        if (V == 0 || lds > 160 * 1024 - 512 || groups >= (1u << 22)) {
            slices.clear(); /* does not fit one workgroup: the row-block variant is used */
        } else {
            const uint32_t Q = pk_bits == 20u ? k2p_fmt<20u>::Q : k2p_fmt<32u>::Q;
            rpk.assign((size_t)groups * Q * 256u, 0u);
            for (const k2p_slice_t &sl : slices)
                for (uint32_t j = 0; j < sl.ne; ++j) {
                    const uint32_t id = rcol[(size_t)sl.e0 + j];
                    const uint32_t chunk = j >> 5, k = j & 31u, grp = chunk >> 6, ln = chunk & 63u;
                    /* word t of the chunk sits in piece t / 4 of its lane */
                    uint32_t *gbase = rpk.data() + ((size_t)sl.g0 + grp) * Q * 256u;
                    auto word = [&](uint32_t t) -> uint32_t & { return gbase[((t >> 2) * 64u + ln) * 4u + (t & 3u)]; };
                    if (pk_bits == 32u) {
                        word(k) = id;
                    } else {
                        const uint32_t pos = 20u * k, t = pos >> 5, sh = pos & 31u;
                        word(t) |= id << sh;
                        if (sh > 12u)
                            word(t + 1u) |= id >> (32u - sh);
                    }
                }
        }
    }
    /* modules with long lists, per direction: room for the hub lists of the closure search */
    const uint32_t W = (V + 31u) / 32u;
    uint32_t n_hub[2] = {0, 0};
    for (uint32_t v = 0; v < V; ++v) {
        if (rrow[(size_t)v + 1] - rrow[v] > K2B_HUB)
            ++n_hub[MCX_CLOSURE_HOLDERS];
        if (rowptr[v + 1] - rowptr[v] > K2B_HUB)
            ++n_hub[MCX_CLOSURE_DEPS];
    }
    const uint32_t hub_cap = std::max(n_hub[0], n_hub[1]);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    mcx_free_graph(ctx);
    int rc;
    if ((rc = mcx_dev_upload(ctx, &ctx->d_rowptr, rowptr, (size_t)V + 1)) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_col, col, (size_t)E)) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_rrowptr, rrow.data(), (size_t)V + 1)) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_rcol, rcol.data(), (size_t)E)) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_part, part.data(), part.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, (k2p_slice_t **)&ctx->d_slices, slices.data(), slices.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_rpk, rpk.data(), rpk.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_lvl_order, order.data(), order.size())) != 0) {
        mcx_free_graph(ctx);
        return rc;
    }
    /* two frontier queues of V (first entry, end) pairs (the first doubles as K2a's LOADED
     * bitmap, whose copy reads whole 16-byte pieces); BFS state, the two hub lists, the bitmap
     * of loaded modules not reached yet */
    size_t vb = 8 * ((size_t)V > 4 ? V : 4);
    if (hipMalloc((void **)&ctx->d_queue[0], vb) != hipSuccess ||
        hipMalloc((void **)&ctx->d_queue[1], vb) != hipSuccess ||
        hipMalloc((void **)&ctx->d_bfs, 4 * ((size_t)K2B_STATE_WORDS + 4 * (size_t)hub_cap + (size_t)W + 4)) !=
            hipSuccess) {
        mcx_free_graph(ctx);
        ctx->err = "hipMalloc of graph scratch failed";
        return MCX_E_NOMEM;
    }
    ctx->bfs_clean = false;
    ctx->n_hub[0] = n_hub[0];
    ctx->n_hub[1] = n_hub[1];
    ctx->hub_cap = hub_cap;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->V = V;
    ctx->E = E;
    ctx->n_part = (uint32_t)part.size() - 1u;
    ctx->lvl_ptr = lvl_ptr;
    ctx->lvl_maxdeg = lvl_maxdeg;
    ctx->n_slices = (uint32_t)slices.size();
    ctx->pk_bits = pk_bits;
    ctx->cu_rows_max = cu_rows_max;
    ctx->cu_edges_max = cu_edges_max;
    if (ctx->n_slices) {
        const int lds = (int)k2p_lds_bytes(ctx);
        hipError_t ea = hipFuncSetAttribute(pk_bits == 20u ? (const void *)k2_flags_packed<20u>
                                                           : (const void *)k2_flags_packed<32u>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (ea != hipSuccess)
            ctx->n_slices = 0;
    }
    ctx->bfs_depth[0] = ctx->bfs_depth[1] = 0;
    ctx->bfs_sizes[0].clear();
    ctx->bfs_sizes[1].clear();
    ctx->graph_loaded = true;
    return MCX_OK;
}

/* both kernels of K2a with the LOADED bitmap in `bm` (V words' worth of room is not
 * needed: ceil(V / 32) words rounded up to a multiple of four are used) */
static int k2_enqueue_flags(mcx_ctx *ctx, const void *d_in_flags, void *d_out_flags, void *d_refcnt, uint32_t *bm)
{
    const uint32_t W = (ctx->V + 31u) / 32u, W4 = (W + 3u) & ~3u;
    k2_loaded_bitmap<<<(W4 + K2_THREADS - 1) / K2_THREADS, K2_THREADS, 0, ctx->stream>>>(
        (const uint8_t *)d_in_flags, ctx->V, ((uintptr_t)d_in_flags & 15u) == 0u ? 1u : 0u, bm);
    if (ctx->n_slices) {
        const size_t lds = k2p_lds_bytes(ctx);
        const k2p_slice_t *slices = (const k2p_slice_t *)ctx->d_slices;
        const uint4 *rpk = (const uint4 *)ctx->d_rpk;
        if (ctx->pk_bits == 20u)
            k2_flags_packed<20u><<<ctx->n_slices, K2P_THREADS, lds, ctx->stream>>>(
                slices, ctx->d_rrowptr, rpk, (const uint8_t *)d_in_flags, bm, W,
                (ctx->cu_edges_max + 31u) / 32u, (uint8_t *)d_out_flags, (uint32_t *)d_refcnt);
        else
            k2_flags_packed<32u><<<ctx->n_slices, K2P_THREADS, lds, ctx->stream>>>(
                slices, ctx->d_rrowptr, rpk, (const uint8_t *)d_in_flags, bm, W,
                (ctx->cu_edges_max + 31u) / 32u, (uint8_t *)d_out_flags, (uint32_t *)d_refcnt);
    } else {
        k2_flags_rows<<<ctx->n_part, K2_THREADS, 0, ctx->stream>>>(ctx->d_rrowptr, ctx->d_rcol, ctx->d_part,
                                                                  (const uint8_t *)d_in_flags, bm,
                                                                  (uint8_t *)d_out_flags, (uint32_t *)d_refcnt);
    }
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}

int mcx_dep_flags_dev(mcx_ctx *ctx, const void *d_in_flags, void *d_out_flags, void *d_refcnt)
{
    if (!ctx || !d_in_flags || !d_out_flags)
        return MCX_E_BAD_ARG;
    if (!ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    if (ctx->V == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    /* the bitmap lives in the first frontier queue (V words, idle outside a closure) */
    return k2_enqueue_flags(ctx, d_in_flags, d_out_flags, d_refcnt, ctx->d_queue[0]);
}

int mcx_dep_flags_noalloc_dev(mcx_ctx *ctx, const void *d_in_flags, void *d_out_flags, void *d_refcnt)
{
    if (!ctx || !d_in_flags || !d_out_flags)
        return MCX_E_BAD_ARG;
    if (!ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    if (ctx->V > MCX_NOALLOC_MAX_MODULES) {
        ctx->err = "mcx_dep_flags_noalloc_dev: graph of " + std::to_string(ctx->V) + " modules, the static bitmap holds " +
                   std::to_string(MCX_NOALLOC_MAX_MODULES);
        return MCX_E_TOO_LARGE;
    }
    if (ctx->V == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    uint32_t *bm = k2_static_bitmap(ctx);
    if (!bm)
        return MCX_E_HIP;
    /* the bitmap kernel writes every word the flag kernel reads, so what the last
     * call left in the array does not matter */
    return k2_enqueue_flags(ctx, d_in_flags, d_out_flags, d_refcnt, bm);
}

/* graphs up to this size take their host-buffer calls through pinned host memory */
#define K2_ZERO_COPY_MODULES 65536u

int mcx_dep_flags(mcx_ctx *ctx, const uint8_t *in_flags, uint8_t *out_flags, uint32_t *refcnt)
{
    if (!ctx || !in_flags || !out_flags)
        return MCX_E_BAD_ARG;
    if (!ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    uint32_t V = ctx->V;
    if (V == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    /* flags in, flags out, reference counts: three pieces of one staging area */
    const size_t piece = ((size_t)V + 255) & ~(size_t)255;
    if (V <= K2_ZERO_COPY_MODULES) {
        /* the module table of one host: in pinned host memory, read and written by the kernels over
         * the bus (no copies, no stream synchronisation: mcx_internal.h) */
        uint8_t *hb = nullptr, *db = nullptr;
        int rc0 = mcx_pinned_io(ctx, &hb, &db);
        if (rc0 != MCX_OK)
            return rc0;
        memcpy(hb, in_flags, V);
        if ((rc0 = mcx_dep_flags_dev(ctx, db, db + piece, refcnt ? db + 2 * piece : nullptr)) != MCX_OK ||
            (rc0 = mcx_signal_and_wait(ctx)) != MCX_OK)
            return rc0;
        memcpy(out_flags, hb + piece, V);
        if (refcnt)
            memcpy(refcnt, hb + 2 * piece, 4 * (size_t)V);
        return MCX_OK;
    }
    uint8_t *base = nullptr;
    int rc = mcx_io_reserve(ctx, 2 * piece + 4 * (size_t)V, &base);
    if (rc != MCX_OK)
        return rc;
    uint8_t *d_in = base, *d_out = base + piece;
    uint32_t *d_ref = refcnt ? (uint32_t *)(base + 2 * piece) : nullptr;
    if (hipMemcpyAsync(d_in, in_flags, V, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = MCX_E_HIP;
    if (rc == MCX_OK)
        rc = mcx_dep_flags_dev(ctx, d_in, d_out, d_ref);
    if (rc == MCX_OK && hipMemcpyAsync(out_flags, d_out, V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = MCX_E_HIP;
    if (rc == MCX_OK && refcnt &&
        hipMemcpyAsync(refcnt, d_ref, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = MCX_E_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MCX_OK)
        rc = MCX_E_HIP;
    return rc;
}

/* the queue sizes of a batch of levels, the edge count and the record of the one-workgroup
 * kernel, written straight into pinned host memory: one small launch instead of
 * device-to-host copies.  A batch with an empty level in it was the last one: the state
 * is then cleared for the next call. */
__global__ void k2b_publish(uint32_t *__restrict__ cnt, uint32_t first, uint32_t nwords, k2b_report_t rep, uint32_t seq)
{
    __shared__ uint32_t ended;
    const uint32_t i = threadIdx.x;
    if (i == 0)
        ended = 0u;
    __syncthreads();
    if (i < nwords) {
        const uint32_t x = cnt[first + i];
        rep.sizes[i] = x;
        /* even words are queue sizes, the odd word behind each the hub count of that level */
        if (!(i & 1u) && x == 0u && cnt[first + i + 1u] == 0u)
            ended = 1u;
    }
    if (i == 0) {
        *rep.edges = *(const unsigned long long *)(cnt + K2B_W_EDGES);
        rep.small[0] = cnt[K2B_W_SMALL_LEVELS];
        rep.small[1] = cnt[K2B_W_HANDOVER];
    }
    /* the sequence word goes last: the host spins on it instead of synchronising the stream */
    __threadfence_system();
    __syncthreads();
    if (i == 0)
        __hip_atomic_store(rep.seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (ended)
        for (uint32_t k = i; k < K2B_STATE_WORDS; k += blockDim.x)
            cnt[k] = 0u;
}

/* restart the window of per-level counters at level `base` (even, so that queue and hub list
 * keep their parity): its two counters become those of level 0, the rest is cleared */
__global__ void k2b_rewind(uint32_t *__restrict__ cnt, uint32_t base)
{
    __shared__ uint32_t keep[2];
    if (threadIdx.x < 2u)
        keep[threadIdx.x] = cnt[2u * base + threadIdx.x];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < K2B_W_EDGES; i += blockDim.x)
        cnt[i] = i < 2u ? keep[i] : 0u;
}

int mcx_dep_closure_dev(mcx_ctx *ctx, const void *d_in_flags, const void *d_seed_bitmap, void *d_out_bitmap,
                        int direction, uint64_t *edges_scanned, uint32_t *levels)
{
    if (!ctx || !d_in_flags || !d_seed_bitmap || !d_out_bitmap ||
        (direction != MCX_CLOSURE_HOLDERS && direction != MCX_CLOSURE_DEPS))
        return MCX_E_BAD_ARG;
    if (!ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    if (edges_scanned)
        *edges_scanned = 0;
    if (levels)
        *levels = 0;
    const uint32_t V = ctx->V;
    if (V == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    const uint32_t *rp = direction == MCX_CLOSURE_HOLDERS ? ctx->d_rrowptr : ctx->d_rowptr;
    const uint32_t *cl = direction == MCX_CLOSURE_HOLDERS ? ctx->d_rcol : ctx->d_col;
    const uint32_t W = (V + 31) / 32;
    uint2 *queue[2] = {(uint2 *)ctx->d_queue[0], (uint2 *)ctx->d_queue[1]};
    k2b_state_t st;
    st.cnt = ctx->d_bfs;
    st.edges = (unsigned long long *)(ctx->d_bfs + K2B_W_EDGES); /* 8-byte aligned */
    st.hubq[0] = (uint2 *)(ctx->d_bfs + K2B_STATE_WORDS);
    st.hubq[1] = st.hubq[0] + ctx->hub_cap;
    st.avail = (uint32_t *)(st.hubq[1] + ctx->hub_cap);
    /* the kernel that reports the end of a search clears the state; a call that failed on the
     * way, or the first one on this graph, clears it here */
    if (!ctx->bfs_clean)
        HIPCHK(ctx, hipMemsetAsync(ctx->d_bfs, 0, 4 * (size_t)K2B_STATE_WORDS, ctx->stream));
    ctx->bfs_clean = false;
    k2b_seed<<<(W + K2_THREADS - 1) / K2_THREADS, K2_THREADS, 0, ctx->stream>>>(
        (const uint32_t *)d_seed_bitmap, (const uint8_t *)d_in_flags, ((uintptr_t)d_in_flags & 15u) == 0u ? 1u : 0u, V,
        rp, (uint32_t *)d_out_bitmap, queue[0], st);
    uint32_t *sizes = ctx->h_pinned;                                  /* 2 * K2B_BATCH words */
    unsigned long long *h_edges = (unsigned long long *)(ctx->h_pinned + 2 * K2B_BATCH);
    uint32_t *h_small = ctx->h_pinned + 2 * K2B_BATCH + 2;            /* levels, handed over */
    uint32_t *d_sizes = ctx->d_pinned; /* the same pinned memory as the device addresses it */
    k2b_report_t rep;
    rep.sizes = d_sizes;
    rep.edges = (unsigned long long *)(d_sizes + 2 * K2B_BATCH);
    rep.small = d_sizes + 2 * K2B_BATCH + 2;
    rep.seq = d_sizes + MCX_PINNED_SEQ_WORD;
    rep.hist = nullptr;
    /* first the one-workgroup kernel, as far as the levels stay small (most closures of a few
     * modules end there: one launch, which also reports).  A search that needed per-level
     * launches last time in this direction will most likely do so again: its first batch is
     * then enqueued without a look in between. */
    const bool expect_handover = ctx->bfs_depth[direction] != 0u;
    const uint32_t seq0 = expect_handover ? 0u : mcx_next_seq(ctx);
    const uint32_t dedup = direction == MCX_CLOSURE_DEPS ? 1u : 0u;
    k2b_small<<<1, K2S_THREADS, 0, ctx->stream>>>(rp, cl, (uint32_t *)d_out_bitmap, queue[0], queue[1], st, 0u, rep, seq0,
                                                  0u, 0u, dedup);
    HIPCHK(ctx, hipGetLastError());
    if (!expect_handover && mcx_wait_seq(ctx, seq0))
        return MCX_E_HIP;
    uint32_t lv = 0;
    bool ended_early = false; /* the totals then come from the words of the launch that ended the search */
    unsigned long long ended_edges = 0;
    uint32_t ended_small = 0;
    if (expect_handover || h_small[1] != 0u) {
        /* a launch per level from here, with the one-workgroup kernel in front of every second
         * level (it returns at once while the frontier is large, and takes the rest of the
         * search when it has become short again).  Levels are enqueued a batch at a time; a
         * level whose frontier is empty returns at once, and the queue sizes of a batch tell
         * where the search ended.  Searches over one graph tend to be equally deep, so a
         * batch is as long as the last search in this direction was, plus two (even, so that
         * the counter window can be restarted at a batch boundary).  The window of
         * K2B_MAX_LEVELS per-level counters is reused as often as needed. */
        const uint32_t hint = ctx->bfs_depth[direction];
        uint32_t batch = hint ? (hint + K2B_TAIL < 4u ? 4u : (hint + K2B_TAIL > K2B_BATCH ? K2B_BATCH : hint + K2B_TAIL)) : 8u;
        batch = (batch + 1u) & ~1u;
        uint32_t base = 0; /* level of the counter window the next batch starts at */
        bool done = false;
        /* where a one-workgroup launch inside a batch reports the end of the search (the batch's
         * own report comes a few launches later and would read a state that launch has cleared) */
        uint32_t *h_end = ctx->h_pinned + K2B_PINNED_END;
        k2b_report_t end_rep;
        end_rep.sizes = d_sizes + K2B_PINNED_END;
        end_rep.edges = (unsigned long long *)(d_sizes + K2B_PINNED_END + 2u);
        end_rep.small = d_sizes + K2B_PINNED_END + 4u;
        end_rep.seq = rep.seq;
        end_rep.hist = d_sizes + MCX_PINNED_HIST_WORD;
        const uint32_t *h_hist = ctx->h_pinned + MCX_PINNED_HIST_WORD;
        /* the frontier sizes of the last search in this direction, per-level launch by per-level
         * launch: searches over one graph tend to have the same shape, and the one-workgroup
         * kernel is worth a launch only in front of a level that will be short */
        const std::vector<uint32_t> last = ctx->bfs_sizes[direction];
        std::vector<uint32_t> &mine = ctx->bfs_sizes[direction];
        mine.clear();
        uint32_t g0 = 0; /* per-level launches enqueued by the batches before this one */
        /* Towards the dependencies every list leads to the same few modules everything needs:
         * at the start of a level each wave still sees them open and claims them, and claims
         * on one cache line are served one after the other - the first step of such a level
         * takes longer the more waves take it at once (10^5 seeds: 504 us with 4096 waves,
         * 171 with 1024).  So there a quarter of the grid sweeps the frontier; all of it
         * still shares the hubs' lists. */
        const uint32_t sweep_waves = direction == MCX_CLOSURE_DEPS ? K2B_BLOCKS * (K2_THREADS / 64u) / 4u
                                                                   : K2B_BLOCKS * (K2_THREADS / 64u);
        while (!done) {
            if (base + batch + 1u > K2B_MAX_LEVELS) {
                k2b_rewind<<<1, 256, 0, ctx->stream>>>(st.cnt, base);
                base = 0;
            }
            const uint32_t seq_end = mcx_next_seq(ctx);
            for (uint32_t b = 0; b < batch; ++b) {
                const uint32_t level = base + b;
                /* not before the level at which the last search in this direction had become
                 * short again, less one: while the frontier is large that launch only looks and
                 * returns (a search that shrinks earlier than the last one runs its short levels
                 * as launches of their own until then) */
                const uint32_t g = g0 + b;
                const bool known = g < last.size();
                const bool try_small = known ? (last[g] != 0u && last[g] <= K2S_TRY_N)
                                             : (b % K2B_SMALL_EVERY == 0u && level + 1u >= hint);
                if (b != 0u && try_small)
                    k2b_small<<<1, K2S_THREADS, 0, ctx->stream>>>(rp, cl, (uint32_t *)d_out_bitmap, queue[level & 1u],
                                                                  queue[(level + 1u) & 1u], st, level, end_rep, seq_end,
                                                                  1u, base, dedup);
                /* without the look where the last search's level was short (a wrong guess costs time only) */
                if (dedup == 0u && known && last[g] <= K2B_LOOK_MAX && last[g] <= V / K2B_LOOK_PER)
                    k2b_expand<false><<<K2B_BLOCKS, K2_THREADS, 0, ctx->stream>>>(
                        rp, cl, queue[level & 1u], (uint32_t *)d_out_bitmap, queue[(level + 1u) & 1u], st, level,
                        sweep_waves, V, dedup);
                else
                    k2b_expand<true><<<K2B_BLOCKS, K2_THREADS, 0, ctx->stream>>>(
                        rp, cl, queue[level & 1u], (uint32_t *)d_out_bitmap, queue[(level + 1u) & 1u], st, level,
                        sweep_waves, V, dedup);
            }
            HIPCHK(ctx, hipGetLastError());
            k2b_publish<<<1, 64, 0, ctx->stream>>>(st.cnt, 2u * base, 2u * batch, rep, mcx_next_seq(ctx));
            HIPCHK(ctx, hipGetLastError());
            if (mcx_wait_seq2(ctx, seq_end, ctx->pinned_seq))
                return MCX_E_HIP;
            if (h_end[1] == seq_end) {
                /* a one-workgroup launch of this batch ended the search: every level before the
                 * one it was launched for had a frontier */
                lv += h_end[0] - base;
                for (uint32_t b = 0; b < h_end[0] - base && b < K2B_BATCH; ++b)
                    mine.push_back(h_hist[b]);
                ended_edges = *(const unsigned long long *)(h_end + 2);
                ended_small = h_end[4];
                ended_early = true;
                break;
            }
            for (uint32_t b = 0; b < batch; ++b) {
                if (sizes[2u * b] == 0u && sizes[2u * b + 1u] == 0u) {
                    done = true;
                    break;
                }
                mine.push_back(sizes[2u * b + 1u] ? 0xFFFFFFFFu : sizes[2u * b]); /* a level with hubs is never short */
                ++lv;
            }
            base += batch;
            g0 += batch;
        }
        /* 0: the one-workgroup kernel did it all */
        ctx->bfs_depth[direction] = lv > K2B_BATCH ? K2B_BATCH : lv;
    }
    ctx->bfs_clean = true;
    if (edges_scanned)
        *edges_scanned = ended_early ? ended_edges : *h_edges;
    if (levels)
        *levels = (ended_early ? ended_small : h_small[0]) + lv;
    return MCX_OK;
}

#ifdef K2B_STAMPS
int mcx_debug_k2b_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k2b_stamp), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : -1;
}
int mcx_debug_k2s_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k2s_stamp), sizeof(unsigned long long) * 32 * 8) == hipSuccess ? 0 : -1;
}
#endif

int mcx_dep_closure(mcx_ctx *ctx, const uint8_t *in_flags, const uint32_t *seed_bitmap, uint32_t *out_bitmap,
                    int direction, uint64_t *edges_scanned, uint32_t *levels)
{
    if (!ctx || !in_flags || !seed_bitmap || !out_bitmap)
        return MCX_E_BAD_ARG;
    if (!ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    uint32_t V = ctx->V;
    if (V == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    /* flags, seed bitmap, result bitmap: three pieces of the context's staging area */
    const size_t wb = 4 * (size_t)((V + 31) / 32);
    const size_t p_in = ((size_t)V + 255) & ~(size_t)255, p_bm = (wb + 255) & ~(size_t)255;
    uint8_t *base = nullptr;
    int rc = mcx_io_reserve(ctx, p_in + 2 * p_bm, &base);
    if (rc != MCX_OK)
        return rc;
    if (V <= K2_ZERO_COPY_MODULES) {
        /* one host's module table: flags and seeds are read in pinned host memory; the result stays in
         * device memory while the search claims its bits and goes to pinned memory with the launch
         * that tells the host */
        uint8_t *hb = nullptr, *db = nullptr;
        if ((rc = mcx_pinned_io(ctx, &hb, &db)) != MCX_OK)
            return rc;
        memcpy(hb, in_flags, V);
        memcpy(hb + p_in, seed_bitmap, wb);
        uint32_t *d_res = (uint32_t *)(base + p_in + p_bm);
        if ((rc = mcx_dep_closure_dev(ctx, db, db + p_in, d_res, direction, edges_scanned, levels)) != MCX_OK ||
            (rc = mcx_signal_and_wait(ctx, d_res, (uint32_t *)(db + p_in + p_bm), (uint32_t)(wb / 4))) != MCX_OK)
            return rc;
        memcpy(out_bitmap, hb + p_in + p_bm, wb);
        return MCX_OK;
    }
    uint8_t *d_in = base;
    uint32_t *d_seed = (uint32_t *)(base + p_in), *d_out = (uint32_t *)(base + p_in + p_bm);
    if (hipMemcpyAsync(d_in, in_flags, V, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_seed, seed_bitmap, wb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = MCX_E_HIP;
    if (rc == MCX_OK)
        rc = mcx_dep_closure_dev(ctx, d_in, d_seed, d_out, direction, edges_scanned, levels);
    if (rc == MCX_OK && hipMemcpyAsync(out_bitmap, d_out, wb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = MCX_E_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MCX_OK)
        rc = MCX_E_HIP;
    return rc;
}

} /* extern "C" */
