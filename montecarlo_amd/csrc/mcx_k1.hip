/*
 * mcx_k1.hip - K1: device modalias strings x alias patterns (fnmatch flags 0),
 * HIP kernels for gfx950 and their C ABI entry points (include/mcx.h).
 * There is no CPU path here: every entry point needs a HIP device.
 */
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "glob_compile.h"
#include "mcx_internal.h"
#include "mcx_scan.h"

/* ------------------------------------------------------------------------- */
/* K1: alias match                                                            */
/* ------------------------------------------------------------------------- */

#define K1_THREADS 256
#ifndef K1_MIN_WAVES
#define K1_MIN_WAVES 4 /* waves per SIMD the small-batch kernel's register allocation leaves room for */
#endif
#define K1_WAVES (K1_THREADS / 64)
#define K1_MAX_SPB 8                        /* strings per block, at most      */
#define K1_OCC_WORDS 512                     /* 256 x 64-bit position sets      */
#define K1_LDS_STR_BYTES 8192               /* staging buffer for a tile       */
#define K1_RING 2048u       /* survivors waiting for a program run, per block    */
#define K1_UNIT 256u        /* patterns screened per wave step (4 per lane)       */
#define K1_MAX_PATTERNS (1u << 26) /* survivor entries and group counts keep a pattern number in 26 bits */
#define K1_QUARTER 64u   /* a group of at most this many patterns takes a quarter of a
                            screening step (16 lanes): four such groups share one */
#define K1_LANEMAP 512u  /* small groups of one block that can share steps         */
#define K1_GROUP 16u        /* survivors one wave stages per pass                 */
#define K1_SUB 8u           /* of which it runs eight at a time, eight lanes each */
#define K1_STAGE_STRIDE (MCX_SLOT_WORDS + 1u) /* odd: lanes reading the same word
                                                 of their own slot hit 32 banks  */
#define K1_STAGE_WORDS (K1_GROUP * K1_STAGE_STRIDE + 4u)

/* what the match kernel needs from the loaded table; passed by value */
struct k1_table_t {
    const uint32_t *slot;   /* n_indexed x MCX_SLOT_WORDS                       */
    const uint32_t *ext;    /* overflow bodies                                 */
    const uint32_t *sig;    /* n_indexed gram signatures                       */
    const mcx_slot_t *slots;
    const uint8_t *tags;    /* per index slot: 0 = empty, else mcx_key_tag(hash)  */
    const mcx_sub_t *sub;   /* second index of the large groups                 */
    const uint8_t *sub_tags;
    uint32_t sub_mask;
    const uint32_t *sets;
    const uint32_t *lens;   /* nlens key lengths                               */
    uint32_t slot_mask;
    uint32_t nlens;
    uint32_t nlens_magic;   /* ceil(2^32 / nlens): x / nlens = mulhi(x, magic) for x < 2^16, nlens >= 2 */
    uint32_t norm_dash;     /* strings get kmod's '-' -> '_' rule before they are matched */
};

/* where the list variant of the kernel writes: string i owns
 * idx[off[i] .. off[i+1]), as many entries as the counting pass found */
struct k1_list_t {
    const uint32_t *off;
    uint32_t *idx;
    const uint32_t *dup_start; /* per slot: its copies in dup_idx, for slots with a multiplicity above 1 */
    const uint32_t *dup_idx;
    /* short lists (one pass): per string how many distinct pattern texts matched, and the first
     * K1T_LIST_CAP of them, in no particular order: (table position, 1) for a text that stands
     * once in the table, (slot number, copies) for one that stands there several times */
    uint32_t *nslots;
    uint2 *texts;
    /* per tile of the pass: entries its strings' lists hold together (each list cut at K, 0 = whole),
     * and how many of its strings are matched by more texts than a short list holds */
    unsigned long long *tile_sum;
    uint32_t *tile_over;
    uint32_t K;
};

typedef const __attribute__((address_space(3))) uint32_t *lds_words_t;

/* The words of an index entry that a probe looks at: hash, key length, first pattern, count; the key.
 * All 48 bytes are asked for at once and held to that: read field by field, the compiler asks
 * for the count first and for each further field behind the look at the one before - a chain of
 * round trips where the kernel's time is its chain of round trips. */
struct k1_slot_words_t {
    uint4 head, key_lo, key_hi;
};

__device__ __forceinline__ k1_slot_words_t k1_fetch_slot(const mcx_slot_t *p)
{
    static_assert(sizeof(mcx_slot_t) == 64 && MCX_KEY_MAX == 32, "entry layout as read here");
    const uint4 *q = (const uint4 *)p;
    k1_slot_words_t s{q[0], q[1], q[2]};
    asm volatile("" : "+v"(s.head.x), "+v"(s.head.y), "+v"(s.head.z), "+v"(s.head.w), "+v"(s.key_lo.x),
                 "+v"(s.key_lo.y), "+v"(s.key_lo.z), "+v"(s.key_lo.w), "+v"(s.key_hi.x), "+v"(s.key_hi.y),
                 "+v"(s.key_hi.z), "+v"(s.key_hi.w));
    return s;
}

/* an entry of the second index (gram, group, first pattern, count) in one 16-byte load, for the same reason */
__device__ __forceinline__ uint4 k1_fetch_sub(const mcx_sub_t *p)
{
    static_assert(sizeof(mcx_sub_t) == 16, "entry layout as read here");
    uint4 e = *(const uint4 *)p;
    asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w));
    return e;
}

/* the entry holds this key (an empty entry holds none: its length is 0) */
__device__ __forceinline__ bool k1_slot_is(const k1_slot_words_t &s, uint32_t hash, uint32_t len, const uint32_t (&kw)[8])
{
    return ((s.head.x ^ hash) | (s.head.y ^ len) | (s.key_lo.x ^ kw[0]) | (s.key_lo.y ^ kw[1]) | (s.key_lo.z ^ kw[2]) |
            (s.key_lo.w ^ kw[3]) | (s.key_hi.x ^ kw[4]) | (s.key_hi.y ^ kw[5]) | (s.key_hi.z ^ kw[6]) |
            (s.key_hi.w ^ kw[7])) == 0u;
}

/* a string staged in LDS: `at` is its byte offset in the word array; the array
 * is padded, so reading a few bytes past the string's end is safe */
struct LdsStr {
    lds_words_t w;
    uint32_t at;
    __device__ __forceinline__ uint32_t word(uint32_t q) const
    {
        uint32_t a = at + q, i = a >> 2;
        return __builtin_amdgcn_alignbit(w[i + 1], w[i], (a & 3u) * 8u);
    }
    __device__ __forceinline__ uint32_t byte(uint32_t q) const
    {
        uint32_t a = at + q;
        return (w[a >> 2] >> ((a & 3u) * 8u)) & 0xFFu;
    }
};

/* a string read in place (tiles too long for the staging buffer): bounded
 * byte reads, bytes past the end read as 0.  With `norm` a '-' outside a
 * bracket pair reads as '_': the nearest '[' or ']' in front of it says which
 * (this path is rare and favours being right over being fast). */
__device__ __attribute__((noinline)) uint32_t k1_dash_in_place(const uint8_t *p, uint32_t q)
{
    for (uint32_t r = q; r-- > 0u;) {
        if (p[r] == '[')
            return '-';
        if (p[r] == ']')
            break;
    }
    return '_';
}

struct GlobalStr {
    const uint8_t *p;
    uint32_t len;
    uint32_t norm;
    __device__ __forceinline__ uint32_t byte(uint32_t q) const
    {
        if (q >= len)
            return 0u;
        const uint32_t c = p[q];
        if (!norm || c != '-')
            return c;
        return k1_dash_in_place(p, q); /* out of line: rare, and kept off the common path's registers */
    }
    __device__ __forceinline__ uint32_t word(uint32_t q) const
    {
        return byte(q) | (byte(q + 1) << 8) | (byte(q + 2) << 16) | (byte(q + 3) << 24);
    }
};

struct GlobalProg {
    const uint32_t *p;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[i]; }
};
struct LdsProg {
    lds_words_t w;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return w[i]; }
};
/* the rare cases (a body in the overflow array, a string of more than 64 bytes)
 * read their program through a generic pointer: LDS or global */
struct FlatProg {
    const uint32_t *p;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[i]; }
};
/* occurrence table of one string: entry b = 64-bit set of positions holding b */
struct LdsOcc {
    lds_words_t w;
    __device__ __forceinline__ uint64_t operator[](uint32_t b) const
    {
        return (uint64_t)w[2u * b] | ((uint64_t)w[2u * b + 1u] << 32);
    }
};

struct __attribute__((aligned(4))) k1_sig4_t {
    uint32_t v[4]; /* gram signatures of four consecutive patterns */
};

/*
 * One block (4 waves) = `spb` consecutive strings x the whole table.
 *  1. the tile's bytes are staged in LDS; every string gets its 256-bit gram
 *     set and (up to 64 bytes) its occurrence table;
 *  2. one lane per (string, key length) hashes the string's prefix of that
 *     length and probes the prefix index; a hit names a group of patterns that
 *     all start with exactly those literal bytes, and reserves the group's
 *     units (up to 256 consecutive patterns each) in the block's unit space;
 *  3. the waves share the units, four patterns per lane, and keep the patterns
 *     whose gram signature is inside the string's gram set: survivors are
 *     compacted by ballot into one ring per block.  The signatures of a wave's
 *     next unit are requested before the current one is screened;
 *  4. survivors are run eight per wave and pass, all waves at once (a pass
 *     costs the same whatever its number of live lanes, so runs wait until
 *     every wave has a full group, or for the end).  Eight lanes share one
 *     survivor: they copy its slot into LDS, each lane computes the fit set of
 *     one segment from the string's occurrence table, and the sets are put
 *     together through lane shuffles.  Patterns in the overflow array and
 *     strings of more than 64 bytes are run by one lane, position by position.
 *     Matches go to per-string min / count cells in LDS, stored at the end.
 * The block owns its strings: no global atomics, no pre-initialised output.
 */
/*
 * kmod's rule for the queried alias: '-' becomes '_' outside [...]; a ']'
 * without its '[' or a '[' that is never closed makes the query unresolvable
 * (refused[slot] = 1).  The staged bytes of a tile are rewritten in LDS, one
 * wave per string, 64 bytes a step: the bracket pairs of a step come from two
 * ballots and a scalar walk over them (none at all in a modalias the kernel
 * wrote).  A tile read in place (in_place != nullptr) is only checked; its
 * dashes are translated when they are read.  Out of line, so that its registers
 * stay out of the matching path.  Called by all threads of the block; the
 * caller synchronises afterwards.
 */
__device__ __attribute__((noinline)) void k1_normalize_tile(uint32_t *str_w, const uint32_t *str_at,
                                                            const uint32_t *str_len, uint32_t *refused,
                                                            uint32_t nstr, const uint8_t *in_place)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (in_place) {
        for (uint32_t i = tid; i < nstr; i += K1_THREADS) {
            const uint8_t *p = in_place + str_at[i];
            const uint32_t len = str_len[i];
            bool inside = false, bad = false;
            for (uint32_t q = 0; q < len && !bad; ++q) {
                if (inside)
                    inside = p[q] != ']';
                else if (p[q] == ']')
                    bad = true;
                else
                    inside = p[q] == '[';
            }
            refused[i] = (bad || inside) ? 1u : 0u;
        }
        return;
    }
    for (uint32_t slot = wave; slot < nstr; slot += K1_WAVES) {
        const uint32_t len = str_len[slot], at = str_at[slot];
        bool inside = false, bad = false;
        for (uint32_t base = 0; base < len && !bad; base += 64u) {
            const uint32_t q = base + lane, a = at + q;
            const uint32_t c = q < len ? (str_w[a >> 2] >> ((a & 3u) * 8u)) & 0xFFu : 0u;
            const unsigned long long O = __ballot(c == '['), C = __ballot(c == ']');
            unsigned long long in_mask = 0;
            uint32_t pos = 0;
            while (pos < 64u) {
                const unsigned long long from = ~0ull << pos;
                if (inside) {
                    const unsigned long long rest = C & from;
                    if (!rest) {
                        in_mask |= from;
                        break;
                    }
                    const uint32_t cp = (uint32_t)__builtin_ctzll(rest);
                    in_mask |= from & (cp == 63u ? ~0ull : ((1ull << (cp + 1u)) - 1ull));
                    pos = cp + 1u;
                    inside = false;
                } else {
                    const unsigned long long ev = (O | C) & from;
                    if (!ev)
                        break;
                    const uint32_t ep = (uint32_t)__builtin_ctzll(ev);
                    if ((C >> ep) & 1ull) {
                        bad = true;
                        break;
                    }
                    inside = true;
                    pos = ep + 1u;
                }
            }
            if (!bad && q < len && c == '-' && !((in_mask >> lane) & 1ull))
                atomicXor(&str_w[a >> 2], (uint32_t)('-' ^ '_') << ((a & 3u) * 8u));
        }
        if (lane == 0)
            refused[slot] = (bad || inside) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(K1_THREADS, K1_MIN_WAVES)
k1_match_small(const uint8_t *__restrict__ S, const uint32_t *__restrict__ s_off, uint32_t N, uint32_t spb,
               uint32_t str_cap, uint32_t s_aligned, const k1_table_t T,
               uint32_t *__restrict__ first_idx, uint32_t *__restrict__ count
#ifdef MCX_K1_STAMPS
                 , unsigned long long *__restrict__ stamps
#endif
                 )
{
#ifdef MCX_K1_STAMPS
    /* diagnostic build only: shader-clock stamps at the phase boundaries */
#define K1_STAMP(i)                                                          \
    do {                                                                     \
        if (threadIdx.x == 0)                                                \
            stamps[blockIdx.x * 16u + (i)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
    unsigned long long acc_stage = 0, acc_mask = 0, acc_comb = 0, acc_n = 0, tmark = 0;
#define K1_TICK() (tmark = __builtin_amdgcn_s_memtime())
#define K1_TOCK(acc) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tmark; tmark = now_; } while (0)
#else
#define K1_STAMP(i) do { } while (0)
#define K1_TICK() do { } while (0)
#define K1_TOCK(acc) do { } while (0)
#endif
    K1_STAMP(0);
    /* dynamic LDS, sized by the launch for its strings-per-block:
     * [spb occurrence tables][str_cap/4 + 5 words of string bytes]
     * [hit_start][hit_cnt][hit_ub] of max_hits words each (hit_cnt: pattern
     * count in bits 0..25, string slot above) */
    extern __shared__ __attribute__((aligned(16))) uint32_t k1_lds[];
    /* survivors waiting for a program run.  A round adds at most K1_WAVES x K1_UNIT and leaves
     * less than K1_WAVES x GROUP behind */
    constexpr uint32_t RING = K1_RING;
    __shared__ uint32_t ring[RING];
    __shared__ uint32_t stage[K1_WAVES][K1_STAGE_WORDS];
    constexpr uint32_t GROUP = K1_GROUP; /* survivors one wave takes per pass */
    /* gram set of each string, one byte per bit number: the screen reads it with
     * byte loads at (set + bit), no shifts, and the gram pass fills it with plain
     * byte stores */
    __shared__ __attribute__((aligned(4))) uint8_t bloom[K1_MAX_SPB][256];
    __shared__ uint32_t res_first[K1_MAX_SPB], res_count[K1_MAX_SPB];
    __shared__ uint32_t str_at[K1_MAX_SPB], str_len[K1_MAX_SPB], refused[K1_MAX_SPB];
    __shared__ uint32_t key_lens[MCX_KEY_MAX + 1];
    __shared__ uint32_t nhits, nunits_sh, ring_cum[2];
    const uint32_t max_hits = spb * T.nlens;
    uint32_t *occ_w = k1_lds;
    uint32_t *str_w = occ_w + spb * K1_OCC_WORDS;
    uint32_t *hit_start = str_w + str_cap / 4u + 5u;
    uint32_t *hit_cnt = hit_start + max_hits;
    uint32_t *hit_ub = hit_cnt + max_hits;
    /* small groups (at most K1_QUARTER patterns) share screening steps, a quarter of a step
     * each: lane_map = their hit numbers */
    __shared__ uint16_t lane_map[K1_LANEMAP];
    __shared__ uint32_t nlanes_sh;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t t0 = blockIdx.x * spb;
    const uint32_t nstr = min(spb, N - t0);
    const uint32_t gbase = s_off[t0];
    const uint32_t gend = s_off[t0 + nstr];
    const uint32_t arena_end = s_off[N];
    const uint32_t total = gend - gbase;
    const bool use_lds = total <= str_cap;
    /* with a 4-byte aligned arena the tile is copied word by word from the
     * aligned address below its first byte; `shift` bytes of the previous
     * string then sit in front of it in the buffer */
    const uint32_t shift = (use_lds && s_aligned) ? (gbase & 3u) : 0u;

    /* The probe of the index does not wait for the tile to be staged.  Its
     * lanes (the last wave first) read the first bytes of their string straight from the
     * arena and walk the index while the other waves stage and build the gram sets; what
     * they found is recorded after the barrier.  (Not with kmod's dash rule, whose strings
     * are rewritten in LDS first, and not for an arena that cannot be read by words.) */
    const bool early = T.norm_dash == 0u && s_aligned != 0u;
    uint32_t e_start = 0, e_cnt = 0, e_slot = 0;
    {
        const uint32_t pr = (K1_THREADS - 1u) - tid;
        if (early && pr < nstr * T.nlens) {
            const uint32_t slot = T.nlens == 1u ? pr : __umulhi(pr, T.nlens_magic);
            const uint32_t KL = T.lens[pr - slot * T.nlens];
            const uint32_t so = s_off[t0 + slot], len = s_off[t0 + slot + 1u] - so;
            if (KL <= len) {
                const uint32_t sh = 8u * (so & 3u), a0 = so & ~3u;
                uint32_t raw[MCX_KEY_MAX / 4 + 1];
#pragma unroll
                for (uint32_t k = 0; k <= MCX_KEY_MAX / 4; ++k) {
                    const uint32_t b0 = a0 + 4u * k;
                    uint32_t v = 0;
                    if (b0 < so + KL) {
                        if (b0 + 4u <= arena_end) {
                            v = *(const uint32_t *)(S + b0);
                        } else {
                            for (uint32_t j = 0; j < 4u; ++j)
                                if (b0 + j < arena_end)
                                    v |= (uint32_t)S[b0 + j] << (8u * j);
                        }
                    }
                    raw[k] = v;
                }
                uint32_t kw[MCX_KEY_MAX / 4];
#pragma unroll
                for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k) {
                    uint32_t v = (uint32_t)((((uint64_t)raw[k + 1] << 32) | raw[k]) >> sh);
                    if (4u * k >= KL)
                        v = 0;
                    else if ((KL >> 2) == k && (KL & 3u))
                        v &= 0xFFFFFFFFu >> (8u * (4u - (KL & 3u)));
                    kw[k] = v;
                }
                const uint32_t a = mcx_key_hash(kw, KL);
                for (uint32_t at = a & T.slot_mask;; at = (at + 1u) & T.slot_mask) {
                    const k1_slot_words_t sl = k1_fetch_slot(T.slots + at);
                    const bool same = k1_slot_is(sl, a, KL, kw);
                    if (same) {
                        e_start = sl.head.z;
                        e_cnt = sl.head.w;
                        e_slot = slot;
                    }
                    if (sl.head.w == 0u || same)
                        break;
                }
            }
        }
    }

    if (tid < T.nlens)
        key_lens[tid] = T.lens[tid];
    for (uint32_t i = tid; i < nstr; i += K1_THREADS) {
        uint32_t so = s_off[t0 + i];
        str_at[i] = so - gbase + shift;
        str_len[i] = s_off[t0 + i + 1] - so;
        res_first[i] = MCX_NO_MATCH;
        res_count[i] = 0;
        refused[i] = 0;
    }
    for (uint32_t w = tid; w < nstr * 64u; w += K1_THREADS) {
        /* strings read in place get a full set: their patterns are not screened */
        uint32_t v = use_lds ? 0u : 0x01010101u;
        if ((w & 63u) == (MCX_BLOOM_FREE >> 2))
            v |= 1u << (8u * (MCX_BLOOM_FREE & 3u));
        ((uint32_t *)bloom[w >> 6])[w & 63u] = v;
    }
    if (tid == 0) {
        nhits = 0;
        nunits_sh = 0;
        nlanes_sh = 0;
        ring_cum[0] = 0;
        ring_cum[1] = 0;
    }
    for (uint32_t w = tid; w < nstr * K1_OCC_WORDS; w += K1_THREADS)
        occ_w[w] = 0u;
    if (use_lds) {
        const uint32_t a0 = gbase - shift; /* arena byte held by buffer byte 0 */
        const uint32_t nwords = (total + shift + 3u) / 4u + 3u;
        for (uint32_t w = tid; w < nwords; w += K1_THREADS) {
            const uint32_t b0 = a0 + 4u * w;
            uint32_t v = 0;
            if (s_aligned && b0 + 4u <= arena_end) {
                v = *(const uint32_t *)(S + b0);
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (b0 + k < arena_end)
                        v |= (uint32_t)S[b0 + k] << (8u * k);
            }
            str_w[w] = v;
        }
    }
    __syncthreads();
    if (T.norm_dash) {
        k1_normalize_tile(str_w, str_at, str_len, refused, nstr, use_lds ? nullptr : S + gbase);
        __syncthreads();
    }
    K1_STAMP(1);

    /* ---- gram sets: one wave per string, one lane per start position ------- */
    if (use_lds) {
        for (uint32_t slot = wave; slot < nstr; slot += K1_WAVES) {
            const uint32_t len = str_len[slot];
            LdsStr str{(lds_words_t)str_w, str_at[slot]};
            for (uint32_t q = lane; q + 4u <= len; q += 64u) {
                bloom[slot][mcx_gram_bit(str.word(q))] = 1u; /* every writer stores the same byte */
            }
            /* occurrence table, for strings the bit-parallel matcher takes */
            if (len <= MCX_OCC_MAX_LEN && lane < len)
                atomicOr(&occ_w[slot * K1_OCC_WORDS + 2u * str.byte(lane) + (lane >> 5)], 1u << (lane & 31u));
        }
    }

    K1_STAMP(3);
    /* one lane records a group of candidates (a hit) for string `slot`: a large group
     * takes units of its own, a small one lanes of the shared units */
    auto emit_hit = [&](uint32_t start, uint32_t cnt, uint32_t slot) __attribute__((always_inline)) {
        const uint32_t h = atomicAdd(&nhits, 1u);
        hit_start[h] = start;
        hit_cnt[h] = cnt | (slot << 26);
        uint32_t ub = 0xFFFFFFFFu;
        if (cnt > K1_QUARTER) {
            /* the group's units take the next free places of the block's unit
             * space, in whatever order the hits arrive */
            ub = atomicAdd(&nunits_sh, (cnt + K1_UNIT - 1u) / K1_UNIT);
        } else {
            /* a quarter of a shared unit per small group, whatever its size; lane_map is the
             * list of their hit numbers (one write here, on the block's critical path) */
            const uint32_t q = atomicAdd(&nlanes_sh, 1u);
            if (q < K1_LANEMAP)
                lane_map[q] = (uint16_t)h;
            else
                ub = atomicAdd(&nunits_sh, 1u); /* the shared steps are full: a unit of its own */
        }
        hit_ub[h] = ub;
    };
    /* ---- probe: last wave first, so that it overlaps the gram sets -------- */
#ifdef MCX_K1_STAMPS
    unsigned long long pt0 = __builtin_amdgcn_s_memtime(), pt1 = pt0, pt2 = pt0;
#endif
    const uint32_t nprobe = nstr * T.nlens;
    uint32_t pr_first = (K1_THREADS - 1u) - tid;
    if (early) {
        if (e_cnt)
            emit_hit(e_start, e_cnt, e_slot);
        pr_first += K1_THREADS; /* a block with more than K1_THREADS probes: the rest the usual way */
    }
    for (uint32_t pr = pr_first; pr < nprobe; pr += K1_THREADS) {
        /* pr / nlens; a table with a single key length has no 32-bit reciprocal */
        const uint32_t slot = T.nlens == 1u ? pr : __umulhi(pr, T.nlens_magic), L = key_lens[pr - slot * T.nlens];
        const uint32_t len = str_len[slot];
        if (L > len)
            continue;
        uint32_t kw[MCX_KEY_MAX / 4];
        if (use_lds) {
            LdsStr str{(lds_words_t)str_w, str_at[slot]};
#pragma unroll
            for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k)
                kw[k] = (4u * k < L) ? str.word(4u * k) : 0u;
        } else {
            GlobalStr str{S + gbase + str_at[slot], len, T.norm_dash};
#pragma unroll
            for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k)
                kw[k] = (4u * k < L) ? str.word(4u * k) : 0u;
        }
        /* zero the bytes past L in the last word */
#pragma unroll
        for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k)
            if ((L >> 2) == k && (L & 3u))
                kw[k] &= 0xFFFFFFFFu >> (8u * (4u - (L & 3u)));
        const uint32_t a = mcx_key_hash(kw, L);
#ifdef MCX_K1_STAMPS
        pt1 = __builtin_amdgcn_s_memtime();
#endif
        for (uint32_t at = a & T.slot_mask;; at = (at + 1u) & T.slot_mask) {
            /* the 64-byte entry at once: the tag bytes the large-batch kernel asks first would
             * only lengthen this kernel's chain of dependent loads */
            const k1_slot_words_t sl = k1_fetch_slot(T.slots + at);
            const bool same = k1_slot_is(sl, a, L, kw);
            if (same)
                emit_hit(sl.head.z, sl.head.w, slot);
            if (sl.head.w == 0u || same)
                break;
        }
    }
#ifdef MCX_K1_STAMPS
    pt2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == K1_THREADS - 1u) {
        stamps[blockIdx.x * 16u + 14u] = pt1 - pt0;
        stamps[blockIdx.x * 16u + 15u] = pt2 - pt1;
    }
#endif
    __syncthreads();
    K1_STAMP(2);

    /* ---- units -> survivors ----------------------------------------------- */
    const uint32_t nh = nhits;
    const uint32_t nbig = nunits_sh, nlanes = min(nlanes_sh, K1_LANEMAP);
    const uint32_t nunits = nbig + (nlanes + 3u) / 4u;
    /* the first 64 hits live in registers, one per lane */
    uint32_t my_ub = 0, my_units = 0, my_start = 0, my_cnt = 0, my_slot = 0;
    if (lane < nh) {
        my_ub = hit_ub[lane];
        my_cnt = hit_cnt[lane] & 0x03FFFFFFu;
        my_slot = hit_cnt[lane] >> 26;
        my_units = my_ub == 0xFFFFFFFFu ? 0u : (my_cnt + K1_UNIT - 1u) / K1_UNIT;
        my_start = hit_start[lane];
    }
    /* unit u -> per lane: first of its four patterns, how many of the four exist, string slot.
     * Units below nbig are 256 consecutive patterns of one large group; the ones
     * behind them hold small groups, a quarter of the unit per group. */
    auto unit_at = [&](uint32_t u, uint32_t &pk, uint32_t &pn, uint32_t &pslot) __attribute__((always_inline)) {
        if (u >= nbig) {
            pk = 0;
            pn = 0;
            pslot = 0;
            const uint32_t q = 4u * (u - nbig) + (lane >> 4), sub = 4u * (lane & 15u);
            if (q < nlanes) {
                const uint32_t h = lane_map[q], cs = hit_cnt[h], cnt = cs & 0x03FFFFFFu;
                pk = hit_start[h] + sub;
                pn = cnt > sub ? min(4u, cnt - sub) : 0u;
                pslot = cs >> 26;
            }
            return;
        }
        const unsigned long long m = __ballot(u - my_ub < my_units); /* lanes without a large group: 0 units */
        uint32_t ub, cnt, start, slot;
        if (m) {
            const uint32_t src = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_ctzll(m));
            ub = __builtin_amdgcn_readlane(my_ub, src);
            cnt = __builtin_amdgcn_readlane(my_cnt, src);
            start = __builtin_amdgcn_readlane(my_start, src);
            slot = __builtin_amdgcn_readlane(my_slot, src);
        } else {
            /* more than 64 hits in the block: the rest is searched in LDS (a
             * small group's base is 2^32 - 1: it never holds u) */
            uint32_t h = 64u;
            while (u - hit_ub[h] >= ((hit_cnt[h] & 0x03FFFFFFu) + K1_UNIT - 1u) / K1_UNIT)
                ++h;
            ub = hit_ub[h];
            cnt = hit_cnt[h] & 0x03FFFFFFu;
            start = hit_start[h];
            slot = hit_cnt[h] >> 26;
        }
        const uint32_t j = u - ub;
        const uint32_t n = min(K1_UNIT, cnt - K1_UNIT * j);
        pk = start + K1_UNIT * j + 4u * lane;
        pn = n > 4u * lane ? min(4u, n - 4u * lane) : 0u;
        pslot = slot;
    };
    auto load_sigs = [&](uint32_t pk, uint32_t pn, k1_sig4_t &out) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t jj = 0; jj < 4; ++jj)
            out.v[jj] = 0u;
        /* a partly used group of four reads up to three signatures past the
         * group: the array is padded for that */
        if (pn)
            out = *(const k1_sig4_t *)(T.sig + pk);
    };

    /* run n <= K1_GROUP survivors ring[from .. from+n): all their slots are
     * copied to LDS in one go (64 lanes x 2 x 16 bytes = 16 slots), then
     * lanes 8g .. 8g+7 share survivor 8t+g in sub-pass t */
    /* one lane, position by position: bodies in the overflow array, strings of
     * more than 64 bytes, tiles read in place */
    auto slow_match = [&](const uint32_t *prog, uint32_t at, uint32_t hdr, uint32_t slot, uint32_t len)
        __attribute__((always_inline)) {
        const FlatProg fp{prog};
        if (use_lds)
            return mcx_prog_match(fp, at, hdr, LdsStr{(lds_words_t)str_w, str_at[slot]}, len, T.sets);
        return mcx_prog_match(fp, at, hdr, GlobalStr{S + gbase + str_at[slot], len, T.norm_dash}, len, T.sets);
    };
    /* the pattern at table position orig (mult copies of its text) matches string `slot` */
    auto hit = [&](uint32_t slot, uint32_t orig, uint32_t mult) __attribute__((always_inline)) {
        atomicMin(&res_first[slot], orig);
        atomicAdd(&res_count[slot], mult);
    };
    uint32_t *my_stage = stage[wave];
    auto run_groups = [&](uint32_t from, uint32_t n) __attribute__((always_inline)) {
        K1_TICK();
#pragma unroll
        for (uint32_t t = 0; t < K1_GROUP * (MCX_SLOT_WORDS / 4u) / 64u; ++t) {
            const uint32_t piece = lane + 64u * t, i = piece / (MCX_SLOT_WORDS / 4u),
                           quarter = piece % (MCX_SLOT_WORDS / 4u);
            if (i < n) {
                const uint32_t kk = ring[(from + i) % RING] & 0x03FFFFFFu;
                const uint4 v = *(const uint4 *)(T.slot + (size_t)kk * MCX_SLOT_WORDS + 4u * quarter);
                uint32_t *dst = my_stage + i * K1_STAGE_STRIDE + 4u * quarter;
                dst[0] = v.x;
                dst[1] = v.y;
                dst[2] = v.z;
                dst[3] = v.w;
            }
        }
        K1_STAMP(7);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        K1_TOCK(acc_stage);
        const uint32_t k = lane & 7u;
        for (uint32_t g = lane >> 3; g < n; g += K1_SUB) {
            K1_TICK();
            /* the eight lanes of a group stay together: g < n is the same for them */
            const uint32_t slot = ring[(from + g) % RING] >> 26;
            const uint32_t *st = my_stage + g * K1_STAGE_STRIDE;
            const LdsProg lp{(lds_words_t)st};
            const uint32_t hdr = lp[0];
            const uint32_t len = str_len[slot];
            const bool is_ext = (MCX_HDR_FLAGS(hdr) & MCX_PF_EXT) != 0u;
            bool ok = false;
            if (is_ext || !use_lds || len > MCX_OCC_MAX_LEN) {
                /* rare: one lane, position by position */
                if (k == 0u) {
                    ok = slow_match(is_ext ? T.ext : st, is_ext ? lp[MCX_SLOT_DIR] : MCX_SLOT_BODY, hdr, slot, len);
                }
            } else {
                /* lane k of the group prepares the steps of segments k and
                 * k + 8; the state of the chain then moves one lane to the right
                 * per step (row_shr:1 stays inside the group except for lane 0,
                 * which feeds the start state), so after eight steps lane 7
                 * holds the state behind segment 7, and after eight more, fed
                 * with that, the state behind segment 15 */
                const uint32_t nseg = MCX_HDR_NSEG(hdr);
                const uint64_t all = len >= 64u ? ~(uint64_t)0 : (((uint64_t)1 << len) - 1u);
                mcx_step_t s0 = mcx_step_none(), s1 = mcx_step_none();
                const bool pre = mcx_precheck(hdr, len);
                if (pre) {
                    LdsOcc occ{(lds_words_t)occ_w + slot * K1_OCC_WORDS};
                    if (k < nseg) {
                        const mcx_seg_t sg = mcx_seg_at(lp, mcx_dir_word(lp, k), k, hdr);
                        s0 = mcx_step_make(hdr, len, k, mcx_seg_mask(lp, sg, occ, all, T.sets), sg.nchars);
                    }
                    if (k + 8u < nseg) {
                        const mcx_seg_t sg = mcx_seg_at(lp, mcx_dir_word(lp, k + 8u), k + 8u, hdr);
                        s1 = mcx_step_make(hdr, len, k + 8u, mcx_seg_mask(lp, sg, occ, all, T.sets), sg.nchars);
                    }
                }
                K1_TOCK(acc_mask);
                uint32_t state = MCX_STATE_START;
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)state, 0x111, 0xF, 0xF, false);
                    state = mcx_step_apply(s0, len, k == 0u ? MCX_STATE_START : left);
                }
                if (__any(nseg > 8u)) {
                    const uint32_t carry = __shfl(state, (int)((lane & ~7u) + 7u), 64);
                    state = carry;
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j) {
                        const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)state, 0x111, 0xF, 0xF, false);
                        state = mcx_step_apply(s1, len, k == 0u ? carry : left);
                    }
                }
                ok = pre && k == 7u && MCX_STATE_OK(state) != 0u;
            }
            if (ok)
                hit(slot, lp[1], lp[MCX_SLOT_MULT]);
            K1_TOCK(acc_comb);
#ifdef MCX_K1_STAMPS
            ++acc_n;
#endif
        }
        K1_STAMP(8);
        K1_STAMP(9);
        /* the staging area is reused by this wave's next pass */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    /* the block walks the units in rounds of K1_WAVES (one unit per wave) */
    const uint32_t nrounds = (nunits + K1_WAVES - 1u) / K1_WAVES;
    /* Ring bookkeeping without a second barrier per round: survivors of round r
     * are counted in ring_cum[r & 1], which only grows.  After the round's
     * barrier that cell is stable until round r+2 (which cannot start before
     * every wave has passed the barrier of round r+1), so every thread reads
     * the same value and keeps the ring's head and tail in registers. */
    uint32_t head = 0, tail = 0, seen0 = 0, seen1 = 0;
    uint32_t k0 = 0, n = 0, slot = 0; /* this lane's four patterns: first, how many exist, string slot */
    k1_sig4_t cur;
    if (wave < nunits) {
        unit_at(wave, k0, n, slot);
        load_sigs(k0, n, cur);
    }
    for (uint32_t r = 0; r < nrounds; ++r) {
        const uint32_t u = r * K1_WAVES + wave;
        /* request the next unit's signatures before this one's are looked at */
        uint32_t nk0 = 0, nn = 0, nslot = 0;
        k1_sig4_t nxt;
        const bool more = u + K1_WAVES < nunits;
        if (more) {
            unit_at(u + K1_WAVES, nk0, nn, nslot);
            load_sigs(nk0, nn, nxt);
        }
        if (u < nunits) {
            const uint8_t *bl = bloom[slot]; /* per lane: a unit of small groups spans strings */
            bool pass[4];
#pragma unroll
            for (uint32_t jj = 0; jj < 4; ++jj) {
                /* the four bits the signature names, read together */
                const uint32_t sg = cur.v[jj];
                const uint32_t all = bl[sg & 0xFFu] & bl[(sg >> 8) & 0xFFu] & bl[(sg >> 16) & 0xFFu] & bl[sg >> 24];
                pass[jj] = jj < n && all != 0u;
            }
            /* one reservation per wave; the order of the survivors in the ring
             * does not matter, so every one of the four pattern columns is
             * compacted by its own ballot */
            unsigned long long bm[4];
            uint32_t col[4], wtotal = 0;
#pragma unroll
            for (uint32_t jj = 0; jj < 4; ++jj) {
                bm[jj] = __ballot(pass[jj]);
                col[jj] = wtotal;
                wtotal += (uint32_t)__popcll(bm[jj]);
            }
            if (wtotal) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&ring_cum[r & 1u], wtotal);
                base = tail + (__builtin_amdgcn_readfirstlane(base) - ((r & 1u) ? seen1 : seen0));
#pragma unroll
                for (uint32_t jj = 0; jj < 4; ++jj)
                    if (pass[jj]) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                            (uint32_t)(bm[jj] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm[jj], 0u));
                        ring[(base + col[jj] + rank) % RING] = (slot << 26) | (k0 + jj);
                    }
            }
        }
        if (more) {
            k0 = nk0;
            n = nn;
            slot = nslot;
            cur = nxt;
        }
        __syncthreads();
        {
            /* whole rounds of K1_WAVES full groups go now; fewer than that many
             * survivors wait for more.  At most K1_WAVES x K1_UNIT new survivors
             * a round plus what was left over fit the ring. */
            const uint32_t now = ring_cum[r & 1u];
            tail += now - ((r & 1u) ? seen1 : seen0);
            if (r & 1u)
                seen1 = now;
            else
                seen0 = now;
            const uint32_t avail = tail - head;
            const uint32_t groups = (avail / GROUP) & ~(K1_WAVES - 1u);
            for (uint32_t g = wave; g < groups; g += K1_WAVES)
                run_groups(head + GROUP * g, GROUP);
            head += GROUP * groups;
            /* next round's survivors may only overwrite entries that have been
             * run: make sure of it when this round ran a large part of the ring */
            if (GROUP * groups + K1_WAVES * GROUP + K1_WAVES * K1_UNIT > RING)
                __syncthreads();
        }
    }
    K1_STAMP(4);
    {
        /* what is left goes to as few waves as the passes allow: a pass costs
         * a wave the same whatever its number of live lanes, so the survivors
         * are dealt out in whole passes (eight groups of lanes, or 64 lanes)
         * and a wave without one stays out of the way of the other blocks'
         * waves on its SIMD */
        constexpr uint32_t PASS = K1_SUB;
        const uint32_t avail = tail - head;
        const uint32_t passes = (avail + PASS - 1u) / PASS;
        const uint32_t share = PASS * ((passes + K1_WAVES - 1u) / K1_WAVES);
        const uint32_t from = wave * share;
        if (from < avail) {
            const uint32_t mine = min(share, avail - from);
            for (uint32_t off = 0; off < mine; off += GROUP)
                run_groups(head + from + off, min(GROUP, mine - off));
        }
    }
    __syncthreads();
    K1_STAMP(5);
    for (uint32_t i = tid; i < nstr; i += K1_THREADS) {
        first_idx[t0 + i] = refused[i] ? MCX_NO_MATCH : res_first[i];
        count[t0 + i] = refused[i] ? 0u : res_count[i];
    }
    K1_STAMP(6);
#ifdef MCX_K1_STAMPS
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 16u + 10u] = acc_stage;
        stamps[blockIdx.x * 16u + 11u] = acc_mask;
        stamps[blockIdx.x * 16u + 12u] = acc_comb;
        stamps[blockIdx.x * 16u + 13u] = acc_n;
    }
#endif
}


/* ------------------------------------------------------------------------- */
/* K1 for large batches: tiles of up to 64 strings per block.                  */
/* ------------------------------------------------------------------------- */
/*
 * What a string costs against a modules.alias-like table is a couple of index
 * hits, a few dozen candidates and two or three program runs of which one or
 * two match; a block that owns a handful of strings spends its life in the
 * latency of those few steps with most lanes idle.  Here a block owns up to 64
 * strings and every step is a flat pass over work items, a lane per item:
 *   probe    (string, key length)          -> candidate runs, or a (string,
 *                                             large group) pair for the second index
 *   second   (pair, 4-byte window)         -> candidate runs of that group
 *   screen   one candidate pattern         -> survivors (gram signature inside
 *                                             the string's gram set)
 *   run      one survivor                  -> min / count cells of its string
 * The items wait in bounded LDS queues; a producer step runs only when the
 * queue behind it has room for the most it can add, otherwise that queue is
 * drained first, so no table and no batch can overflow them.  Strings stay in
 * LDS as bytes (any length that fits the staging buffer) and programs are run
 * position by position: no occurrence tables, which is what lets 64 strings
 * share a block.
 */
#define K1T_THREADS 256u
/* Six blocks per CU: the kernel is a chain of short dependent steps, and every resident block
 * more hides more of them (10^6 strings x 30 000 patterns: 1.14 ms with four blocks, 0.96 with
 * five, 0.84 with six, the same with seven).  Six means at most 84 registers (80 used, three
 * spilled) and at most 26 880 bytes of LDS per block, which is what the queue sizes below are
 * cut to: LDS is handed out in pieces of 1280 bytes. */
#ifndef K1T_MIN_WAVES
#define K1T_MIN_WAVES 6 /* waves per SIMD the register allocation leaves room for */
#endif
#define K1T_WAVES (K1T_THREADS / 64u)
#ifndef K1T_RESIDENT
#define K1T_RESIDENT (K1T_MIN_WAVES * 256u) /* blocks of this kernel the 256 CUs hold at once */
#endif
#define K1T_MAX_SPB 64u
#ifndef K1T_STR_BYTES
#define K1T_STR_BYTES 5120u  /* staging buffer of a tile; a tile that needs more is read in place */
#endif
#ifndef K1T_RPT
#define K1T_RPT 3u           /* candidate runs per thread the screen's queue holds */
#endif
#define K1T_RCAP (K1T_RPT * K1T_THREADS) /* candidate runs waiting for the screen */
#ifndef K1T_BCAP
#define K1T_BCAP 272u        /* (string, large group) pairs waiting for the second index; a probe
                                step adds up to K1T_THREADS of them */
#endif
#ifndef K1T_SCAP
#define K1T_SCAP 896u        /* survivors waiting for a program run                  */
#endif
#ifndef K1T_LOWER
#define K1T_LOWER 1          /* 0: program runs without the strings' sets of lower-case positions (timing experiments) */
#endif
#define K1T_SUB_MAX_LEN 67u  /* strings up to this length go through the second index: one
                                lane per 4-byte window; longer ones have the whole group screened */
#ifndef K1T_PAIRS
#define K1T_PAIRS 4u         /* pairs a wave looks up at a time */
#endif
#define K1T_SUB_SPARE (2u * K1T_WAVES * K1T_PAIRS) /* runs a turn adds whatever happens: one for the
                                patterns without an anchor gram and one fallback per pair */

/* diagnostic builds only (-DK1T_SKIP=n): leave out the phases from n on, to time the rest */
#ifndef K1T_SKIP
#define K1T_SKIP 99
#endif

/* The tag bytes of four consecutive index entries from entry `at` on, lowest byte first.  The
 * tag arrays carry their first three bytes once more behind the last one, so a word read at any
 * entry stays inside and sees the walk wrap around. */
__device__ __forceinline__ uint32_t k1_tags4(const uint8_t *tags, uint32_t at)
{
    uint32_t v;
    __builtin_memcpy(&v, tags + at, 4);
    return v;
}

/* 0x80 in every byte of x that is zero; bytes above the lowest zero byte may be flagged wrongly */
__device__ __forceinline__ uint32_t k1_zero_bytes(uint32_t x)
{
    return (x - 0x01010101u) & ~x & 0x80808080u;
}

/* all bits below the lowest set bit of m (everything for m = 0) */
__device__ __forceinline__ uint32_t k1_below_lowest(uint32_t m)
{
    return (m & (0u - m)) - 1u;
}

#define K1T_DEDUP_BITS 7u
#define K1T_DEDUP_CELLS (1u << K1T_DEDUP_BITS)
static_assert(K1T_DEDUP_CELLS * K1T_WAVES <= K1T_RCAP, "the second index borrows the run sums' room");

/* MODE 0: first_idx and count.  MODE 1: the same plus, per string, the slots of the pattern
 * texts that matched (up to K1T_LIST_CAP, written to global memory as they are found): the match lists of
 * nearly every string follow from those without a second pass.  MODE 2: every match is written
 * into L.idx at the place L.off (the exclusive sums of a counting pass) gives its string. */
#define K1T_LIST_CAP 16u

/* what a block of the tile kernel keeps in LDS */
template <int MODE>
struct k1t_shared_t {
    __attribute__((aligned(16))) uint32_t str_w[K1T_STR_BYTES / 4u + 8u];
    uint32_t bloom[K1T_MAX_SPB][MCX_BLOOM_WORDS];
    uint32_t str_at[K1T_MAX_SPB], str_len[K1T_MAX_SPB], refused[K1T_MAX_SPB];
    uint32_t res_first[K1T_MAX_SPB], res_count[K1T_MAX_SPB];
    uint32_t key_lens[MCX_KEY_MAX + 1];
    uint32_t lcount[MODE == 1 ? K1T_MAX_SPB : 1u]; /* matching pattern texts per string */
    /* candidate runs: first pattern, count (bits 0..25) | string slot (26..31); run_cum = exclusive sums */
    uint32_t run_start[K1T_RCAP], run_cnt[K1T_RCAP], run_cum[K1T_RCAP + 1u];
    uint32_t big_key[K1T_BCAP], big_start[K1T_BCAP], big_count[K1T_BCAP], big_rest[K1T_BCAP];
    uint32_t surv[K1T_SCAP];
    uint32_t nruns_sh, nbig_sh, nsurv_sh, wsum[K1T_WAVES];
    uint32_t lowm[K1T_MAX_SPB][2]; /* per string: the positions below 64 that hold a lower-case letter */
    uint8_t mark[K1T_WAVES][64]; /* screen: where runs start inside a wave's window of 64 candidates */
};

/* The kernel's work on one tile.  INPLACE: the tile's strings do not fit the staging buffer and are
 * read where they lie, byte by byte - rare, and compiled apart from the other case: with both in
 * one body every step of the kernel carries the byte-wise path along, twice the code and twice the
 * scalar registers. */
template <int MODE, bool INPLACE>
__device__ __forceinline__ void
k1_tiles_tile(const uint8_t *__restrict__ S, const uint32_t *__restrict__ s_off, uint32_t N, uint32_t spb,
              uint32_t s_aligned, const k1_table_t &T, uint32_t *__restrict__ first_idx,
              uint32_t *__restrict__ count, const k1_list_t &L, k1t_shared_t<MODE> &sh, uint32_t gbase, uint32_t gend)
{
    auto &str_w = sh.str_w;
    auto &bloom = sh.bloom;
    auto &str_at = sh.str_at;
    auto &str_len = sh.str_len;
    auto &refused = sh.refused;
    auto &res_first = sh.res_first;
    auto &res_count = sh.res_count;
    auto &key_lens = sh.key_lens;
    constexpr bool LIST = MODE == 2;
    auto &lcount = sh.lcount;
    auto &run_start = sh.run_start;
    auto &run_cnt = sh.run_cnt;
    auto &run_cum = sh.run_cum;
    auto &big_key = sh.big_key;
    auto &big_start = sh.big_start;
    auto &big_count = sh.big_count;
    auto &big_rest = sh.big_rest;
    auto &surv = sh.surv;
    auto &nruns_sh = sh.nruns_sh;
    auto &nbig_sh = sh.nbig_sh;
    auto &nsurv_sh = sh.nsurv_sh;
    auto &wsum = sh.wsum;
    auto &mark = sh.mark;
    auto &lowm = sh.lowm;

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t t0 = blockIdx.x * spb;
    const uint32_t nstr = min(spb, N - t0);
    const uint32_t arena_end = s_off[N];
    const uint32_t total = gend - gbase;
    constexpr bool use_lds = !INPLACE;
    const uint32_t shift = (use_lds && s_aligned) ? (gbase & 3u) : 0u;

    if (tid < T.nlens)
        key_lens[tid] = T.lens[tid];
    for (uint32_t i = tid; i < nstr; i += K1T_THREADS) {
        const uint32_t so = s_off[t0 + i];
        str_at[i] = so - gbase + shift;
        str_len[i] = s_off[t0 + i + 1u] - so;
        res_first[i] = LIST ? L.off[t0 + i] : MCX_NO_MATCH;
        res_count[i] = 0u;
        refused[i] = 0u;
        if constexpr (MODE == 1)
            lcount[i] = 0u;
    }
    for (uint32_t w = tid; w < nstr * MCX_BLOOM_WORDS; w += K1T_THREADS) {
        /* strings read in place get a full set: their patterns are not screened */
        uint32_t v = use_lds ? 0u : 0xFFFFFFFFu;
        if ((w & (MCX_BLOOM_WORDS - 1u)) == (MCX_BLOOM_FREE >> 5))
            v |= 1u << (MCX_BLOOM_FREE & 31u);
        bloom[w / MCX_BLOOM_WORDS][w & (MCX_BLOOM_WORDS - 1u)] = v;
    }
    if (tid == 0) {
        nruns_sh = 0;
        nbig_sh = 0;
        nsurv_sh = 0;
    }
    if (use_lds) {
        const uint32_t a0 = gbase - shift;
        const uint32_t nwords = (total + shift + 3u) / 4u + 3u;
        for (uint32_t w = tid; w < nwords; w += K1T_THREADS) {
            const uint32_t b0 = a0 + 4u * w;
            uint32_t v = 0;
            if (s_aligned && b0 + 4u <= arena_end) {
                v = *(const uint32_t *)(S + b0);
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (b0 + k < arena_end)
                        v |= (uint32_t)S[b0 + k] << (8u * k);
            }
            str_w[w] = v;
        }
    }
    __syncthreads();
    if (T.norm_dash) {
        k1_normalize_tile(str_w, str_at, str_len, refused, nstr, use_lds ? nullptr : S + gbase);
        __syncthreads();
    }
    /* gram sets: a wave per string, a lane per start position */
    if (use_lds) {
        for (uint32_t slot = wave; slot < nstr; slot += K1T_WAVES) {
            const uint32_t len = str_len[slot];
            const LdsStr str{(lds_words_t)str_w, str_at[slot]};
            /* the first 64 positions also say where the string has lower-case letters (the start
             * positions of the program runs' tag segments: mcx_prog_match); the words of the last
             * three reach past the string's end, which the staging buffer allows */
            {
                const uint32_t w = lane < len ? str.word(lane) : 0u;
                if (lane + 4u <= len) {
                    const uint32_t bit = mcx_gram_bit(w);
                    atomicOr(&bloom[slot][bit >> 5], 1u << (bit & 31u));
                }
                if (K1T_LOWER) {
                    const unsigned long long lw = __ballot(lane < len && mcx_is_lower(w & 0xFFu));
                    lowm[slot][lane & 1u] = (lane & 1u) ? (uint32_t)(lw >> 32) : (uint32_t)lw; /* (all lanes: no branch) */
                }
            }
            for (uint32_t q = lane + 64u; q + 4u <= len; q += 64u) {
                const uint32_t bit = mcx_gram_bit(str.word(q));
                atomicOr(&bloom[slot][bit >> 5], 1u << (bit & 31u));
            }
        }
    }

    auto word_of = [&](uint32_t slot, uint32_t q) __attribute__((always_inline)) -> uint32_t {
        if (use_lds)
            return LdsStr{(lds_words_t)str_w, str_at[slot]}.word(q);
        return GlobalStr{S + gbase + str_at[slot], str_len[slot], T.norm_dash}.word(q);
    };
    auto push_run = [&](uint32_t start, uint32_t cnt, uint32_t slot) __attribute__((always_inline)) {
        const uint32_t r = atomicAdd(&nruns_sh, 1u);
        run_start[r] = start;
        run_cnt[r] = cnt | (slot << 26);
    };
    auto hit = [&](uint32_t slot, uint32_t kk, uint32_t orig, uint32_t mult) __attribute__((always_inline)) {
        if constexpr (LIST) {
            if (!refused[slot]) {
                uint32_t *dst = L.idx + res_first[slot] + atomicAdd(&res_count[slot], mult);
                if (mult == 1u) {
                    dst[0] = orig;
                } else {
                    const uint32_t *src = L.dup_idx + L.dup_start[kk];
                    for (uint32_t m = 0; m < mult; ++m)
                        dst[m] = src[m];
                }
            }
        } else {
            atomicMin(&res_first[slot], orig);
            atomicAdd(&res_count[slot], mult);
            if constexpr (MODE == 1) {
                /* the short list itself goes straight to global memory: an entry or two per string */
                const uint32_t at = atomicAdd(&lcount[slot], 1u);
                if (at < K1T_LIST_CAP)
                    L.texts[(size_t)(t0 + slot) * K1T_LIST_CAP + at] = make_uint2(mult == 1u ? orig : kk, mult);
            }
        }
    };

    /* A queue counter as every thread of the block sees it: all appends made so far are
     * in, and nobody appends again before everybody has read it.  The queues' room is
     * checked with such reads only, so that the block takes every branch as one. */
    auto settled = [&](const uint32_t &counter) __attribute__((always_inline)) -> uint32_t {
        __syncthreads();
        const uint32_t v = counter;
        __syncthreads();
        return v;
    };

    /* ---- run: a lane per survivor ------------------------------------------ */
    auto drain_surv = [&]() __attribute__((always_inline)) {
        /* called by all threads behind a barrier: nsurv_sh is stable */
        const uint32_t n = K1T_SKIP <= 4 ? 0u : nsurv_sh;
        for (uint32_t i = tid; i < n; i += K1T_THREADS) {
            const uint32_t e = surv[i], slot = e >> 26, kk = e & 0x03FFFFFFu;
            const uint32_t len = str_len[slot];
            const uint32_t *sp = T.slot + (size_t)kk * MCX_SLOT_WORDS;
            /* header, table position, multiplicity and the first directory word in one request:
             * a match then has what it reports without another round trip */
            uint4 head = *(const uint4 *)sp;
            asm volatile("" : "+v"(head.x), "+v"(head.y), "+v"(head.z), "+v"(head.w)); /* (held to one request) */
            static_assert(MCX_SLOT_MULT == 2u && MCX_SLOT_DIR == 3u, "slot words 0..3 as read here");
            const uint32_t hdr = head.x;
            const bool is_ext = (MCX_HDR_FLAGS(hdr) & MCX_PF_EXT) != 0u;
            const FlatProg fp{is_ext ? T.ext : sp};
            const uint32_t at = is_ext ? head.w : MCX_SLOT_BODY;
            bool ok;
            if (use_lds)
                ok = mcx_prog_match(fp, at, hdr, LdsStr{(lds_words_t)str_w, str_at[slot]}, len, T.sets,
                                    (uint64_t)lowm[slot][0] | ((uint64_t)lowm[slot][1] << 32),
                                    K1T_LOWER != 0 && len <= 64u);
            else
                ok = mcx_prog_match(fp, at, hdr, GlobalStr{S + gbase + str_at[slot], len, T.norm_dash}, len, T.sets);
            if (ok)
                hit(slot, kk, head.y, head.z);
        }
        __syncthreads();
        if (tid == 0)
            nsurv_sh = 0;
        __syncthreads();
    };

    /* ---- screen: a lane per candidate of the queued runs -------------------- */
    auto drain_runs = [&]() __attribute__((always_inline)) {
        /* called by all threads behind a barrier */
        const uint32_t nr = K1T_SKIP <= 3 ? 0u : nruns_sh;
        /* exclusive sums of the run sizes, K1T_RPT runs per thread */
        {
            uint32_t c[K1T_RPT], mine = 0;
#pragma unroll
            for (uint32_t k = 0; k < K1T_RPT; ++k) {
                const uint32_t r = K1T_RPT * tid + k;
                c[k] = r < nr ? (run_cnt[r] & 0x03FFFFFFu) : 0u;
                mine += c[k];
            }
            uint32_t incl = mine;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d)
                    incl += o;
            }
            if (lane == 63u)
                wsum[wave] = incl;
            __syncthreads();
            uint32_t at = incl - mine;
#pragma unroll
            for (uint32_t k = 0; k < K1T_WAVES; ++k)
                if (k < wave)
                    at += wsum[k];
#pragma unroll
            for (uint32_t k = 0; k < K1T_RPT; ++k) {
                run_cum[K1T_RPT * tid + k] = at;
                at += c[k];
            }
            if (tid == K1T_THREADS - 1u)
                run_cum[K1T_RCAP] = at; /* K1T_RCAP = K1T_RPT x K1T_THREADS: the sum over all entries */
            __syncthreads();
        }
        const uint32_t ncand = run_cum[nr];
        uint32_t wrun = 0; /* the run of the wave's first candidate of the last step (the same in all its lanes) */
        uint32_t bound = nsurv_sh; /* no more survivors than this are queued */
        for (uint32_t c0 = 0; c0 < ncand; c0 += K1T_THREADS) {
            if (bound + K1T_THREADS > K1T_SCAP) {
                bound = settled(nsurv_sh);
                if (bound + K1T_THREADS > K1T_SCAP) {
                    drain_surv();
                    bound = 0;
                }
            }
            const uint32_t c = c0 + tid;
            bool pass = false;
            uint32_t e = 0;
            /* The run that holds candidate c: largest r with run_cum[r] <= c.  The 64 candidates
             * of a wave are consecutive, and so are their runs: the wave keeps the run of its
             * first candidate from step to step (wrun: the next 64 sums are looked at by a lane
             * each, and the ballot of those that lie at or below the first candidate says how
             * far to move on), and a lane's run is that one plus the number of runs that start
             * inside the wave's window at or before the lane's place - the starts are marked in
             * 64 bytes of LDS, and a ballot of the marks is counted up to the lane.  A bisection
             * per candidate was ten dependent LDS reads and sixty instructions a pass. */
            const uint32_t cw = c0 + 64u * wave; /* the wave's first candidate */
            uint32_t lo = 0;
            bool by_bisection = false;
            if (cw < ncand) {
                uint32_t v, adv;
                do {
                    v = run_cum[min(wrun + 1u + lane, nr)]; /* start of run wrun + 1 + lane (ncand beyond the last) */
                    adv = (uint32_t)__popcll(__ballot(v <= cw));
                    wrun += adv;
                } while (adv == 64u);
                wrun = __builtin_amdgcn_readfirstlane(wrun);
                /* v: the starts of runs wrun - adv + 1 + lane; those behind cw and inside the window mark their place */
                const uint32_t v_last = __builtin_amdgcn_readlane(v, 63);
                by_bisection = v_last < cw + 64u && v_last < ncand; /* more than 64 - adv runs start in the window */
                mark[wave][lane] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (v > cw && v < cw + 64u && v < ncand)
                    mark[wave][v - cw] = 1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long starts = __ballot(mark[wave][lane] != 0u);
                lo = wrun + __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u)) +
                     (uint32_t)((starts >> lane) & 1ull);
            }
            if (c < ncand) {
                if (by_bisection) { /* a window of runs of one: the long way round */
                    uint32_t hi = nr;
                    lo = 0;
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (run_cum[mid] <= c)
                            lo = mid;
                        else
                            hi = mid;
                    }
                }
                const uint32_t k = run_start[lo] + (c - run_cum[lo]), slot = run_cnt[lo] >> 26;
                const uint32_t sg = T.sig[k];
                const uint32_t *bl = bloom[slot];
                uint32_t all = 1u;
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t bit = (sg >> (8u * j)) & 0xFFu;
                    all &= bl[bit >> 5] >> (bit & 31u);
                }
                pass = (all & 1u) != 0u;
                e = (slot << 26) | k;
            }
            const unsigned long long bm = __ballot(pass);
            if (bm) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&nsurv_sh, (uint32_t)__popcll(bm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (pass)
                    surv[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = e;
            }
            bound += K1T_THREADS;
        }
        __syncthreads();
        if (tid == 0)
            nruns_sh = 0;
        __syncthreads();
    };

    /* ---- second index: a wave per (string, large group), a lane per window -- */
    auto drain_big = [&]() __attribute__((always_inline)) {
        /* called by all threads behind a barrier.  A wave takes K1T_PAIRS pairs per turn: the
         * words, hashes and tag bytes of all of them are under way before the first answer is
         * looked at.  The runs a pair finds are queued all or none (one reservation per pair);
         * when the queue has no room for them the group's indexed part goes in as one run, which
         * costs time, not correctness.  K1T_SUB_SPARE entries are kept for those single runs. */
        const uint32_t nb = K1T_SKIP <= 2 ? 0u : nbig_sh;
        for (uint32_t b0 = 0; b0 < nb; b0 += K1T_WAVES * K1T_PAIRS) {
            if (settled(nruns_sh) + K1T_SUB_SPARE + K1T_THREADS > K1T_RCAP)
                drain_runs(); /* ends with a barrier */
            uint32_t w[K1T_PAIRS], h[K1T_PAIRS], tag[K1T_PAIRS], key[K1T_PAIRS];
            bool act[K1T_PAIRS];
#pragma unroll
            for (uint32_t i = 0; i < K1T_PAIRS; ++i) {
                const uint32_t b = b0 + K1T_PAIRS * wave + i;
                act[i] = false;
                w[i] = h[i] = tag[i] = 0u;
                key[i] = 0xFFFFFFFFu;
                if (b < nb) {
                    key[i] = big_key[b];
                    const uint32_t slot = key[i] & 63u;
                    if (lane + 4u <= str_len[slot]) {
                        act[i] = true;
                        w[i] = word_of(slot, lane);
                        h[i] = mcx_sub_hash(w[i], key[i] >> 6);
                        tag[i] = k1_tags4(T.sub_tags, h[i] & T.sub_mask);
                    }
                }
            }
            /* nearly every walk ends inside the first word of tags, with at most one entry to look
             * at: those entries - of all the turn's pairs - are asked for together; the other walks
             * loop behind */
            uint4 first_en[K1T_PAIRS];
            uint32_t cands[K1T_PAIRS];
#pragma unroll
            for (uint32_t i = 0; i < K1T_PAIRS; ++i) {
                first_en[i] = make_uint4(0u, 0xFFFFFFFFu, 0u, 0u); /* no group has this number */
                cands[i] = 0u;
                if (act[i]) {
                    const uint32_t tg = mcx_key_tag(h[i]) * 0x01010101u;
                    cands[i] = k1_zero_bytes(tag[i] ^ tg) & k1_below_lowest(k1_zero_bytes(tag[i]));
                    if (cands[i] != 0u)
                        first_en[i] = *(const uint4 *)(T.sub + (((h[i] & T.sub_mask) + ((uint32_t)__builtin_ctz(cands[i]) >> 3)) & T.sub_mask));
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < K1T_PAIRS; ++i) {
                const uint32_t b = b0 + K1T_PAIRS * wave + i;
                if (b >= nb)
                    break; /* the same for the whole wave */
                bool found = false;
                uint32_t fstart = 0, fcnt = 0;
                bool walk_on = false;
                uint32_t t4 = tag[i], cand = cands[i], p = h[i] & T.sub_mask;
                const uint32_t tg = mcx_key_tag(h[i]) * 0x01010101u, gid = key[i] >> 6;
                if (act[i]) {
                    const uint32_t empty = k1_zero_bytes(t4);
                    if (cand != 0u) {
                        const uint4 en = first_en[i];
                        cand &= cand - 1u;
                        if (((en.x ^ w[i]) | (en.y ^ gid)) == 0u) {
                            found = true;
                            fstart = en.z;
                            fcnt = en.w;
                        }
                    }
                    walk_on = !found && (cand != 0u || empty == 0u);
                }
                if (__ballot(walk_on) != 0ull && walk_on) {
                    for (;;) {
                        while (cand != 0u && !found) {
                            const uint4 en = k1_fetch_sub(T.sub + ((p + ((uint32_t)__builtin_ctz(cand) >> 3)) & T.sub_mask));
                            cand &= cand - 1u;
                            if (((en.x ^ w[i]) | (en.y ^ gid)) == 0u) {
                                found = true;
                                fstart = en.z;
                                fcnt = en.w;
                            }
                        }
                        if (found || k1_zero_bytes(t4) != 0u)
                            break;
                        p = (p + 4u) & T.sub_mask;
                        t4 = k1_tags4(T.sub_tags, p);
                        cand = k1_zero_bytes(t4 ^ tg) & k1_below_lowest(k1_zero_bytes(t4));
                    }
                }
                /* a gram that occurs twice in the string names its run once.  The lanes that
                 * found something write their number into a cell of a small table chosen by
                 * the gram (the sums of the run sizes are not in use during a turn: their room
                 * serves), and whoever reads his own number back keeps the run.  Two different
                 * grams in one cell: the long way round, lane by lane. */
                if (const unsigned long long fm0 = __ballot(found); (fm0 & (fm0 - 1ull)) != 0ull) {
                    volatile uint32_t *cell = run_cum + K1T_DEDUP_CELLS * wave + ((w[i] * 0x9E3779B1u) >> (32u - K1T_DEDUP_BITS));
                    if (found)
                        *cell = lane;
                    const uint32_t winner = found ? *cell : lane;
                    const uint32_t theirs = __shfl(w[i], (int)winner, 64);
                    if (__ballot(found && theirs != w[i]) != 0ull) {
                        unsigned long long todo = fm0;
                        while (todo) {
                            const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_ctzll(todo));
                            const uint32_t wl = __builtin_amdgcn_readlane(w[i], l);
                            const unsigned long long same = __ballot(found && w[i] == wl);
                            if (found && w[i] == wl && lane != l)
                                found = false;
                            todo &= ~same;
                        }
                    } else if (winner != lane) {
                        found = false;
                    }
                }
                const unsigned long long fm = __ballot(found);
                const uint32_t nf = (uint32_t)__popcll(fm), slot = key[i] & 63u;
                const uint32_t rest = big_rest[b], gstart = big_start[b], gcount = big_count[b];
                uint32_t base = 0xFFFFFFFFu;
                if (lane == 0u) {
                    if (rest != 0u)
                        push_run(gstart, rest, slot);
                    if (nf != 0u) {
                        uint32_t old = nruns_sh;
                        for (;;) {
                            if (old + nf > K1T_RCAP - K1T_SUB_SPARE)
                                break;
                            const uint32_t seen = atomicCAS(&nruns_sh, old, old + nf);
                            if (seen == old) {
                                base = old;
                                break;
                            }
                            old = seen;
                        }
                        if (base == 0xFFFFFFFFu)
                            push_run(gstart + rest, gcount - rest, slot);
                    }
                }
                base = __builtin_amdgcn_readfirstlane(base);
                if (found && base != 0xFFFFFFFFu) {
                    const uint32_t r = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                    run_start[r] = fstart;
                    run_cnt[r] = fcnt | (slot << 26);
                }
            }
        }
        __syncthreads();
        if (tid == 0)
            nbig_sh = 0;
        __syncthreads();
    };

    /* ---- probe: a lane per (string, key length) ------------------------------ */
    __syncthreads();
    const uint32_t nprobe = K1T_SKIP <= 1 ? 0u : nstr * T.nlens;
    /* ceil(2^32 / nstr): x / nstr = mulhi(x, magic) for x < 2^16, nstr >= 2 */
    const uint32_t nstr_magic = nstr > 1u ? (uint32_t)((0x100000000ull + nstr - 1u) / nstr) : 0u;
    uint32_t rbound = 0, bbound = 0; /* no more runs / pairs than these are queued */
    for (uint32_t p0 = 0; p0 < nprobe; p0 += K1T_THREADS) {
        /* a step adds at most one run or one pair per lane */
        if (bbound + K1T_THREADS > K1T_BCAP) {
            bbound = settled(nbig_sh);
            if (bbound + K1T_THREADS > K1T_BCAP) {
                drain_big(); /* ends with a barrier */
                bbound = 0;
                rbound = K1T_RCAP; /* unknown now: looked up below */
            }
        }
        if (rbound + K1T_THREADS > K1T_RCAP) {
            rbound = settled(nruns_sh);
            if (rbound + K1T_THREADS > K1T_RCAP) {
                drain_runs();
                rbound = 0;
            }
        }
        rbound += K1T_THREADS;
        bbound += K1T_THREADS;
        const uint32_t pr = p0 + tid;
        /* words of the longest key among this wave's items: that of its last one */
        const uint32_t pr_last = min(p0 + 64u * wave + 63u, nprobe - 1u);
        const uint32_t nkw =
            __builtin_amdgcn_readfirstlane((key_lens[nstr == 1u ? pr_last : __umulhi(pr_last, nstr_magic)] + 3u) >> 2);
        if (pr < nprobe) {
            /* key length by key length over all strings of the tile, not string by string: the
             * lanes of a step then hash keys of one length, and what the later phases are handed
             * comes sorted by key length - the patterns of one group, of similar shape, side by
             * side in the survivor queue instead of every string's mix of short and long programs */
            const uint32_t li = nstr == 1u ? pr : __umulhi(pr, nstr_magic), slot = pr - li * nstr;
            const uint32_t KL = key_lens[li];
            const uint32_t len = str_len[slot];
            if (KL <= len) {
                /* the key's words and their hash: only as many words as the longest key of this
                 * wave's lanes reaches into (the lanes of a wave hash keys of one length, two at a
                 * boundary; the lengths ascend with the item number) - the same for the whole wave,
                 * so the words behind are skipped by a branch, not computed and thrown away */
                uint32_t kw[MCX_KEY_MAX / 4];
                uint32_t a = mcx_key_hash_start(KL);
#pragma unroll
                for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k) {
                    kw[k] = 0u;
                    if (k < nkw) {
                        uint32_t wk = (4u * k < KL) ? word_of(slot, 4u * k) : 0u;
                        if ((KL >> 2) == k && (KL & 3u))
                            wk &= 0xFFFFFFFFu >> (8u * (4u - (KL & 3u)));
                        kw[k] = wk;
                        a = 4u * k < KL ? mcx_key_hash_step(a, wk) : a;
                    }
                }
                a = mcx_key_hash_end(a);
                const uint32_t my_tag = mcx_key_tag(a) * 0x01010101u;
                /* four tag bytes a step: nearly every walk ends inside the first word */
                bool done = false;
                for (uint32_t at = a & T.slot_mask; !done; at = (at + 4u) & T.slot_mask) {
                    const uint32_t t4 = k1_tags4(T.tags, at);
                    const uint32_t empty = k1_zero_bytes(t4);
                    uint32_t cand = k1_zero_bytes(t4 ^ my_tag) & k1_below_lowest(empty);
                    done = empty != 0u;
                    while (cand != 0u) {
                        const mcx_slot_t *sp = T.slots + ((at + ((uint32_t)__builtin_ctz(cand) >> 3)) & T.slot_mask);
                        const k1_slot_words_t sl = k1_fetch_slot(sp);
                        uint2 pad = *(const uint2 *)sp->pad; /* second index of the group, and what it leaves out */
                        asm volatile("" : "+v"(pad.x), "+v"(pad.y));
                        cand &= cand - 1u;
                        if (k1_slot_is(sl, a, KL, kw)) {
                            if (pad.x != 0u && use_lds && len <= K1T_SUB_MAX_LEN) {
                                const uint32_t bi = atomicAdd(&nbig_sh, 1u);
                                big_key[bi] = slot | ((pad.x - 1u) << 6);
                                big_start[bi] = sl.head.z;
                                big_count[bi] = sl.head.w;
                                big_rest[bi] = pad.y;
                            } else {
                                push_run(sl.head.z, sl.head.w, slot);
                            }
                            done = true;
                            break;
                        }
                    }
                }
            }
        }
    }
    if (settled(nbig_sh) != 0u)
        drain_big();
    if (settled(nruns_sh) != 0u)
        drain_runs();
    if (settled(nsurv_sh) != 0u)
        drain_surv();
    if constexpr (!LIST) {
        for (uint32_t i = tid; i < nstr; i += K1T_THREADS) {
            first_idx[t0 + i] = refused[i] ? MCX_NO_MATCH : res_first[i];
            count[t0 + i] = refused[i] ? 0u : res_count[i];
            if constexpr (MODE == 1)
                L.nslots[t0 + i] = refused[i] ? 0u : lcount[i];
        }
    }
    if constexpr (MODE == 1) {
        if (wave == 0u) {
            unsigned long long sum = 0;
            uint32_t over = 0;
            for (uint32_t i = lane; i < nstr; i += 64u) {
                if (!refused[i]) {
                    const uint32_t c = res_count[i];
                    sum += L.K != 0u && c > L.K ? L.K : c;
                    over += lcount[i] > K1T_LIST_CAP ? 1u : 0u;
                }
            }
            for (int d = 32; d > 0; d >>= 1) {
                sum += __shfl_xor(sum, d);
                over += __shfl_xor(over, d);
            }
            if (lane == 0u) {
                L.tile_sum[blockIdx.x] = sum;
                L.tile_over[blockIdx.x] = over;
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(K1T_THREADS, K1T_MIN_WAVES)
k1_match_tiles(const uint8_t *__restrict__ S, const uint32_t *__restrict__ s_off, uint32_t N, uint32_t spb,
               uint32_t s_aligned, const k1_table_t T, uint32_t *__restrict__ first_idx,
               uint32_t *__restrict__ count, const k1_list_t L)
{
    static_assert(K1T_MIN_WAVES != 6 || sizeof(k1t_shared_t<MODE>) <= 26880u, "six blocks per CU: 26 880 bytes of LDS each");
    __shared__ k1t_shared_t<MODE> sh;
    const uint32_t t0 = blockIdx.x * spb;
    const uint32_t gbase = s_off[t0], gend = s_off[t0 + min(spb, N - t0)];
    /* the same for the whole block */
    if (gend - gbase + 3u <= K1T_STR_BYTES)
        k1_tiles_tile<MODE, false>(S, s_off, N, spb, s_aligned, T, first_idx, count, L, sh, gbase, gend);
    else
        k1_tiles_tile<MODE, true>(S, s_off, N, spb, s_aligned, T, first_idx, count, L, sh, gbase, gend);
}

/* Lists in ascending table order, in place.  Short lists: one thread each, insertion sort.
 * Long lists (a string that hundreds or thousands of patterns match): one block each, bitonic
 * merge network over the next power of two with every exchange ascending, so that the places
 * past the list's end act as +infinity and never move. */
#define K1_SORT_SHORT 32u

__global__ void __launch_bounds__(256)
k1_list_sort(const uint32_t *__restrict__ off, uint32_t N, uint32_t *__restrict__ idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    uint32_t *a = idx + off[i];
    const uint32_t n = off[i + 1] - off[i];
    if (n > K1_SORT_SHORT)
        return; /* k1_list_sort_long's */
    for (uint32_t k = 1; k < n; ++k) {
        const uint32_t v = a[k];
        uint32_t j = k;
        while (j > 0u && a[j - 1u] > v) {
            a[j] = a[j - 1u];
            --j;
        }
        a[j] = v;
    }
}

__global__ void __launch_bounds__(256)
k1_list_sort_long(const uint32_t *__restrict__ off, uint32_t N, uint32_t *__restrict__ idx)
{
    for (uint32_t i = blockIdx.x; i < N; i += gridDim.x) {
        const uint32_t n = off[i + 1] - off[i];
        if (n <= K1_SORT_SHORT)
            continue; /* the same for the whole block */
        uint32_t *a = idx + off[i];
        uint32_t m = 1;
        while (m < n)
            m <<= 1;
        for (uint32_t k = 2; k <= m; k <<= 1) {
            /* first step of a stage pairs x with its mirror image inside the block of k
             * places, the later ones x with x ^ j: every pair in ascending order */
            for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                for (uint32_t x = threadIdx.x; x < m; x += blockDim.x) {
                    const uint32_t y = j == (k >> 1) ? (x ^ (k - 1u)) : (x ^ j);
                    if (y > x && y < n) {
                        const uint32_t ax = a[x], ay = a[y];
                        if (ax > ay) {
                            a[x] = ay;
                            a[y] = ax;
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
    }
}

/*
 * Match lists from the short lists of a MODE 1 pass: one block per tile of the pass.
 *
 * Places first.  A string's list starts where the lists of all strings before it end: the
 * entries of the tiles before this one (tile_sum, added up here by the block itself, or
 * tile_off when k1_tile_offsets has run) plus those of the tile's earlier strings.  The
 * last tile's block reports the grand total and the number of overflowed short lists to
 * the host, which waits for that report and not for the end of the kernel.
 *
 * Then the lists.  A matching pattern text stands for one table position, or for several (its
 * copies, ascending in dup_idx): the list is the merge of these few ascending runs, cut at
 * `want` entries.  Sixteen lanes per string (four strings after each other) lay the runs out:
 * running sums of their lengths give every run its place in the list.  The entries of all the
 * tile's lists then stand side by side in a room in LDS (K1_EMIT_ROOM entries; lists of up to
 * K1_EMIT_PAIRS entries go there, which is nearly always all of them), a thread per entry: it
 * finds its list and run, fetches its table position, and counts the smaller entries of its
 * list - that count is its place.  A list that found no room ranks every entry by bisection
 * through the other runs in global memory, sixteen lanes per list.  The steps are separated by
 * barriers of the whole block, so that the loads of a step - for all the block's strings - are
 * in flight together, and a thread per entry keeps the lanes busy whatever the lists' lengths:
 * the kernel's time is its instruction count.
 * Nothing is written past `cap` entries (the caller's array; the host then reports the size that
 * is needed), and strings with more matching texts than a short list holds are left alone
 * (another pass fills every list then).
 */
#define K1_WIDE_LANES 16u
#define K1_EMIT_THREADS 256u
#define K1_EMIT_ROUNDS (K1T_MAX_SPB * K1_WIDE_LANES / K1_EMIT_THREADS)
#define K1_EMIT_ROOM 2048u /* entries of a tile's lists that are gathered in LDS */
#define K1_EMIT_SLOTS (K1_EMIT_ROOM / K1_EMIT_THREADS)
#define K1_EMIT_PAIRS 64u /* lists up to this length: gathered, every entry against every other */
#define K1_NO_COPIES 0xFFFFFFFFu
#define K1_REPORT_BITS 40u /* of a word of the report: the number; above them the call's mark */
#define K1_TILES_DIRECT 2048u /* up to this many tiles every block adds up the tiles before it by itself */
static_assert(K1_WIDE_LANES == K1T_LIST_CAP, "a lane per matching text");
static_assert(K1T_MAX_SPB <= 64u, "the strings of a tile are counted by one wave");
__global__ void __launch_bounds__(K1_EMIT_THREADS, 6)
k1_list_place_emit(const uint32_t *__restrict__ count, const uint32_t *__restrict__ nslots,
                   const uint2 *__restrict__ texts, const uint32_t *__restrict__ dup_start,
                   const uint32_t *__restrict__ dup_idx, uint32_t N, uint32_t spb, uint32_t K,
                   const unsigned long long *__restrict__ tile_sum, const uint32_t *__restrict__ tile_over,
                   const unsigned long long *__restrict__ tile_off, uint32_t *__restrict__ off,
                   uint32_t *__restrict__ idx, unsigned long long cap, unsigned long long *__restrict__ h_totals,
                   uint32_t seq)
{
    constexpr uint32_t GPR = K1_EMIT_THREADS / K1_WIDE_LANES; /* lists per round */
    /* sums over the tiles, added up in four cells each: the lanes' addresses differ, so every lane adds
     * for itself in LDS (with one address the compiler adds the lanes' values up in a scalar loop
     * first, lane by lane, which costs far more) */
    __shared__ unsigned long long before_sh[4];
    __shared__ uint32_t over_sh[4];
    __shared__ uint32_t ns_sh[K1T_MAX_SPB], want_sh[K1T_MAX_SPB], first_sh[K1T_MAX_SPB], room_at[K1T_MAX_SPB], fits_sh[K1T_MAX_SPB], room_used;
    /* a run: where its copies stand in dup_idx (K1_NO_COPIES: a text that stands once in the table) and
     * its table position (that of a text with copies: its slot number); its place in the list is in
     * rbase, its length the distance to the next one's place */
    __shared__ uint2 run[K1T_MAX_SPB][K1T_LIST_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t rbase[K1T_MAX_SPB][K1T_LIST_CAP + 4u]; /* [16] = the list's length */
    __shared__ uint4 all4[K1_EMIT_ROOM / 4u];
    uint32_t *all = (uint32_t *)all4;
    const uint32_t tid = threadIdx.x, l = tid % K1_WIDE_LANES;
    const uint32_t tile = blockIdx.x, ntiles = gridDim.x, t0 = tile * spb;
    const uint32_t nstr = min(spb, N - t0);
    const bool last = tile + 1u == ntiles;
    if (tid < 4u) {
        before_sh[tid] = 0ull;
        over_sh[tid] = 0u;
    }
    __syncthreads();
    /* step 1: everything that does not wait for anything else */
    uint2 text[K1_EMIT_ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_ROUNDS; ++k) {
        const uint32_t g = min(tid / K1_WIDE_LANES + GPR * k, nstr - 1u);
        /* past a string's last text the short list holds whatever the memory held, and the groups
         * of lanes past the tile's last string read that one's: neither is looked at */
        text[k] = make_uint2(0u, 0u);
        if (idx != nullptr && GPR * k < nstr)
            text[k] = texts[(size_t)(t0 + g) * K1T_LIST_CAP + l];
    }
    if (tid < K1T_MAX_SPB) {
        const bool has = tid < nstr;
        const uint32_t mine = t0 + min(tid, nstr - 1u);
        const uint32_t cnt = count[mine], ns = nslots[mine];
        ns_sh[tid] = has ? ns : 0u;
        want_sh[tid] = !has ? 0u : K != 0u && cnt > K ? K : cnt;
    }
    unsigned long long before = 0;
    uint32_t over = 0;
    if (tile_off == nullptr) {
        for (uint32_t t = tid; t < tile; t += K1_EMIT_THREADS)
            before += tile_sum[t];
        if (last)
            for (uint32_t t = tid; t < ntiles; t += K1_EMIT_THREADS)
                over += tile_over[t];
        if (before != 0ull)
            atomicAdd(&before_sh[tid & 3u], before);
        if (over != 0u)
            atomicAdd(&over_sh[tid & 3u], over);
    } else if (tid == 0u) {
        before_sh[0] = tile_off[tile];
        over_sh[0] = last ? (uint32_t)tile_off[ntiles + 1u] : 0u;
    }
    __syncthreads();
    /* step 2: the runs of every list (where their copies stand: asked for in one go), and the lists' places */
    uint32_t at_dup[K1_EMIT_ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_ROUNDS; ++k) {
        const uint32_t g = tid / K1_WIDE_LANES + GPR * k;
        if (GPR * k >= nstr)
            continue; /* the whole round is past the tile's last string */
        const uint32_t ns = ns_sh[g];
        const bool listed = idx != nullptr && ns <= K1T_LIST_CAP && l < ns && want_sh[g] != 0u;
        if (!listed)
            text[k].y = 0u; /* 0 in the lanes past the last text */
    }
    /* (a loop of its own: the four requests leave together, none waits for the one before) */
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_ROUNDS; ++k) {
        at_dup[k] = K1_NO_COPIES;
        if (GPR * k < nstr && text[k].y > 1u)
            at_dup[k] = dup_start[text[k].x];
    }
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_ROUNDS; ++k) {
        const uint32_t g = tid / K1_WIDE_LANES + GPR * k;
        if (GPR * k >= nstr)
            continue;
        /* Of a list cut at K, only the first K copies of a text can be among the first K entries, and
         * an entry that is has all the smaller ones among those: the runs are cut at K too, and the
         * places counted among what is left are the places in the whole list. */
        const uint32_t mult = K != 0u && text[k].y > K ? K : text[k].y;
        /* running sums over the sixteen lanes of a list: a row of the data-parallel primitives, whose
         * lanes without a lane that far to their left read 0 */
        uint32_t incl = mult;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);
        run[g][l] = make_uint2(at_dup[k], text[k].x);
        rbase[g][l] = incl - mult;
        if (l == K1_WIDE_LANES - 1u)
            rbase[g][K1_WIDE_LANES] = incl;
    }
    if (tid < 64u) {
        /* one wave: the strings' places in the caller's array, and in the block's room in LDS */
        const uint32_t w = over_sh[0] + over_sh[1] + over_sh[2] + over_sh[3];
        const unsigned long long tiles_before = before_sh[0] + before_sh[1] + before_sh[2] + before_sh[3];
        const uint32_t want = tid < K1T_MAX_SPB ? want_sh[tid] : 0u;
        uint32_t incl = want;
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (tid >= d)
                incl += o;
        }
        const unsigned long long place = tiles_before + (incl - want);
        if (tid < nstr)
            off[t0 + tid] = (uint32_t)place;
        if (tid < K1T_MAX_SPB)
            first_sh[tid] = incl - want;
        if (last && tid == 63u) {
            const unsigned long long total = tiles_before + incl;
            off[N] = (uint32_t)total;
            /* Two words for the host, each stored as a whole and each marked with the call's number
             * in its upper bits: the host waits for both marks, not for the end of the kernel, and no
             * fence (which would write back the whole L2) has to order anything. */
            const unsigned long long mark = (unsigned long long)seq << K1_REPORT_BITS;
            const unsigned long long most = (1ull << K1_REPORT_BITS) - 1ull;
            __hip_atomic_store(h_totals, mark | (total < most ? total : most), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(h_totals + 1, mark | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (idx == nullptr)
        return; /* the places only */
    __syncthreads();
    if (tid < 64u) {
        /* Room in LDS for the entries of the lists of up to K1_EMIT_PAIRS entries, all of them (a
         * list cut at K is gathered whole), each list's room a whole number of 16-byte words.
         * The first list that finds no room ends the part of the room that is used: those
         * behind it would find none either. */
        const uint32_t len = tid < ((nstr + GPR - 1u) & ~(GPR - 1u)) ? rbase[tid][K1_WIDE_LANES] : 0u;
        const uint32_t whole = len <= K1_EMIT_PAIRS ? (len + 3u) & ~3u : K1_EMIT_ROOM + 4u;
        uint32_t incl = whole;
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (tid >= d)
                incl += o;
        }
        const bool fits = incl <= K1_EMIT_ROOM;
        uint32_t used = fits ? incl : 0u; /* the largest end of a list that fits */
        for (int d = 32; d > 0; d >>= 1)
            used = max(used, (uint32_t)__shfl_xor(used, d));
        if (tid < K1T_MAX_SPB) {
            room_at[tid] = fits ? incl - whole : K1_EMIT_ROOM; /* past every place that is used */
            fits_sh[tid] = fits ? 1u : 0u;
        }
        if (tid == 0u)
            room_used = used;
    }
    __syncthreads();
    /* step 3: the entries themselves, a thread per place of the room and round, all asked for in
     * one go.  A place belongs to the last list that starts at or before it (lists without room
     * start where the next one does); entry e of a list belongs to the last run that starts at or
     * before e: the starts ascend, and those of the runs a list does not have equal its length. */
    const uint32_t used = room_used;
    uint32_t lead[K1_EMIT_SLOTS], owner[K1_EMIT_SLOTS], copy_at[K1_EMIT_SLOTS];
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_SLOTS; ++k) {
        const uint32_t s = tid + K1_EMIT_THREADS * k;
        lead[k] = 0xFFFFFFFFu; /* past a list's end: above every table position */
        owner[k] = 0xFFFFFFFFu;
        copy_at[k] = 0xFFFFFFFFu;
        if (s < used) {
            uint32_t g = 0;
#pragma unroll
            for (uint32_t step = K1T_MAX_SPB / 2u; step > 0u; step >>= 1)
                if (room_at[g + step] <= s)
                    g += step;
            const uint32_t e = s - room_at[g];
            if (e < rbase[g][K1_WIDE_LANES]) {
                const uint4 *row = (const uint4 *)rbase[g];
                const uint4 b0 = row[0], b1 = row[1];
                uint32_t r = (b0.y <= e) + (b0.z <= e) + (b0.w <= e) + (b1.x <= e) + (b1.y <= e) + (b1.z <= e) +
                             (b1.w <= e);
                if (r == 7u) { /* few lists have more than eight runs */
                    const uint4 b2 = row[2], b3 = row[3];
                    r += (b2.x <= e) + (b2.y <= e) + (b2.z <= e) + (b2.w <= e) + (b3.x <= e) + (b3.y <= e) +
                         (b3.z <= e) + (b3.w <= e);
                }
                const uint2 mine = run[g][r];
                const uint32_t base = rbase[g][r];
                owner[k] = g;
                lead[k] = mine.y;
                if (mine.x != K1_NO_COPIES)
                    copy_at[k] = mine.x + (e - base);
            }
        }
    }
    /* (a loop of its own: the requests leave together) */
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_SLOTS; ++k)
        if (copy_at[k] != 0xFFFFFFFFu)
            lead[k] = dup_idx[copy_at[k]];
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_SLOTS; ++k)
        if (tid + K1_EMIT_THREADS * k < used)
            all[tid + K1_EMIT_THREADS * k] = lead[k];
    __syncthreads();
    /* step 4: every entry's place in its list */
    const unsigned long long tiles_before = before_sh[0] + before_sh[1] + before_sh[2] + before_sh[3];
    uint32_t rank[K1_EMIT_SLOTS];
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_SLOTS; ++k) {
        const uint32_t g = owner[k];
        rank[k] = 0xFFFFFFFFu;
        if (g == 0xFFFFFFFFu)
            continue;
        const uint32_t total = rbase[g][K1_WIDE_LANES], at = room_at[g];
        uint32_t below = 0;
#pragma unroll 1
        for (uint32_t o = at / 4u; o < (at + total + 3u) / 4u; ++o) {
            const uint4 x = all4[o];
            below += (x.x < lead[k]) + (x.y < lead[k]) + (x.z < lead[k]) + (x.w < lead[k]);
        }
        rank[k] = below;
    }
    /* (the stores in a loop of their own, behind all the counting and with their addresses ready: a
     * register that a store still reads cannot be written, so anything in between would wait for it) */
    uint32_t *to[K1_EMIT_SLOTS];
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_SLOTS; ++k) {
        const uint32_t g = owner[k];
        to[k] = nullptr;
        if (g == 0xFFFFFFFFu)
            continue;
        const unsigned long long place = tiles_before + first_sh[g];
        const unsigned long long room = place < cap ? cap - place : 0ull; /* entries of this list that fit */
        if (rank[k] < want_sh[g] && rank[k] < room)
            to[k] = idx + (place + rank[k]); /* table positions are distinct: so are the places */
    }
#pragma unroll
    for (uint32_t k = 0; k < K1_EMIT_SLOTS; ++k)
        if (to[k] != nullptr)
            *to[k] = lead[k];
    /* the lists without room: sixteen lanes per list, every entry's place by bisection through the
     * other runs in global memory */
#pragma unroll 1
    for (uint32_t k = 0; k < K1_EMIT_ROUNDS; ++k) {
        const uint32_t g = tid / K1_WIDE_LANES + GPR * k;
        if (GPR * k >= nstr)
            break;
        const uint32_t total = rbase[g][K1_WIDE_LANES], want = want_sh[g];
        if (total == 0u || fits_sh[g] != 0u)
            continue;
        const unsigned long long place = tiles_before + first_sh[g];
        uint32_t *dst = idx + (size_t)place;
        const unsigned long long room = place < cap ? cap - place : 0ull;
        const uint32_t ns = ns_sh[g];
        for (uint32_t e = l; e < total; e += K1_WIDE_LANES) {
            uint32_t r = 0;
            while (rbase[g][r + 1u] <= e)
                ++r;
            const uint32_t p = e - rbase[g][r];
            const uint32_t v = run[g][r].x != K1_NO_COPIES ? dup_idx[run[g][r].x + p] : run[g][r].y;
            uint32_t rank = p;
            for (uint32_t o = 0; o < ns; ++o) {
                if (o == r)
                    continue;
                const uint32_t len = rbase[g][o + 1u] - rbase[g][o];
                if (run[g][o].x == K1_NO_COPIES) {
                    rank += run[g][o].y < v ? 1u : 0u;
                } else {
                    const uint32_t *q = dup_idx + run[g][o].x;
                    uint32_t lo = 0, hi = len; /* entries of run o below v */
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (q[mid] < v)
                            lo = mid + 1u;
                        else
                            hi = mid;
                    }
                    rank += lo;
                }
            }
            if (rank < want && rank < room)
                dst[rank] = v;
        }
    }
}

/* Batches of more tiles than every block cares to add up by itself: tile_off[t] = entries of the
 * tiles before t, tile_off[ntiles] = all of them, tile_off[ntiles + 1] = overflowed short lists.
 * One block walks the tiles, 1024 at a step. */
__global__ void __launch_bounds__(1024)
k1_tile_offsets(const unsigned long long *__restrict__ tile_sum, const uint32_t *__restrict__ tile_over,
                uint32_t ntiles, unsigned long long *__restrict__ tile_off)
{
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t wover[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long carry = 0;
    uint32_t over = 0;
    for (uint32_t c = 0; c < ntiles; c += 1024u) {
        const uint32_t t = c + tid;
        const unsigned long long v = t < ntiles ? tile_sum[t] : 0ull;
        uint32_t ov = t < ntiles ? tile_over[t] : 0u;
        unsigned long long incl = v;
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d);
            if (lane >= d)
                incl += o;
        }
        for (int d = 32; d > 0; d >>= 1)
            ov += __shfl_xor(ov, d);
        if (lane == 63u)
            wsum[wave] = incl;
        if (lane == 0u)
            wover[wave] = ov;
        __syncthreads();
        unsigned long long below = 0, whole = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            below += w < wave ? wsum[w] : 0ull;
            whole += wsum[w];
            over += wover[w];
        }
        if (t < ntiles)
            tile_off[t] = carry + below + incl - v;
        carry += whole;
        __syncthreads();
    }
    if (tid == 0u) {
        tile_off[ntiles] = carry;
        tile_off[ntiles + 1u] = over;
    }
}

/* two 64-bit totals from device memory into pinned host memory (one small launch is quicker
 * than a device-to-host copy) */
__global__ void k1_publish2(const unsigned long long *__restrict__ d, unsigned long long *__restrict__ h)
{
    if (threadIdx.x < 2u)
        h[threadIdx.x] = d[threadIdx.x];
}

/* the first entries of every sorted list: from[i] .. -> to[i] .. to[i+1] */
__global__ void __launch_bounds__(256)
k1_list_take(const uint32_t *__restrict__ from_off, const uint32_t *__restrict__ from_idx,
             const uint32_t *__restrict__ to_off, uint32_t N, uint32_t *__restrict__ to_idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const uint32_t a = from_off[i], b = to_off[i], n = to_off[i + 1] - b;
    for (uint32_t k = 0; k < n; ++k)
        to_idx[b + k] = from_idx[a + k];
}

__global__ void __launch_bounds__(256)
k1_driver_mod(const uint32_t *__restrict__ first_idx, const uint32_t *__restrict__ a_mod, uint32_t N,
              uint32_t *__restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        uint32_t j = first_idx[i];
        out[i] = (j == MCX_NO_MATCH) ? MCX_NO_MATCH : a_mod[j];
    }
}

/* hw[m] = 1 for every module owning a pattern that starts with a bus tag */
__global__ void __launch_bounds__(256)
k1_hw_flags(const uint8_t *__restrict__ hw_tag, const uint32_t *__restrict__ a_mod, uint32_t P,
            uint32_t n_modules, uint8_t *__restrict__ hw)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P && hw_tag[j]) {
        uint32_t m = a_mod[j];
        if (m < n_modules)
            hw[m] = 1; /* every writer stores the same byte */
    }
}

extern "C" {

/* ---- K1 ----------------------------------------------------------------- */

int mcx_alias_table_load(mcx_ctx *ctx, const char *arena, const uint32_t *a_off, const uint32_t *a_mod,
                         uint32_t P, uint32_t flags)
{
    if (!ctx || !a_off || !a_mod || (P && !arena && a_off[P] != 0))
        return MCX_E_BAD_ARG;
    if (flags & ~MCX_TABLE_NORMALIZE_DASH) {
        ctx->err = "unknown table flag";
        return MCX_E_BAD_ARG;
    }
    if (P > K1_MAX_PATTERNS) {
        ctx->err = "alias table too large: at most 2^26 patterns (shard it)";
        return MCX_E_BAD_ARG;
    }
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    mcx_compiled_table t;
    std::string err;
    mcx_compile_options copt;
    copt.threads = ctx->tune_threads;
    copt.sub_min = ctx->tune_sub_min;
    int rc = mcx_compile_table(arena, a_off, P, (flags & MCX_TABLE_NORMALIZE_DASH) != 0, t, err, &copt);
    if (rc != 0) {
        ctx->err = err;
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    mcx_free_table(ctx);
    if ((rc = mcx_dev_upload(ctx, &ctx->d_slot, t.slot.data(), t.slot.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_ext, t.ext.data(), t.ext.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_sig, t.sig.data(), t.sig.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_lens, t.lens.data(), t.lens.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_dup_start, t.dup_start.data(), t.dup_start.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_dup_idx, t.dup_idx.data(), t.dup_idx.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_slots, t.slots.data(), t.slots.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_tags, t.tags.data(), t.tags.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_sub, t.sub.data(), t.sub.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_sub_tags, t.sub_tags.data(), t.sub_tags.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_sets, t.sets.data(), t.sets.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_amod, a_mod, (size_t)P)) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_hwtag, t.hw_tag.data(), t.hw_tag.size())) != 0) {
        mcx_free_table(ctx);
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->P = P;
    ctx->n_indexed = t.n_distinct;
    ctx->slot_mask = (uint32_t)t.slots.size() - 1u;
    ctx->sub_mask = (uint32_t)t.sub.size() - 1u;
    ctx->n_sub_groups = t.n_sub_groups;
    ctx->max_group = t.max_group;
    ctx->nlens = (uint32_t)t.lens.size();
    ctx->norm_dash = (flags & MCX_TABLE_NORMALIZE_DASH) != 0;
    ctx->table_bytes = 4 * (uint64_t)(t.slot.size() + t.ext.size() + t.sig.size() + t.sets.size() + t.lens.size() +
                                      t.dup_start.size() + t.dup_idx.size()) +
                       sizeof(mcx_slot_t) * (uint64_t)t.slots.size() + t.tags.size() +
                       sizeof(mcx_sub_t) * (uint64_t)t.sub.size() + t.sub_tags.size() + 4 * (uint64_t)P +
                       t.hw_tag.size();
    ctx->bulk_min_set = ctx->tune_tiles_min_set;
    ctx->bulk_min = ctx->tune_tiles_min;
    ctx->list_two_pass = ctx->tune_two_pass;
    ctx->table_loaded = true;
    return MCX_OK;
}

int mcx_alias_table_info(mcx_ctx *ctx, uint32_t *P_out, uint64_t *device_bytes_out)
{
    if (!ctx)
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    if (P_out)
        *P_out = ctx->P;
    if (device_bytes_out)
        *device_bytes_out = ctx->table_bytes;
    return MCX_OK;
}

#ifdef MCX_K1_STAMPS
static unsigned long long *g_k1_stamps = nullptr;
static uint32_t g_k1_stamp_blocks = 0;
/* diagnostic build only: copy out the stamps of the last launch (8 per block) */
extern "C" int mcx_debug_k1_stamps(unsigned long long *out, uint32_t max_blocks)
{
    uint32_t n = g_k1_stamp_blocks < max_blocks ? g_k1_stamp_blocks : max_blocks;
    if (hipMemcpy(out, g_k1_stamps, 128u * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (int)n;
}
#endif

/* list_off == nullptr: first_idx / count.  Otherwise the list variant: every
 * match is written into d_list_idx at the place list_off (the exclusive sums
 * of a counting pass over the same strings) gives its string. */
static void k1_fill_table(const mcx_ctx *ctx, k1_table_t &T)
{
    T.slot = ctx->d_slot;
    T.ext = ctx->d_ext;
    T.sig = ctx->d_sig;
    T.lens = ctx->d_lens;
    T.slots = ctx->d_slots;
    T.tags = ctx->d_tags;
    T.sub = ctx->d_sub;
    T.sub_tags = ctx->d_sub_tags;
    T.sub_mask = ctx->sub_mask;
    T.sets = ctx->d_sets;
    T.slot_mask = ctx->slot_mask;
    T.nlens = ctx->nlens;
    T.nlens_magic = ctx->nlens ? (uint32_t)((((uint64_t)1 << 32) + ctx->nlens - 1u) / ctx->nlens) : 0u;
    T.norm_dash = ctx->norm_dash ? 1u : 0u;
}

/* Which of the two kernels a batch gets.  The small-batch kernel (a handful of strings per
 * block, occurrence tables, eight lanes per program run) is faster for a few thousand strings;
 * the tile kernel (up to 64 strings per block) from about 7000 strings on (measured against
 * 3 x 10^4 patterns, small / tile: 23 / 28 us at 4096 strings, 29 / 30 at 6144, 33 / 32 at 8192,
 * 42 / 34 at 12 288, 71 / 42 at 24 576).  Only the tile kernel uses the second index of the large groups, and
 * against a table with a very large group (every "of:" line of a table of 3 x 10^6 patterns
 * shares one prefix) that decides it whatever the batch: 1000 strings take 28 us with it
 * and 78 us without.  mcx_ctx_tune(MCX_TUNE_TILES_MIN) replaces the batch size for the tables
 * loaded after it (results do not depend on the kernel). */
static bool k1_use_tiles(const mcx_ctx *ctx, uint32_t N)
{
    if (ctx->bulk_min_set)
        return N >= ctx->bulk_min;
    if (N > 8192u) /* beyond 1024 blocks of eight strings the small-batch kernel needs a second round */
        return true;
    /* the larger the largest group, the earlier the second index pays: any batch from 8192
     * patterns in one group on */
    return ctx->n_sub_groups != 0u &&
           (ctx->max_group >= 8192u || (uint64_t)N * ctx->max_group >= 24000000ull);
}

/* Strings per block of the tile kernel: the chip holds K1T_RESIDENT blocks at once, and a launch
 * of a few more than that ends with a nearly empty second round (100 000 strings: 139 us as 1563
 * blocks of 64 strings, 126 as 2041 of 49).  So: the number of rounds the batch needs with full
 * tiles, and then the tile size that fills those rounds evenly. */
static uint32_t k1t_strings_per_block(uint32_t N)
{
    const uint64_t full = (uint64_t)K1T_MAX_SPB * K1T_RESIDENT;
    const uint64_t rounds = (N + full - 1u) / full;
    if (rounds >= 4u)
        return K1T_MAX_SPB; /* many rounds: the last one's emptiness no longer shows, full tiles do */
    const uint64_t blocks = rounds * K1T_RESIDENT;
    uint32_t spb = (uint32_t)((N + blocks - 1u) / blocks);
    if (spb > K1T_MAX_SPB)
        spb = K1T_MAX_SPB;
    return spb ? spb : 1u;
}

static int launch_alias_match(mcx_ctx *ctx, const uint8_t *d_S, const uint32_t *d_s_off, uint32_t N,
                              uint32_t *d_first, uint32_t *d_count)
{
    if (N == 0)
        return MCX_OK;
    k1_table_t T;
    k1_fill_table(ctx, T);
    const uint32_t s_aligned = ((uintptr_t)d_S & 3u) == 0u ? 1u : 0u;
    /* up to 1024 blocks, four to a CU: a block's time is mostly its chain of dependent loads, and a
     * second string costs it a tenth more, while a fifth block per CU costs every block of that CU
     * (2000 strings: 12.8 us as 1000 blocks of two, 18.4 us as 2000 blocks of one) */
    uint32_t spb = (N + 1023u) / 1024u;
#ifndef MCX_K1_STAMPS
    if (k1_use_tiles(ctx, N)) {
        spb = k1t_strings_per_block(N);
        k1_match_tiles<0><<<(N + spb - 1u) / spb, K1T_THREADS, 0, ctx->stream>>>(d_S, d_s_off, N, spb, s_aligned, T,
                                                                                  d_first, d_count, k1_list_t{});
        HIPCHK(ctx, hipGetLastError());
        return MCX_OK;
    }
#endif
    if (spb > K1_MAX_SPB)
        spb = K1_MAX_SPB;
    const uint32_t blocks = (N + spb - 1u) / spb;
    /* staging buffer: 256 bytes per string of the tile, between 1 KiB and 8 KiB;
     * a tile that needs more is read in place */
    uint32_t str_cap = spb * 256u;
    if (str_cap < 1024u)
        str_cap = 1024u;
    if (str_cap > K1_LDS_STR_BYTES)
        str_cap = K1_LDS_STR_BYTES;
    const size_t lds = 4 * ((size_t)spb * K1_OCC_WORDS + (size_t)str_cap / 4 + 5 + 3 * (size_t)spb * T.nlens + 1);
#ifdef MCX_K1_STAMPS
    static uint32_t stamp_cap = 0;
    if (blocks > stamp_cap) {
        (void)hipDeviceSynchronize();
        (void)hipFree(g_k1_stamps);
        (void)hipMalloc((void **)&g_k1_stamps, 16u * 8u * (size_t)blocks);
        stamp_cap = blocks;
    }
    g_k1_stamp_blocks = blocks;
    k1_match_small<<<blocks, K1_THREADS, lds, ctx->stream>>>(d_S, d_s_off, N, spb, str_cap, s_aligned, T, d_first,
                                                             d_count, g_k1_stamps);
#else
    k1_match_small<<<blocks, K1_THREADS, lds, ctx->stream>>>(d_S, d_s_off, N, spb, str_cap, s_aligned, T, d_first,
                                                             d_count);
#endif
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}

int mcx_alias_match_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, void *d_first_idx,
                        void *d_count)
{
    if (!ctx || !d_s_off || !d_first_idx || !d_count)
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    return launch_alias_match(ctx, (const uint8_t *)d_S, (const uint32_t *)d_s_off, N,
                              (uint32_t *)d_first_idx, (uint32_t *)d_count);
}

int mcx_alias_match(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, uint32_t *first_idx,
                    uint32_t *count)
{
    if (!ctx || !s_off || (N && (!first_idx || !count)))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    if (N == 0)
        return MCX_OK;
    if (!S && s_off[N] != 0)
        return MCX_E_BAD_ARG;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    for (uint32_t i = 0; i < N; ++i)
        if (s_off[i + 1] < s_off[i]) {
            ctx->err = "s_off not monotonic";
            return MCX_E_BAD_ARG;
        }
    /* staging area kept by the context and grown on demand: one allocation
     * holds the strings, their offsets and both result arrays */
    const size_t sbytes = s_off[N];
    const size_t o_off = (sbytes + 255) & ~(size_t)255, o_first = o_off + ((4 * ((size_t)N + 1) + 255) & ~(size_t)255);
    const size_t o_count = o_first + ((4 * (size_t)N + 255) & ~(size_t)255);
    const size_t need = o_count + 4 * (size_t)N;
    /* Small calls are all transfer set-up: four copies between pageable memory and the
     * device cost several times the kernel.  They use no copies at all: strings and
     * offsets are put into pinned host memory, the kernel reads them there (a few reads
     * over the bus per block) and writes the results there. */
    if (need <= MCX_ZERO_COPY_MAX) {
        uint8_t *hb = nullptr, *db = nullptr;
        const int rc_io = mcx_pinned_io(ctx, &hb, &db);
        if (rc_io != MCX_OK)
            return rc_io;
        if (sbytes)
            memcpy(hb, S, sbytes);
        memcpy(hb + o_off, s_off, 4 * ((size_t)N + 1));
        int rc0 = launch_alias_match(ctx, db, (const uint32_t *)(db + o_off), N, (uint32_t *)(db + o_first),
                                     (uint32_t *)(db + o_count));
        if (rc0 != MCX_OK)
            return rc0;
        if ((rc0 = mcx_signal_and_wait(ctx)) != MCX_OK)
            return rc0;
        memcpy(first_idx, hb + o_first, 4 * (size_t)N);
        memcpy(count, hb + o_count, 4 * (size_t)N);
        return MCX_OK;
    }
    uint8_t *base = nullptr;
    {
        const int rc_io = mcx_io_reserve(ctx, need, &base);
        if (rc_io != MCX_OK)
            return rc_io;
    }
    uint32_t *d_off = (uint32_t *)(base + o_off), *d_first = (uint32_t *)(base + o_first);
    uint32_t *d_count = (uint32_t *)(base + o_count);
    if (sbytes)
        HIPCHK(ctx, hipMemcpyAsync(base, S, sbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_off, s_off, 4 * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_alias_match(ctx, base, d_off, N, d_first, d_count);
    if (rc != MCX_OK)
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(first_idx, d_first, 4 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(count, d_count, 4 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MCX_OK;
}

/* Wait for the two marked words k1_list_place_emit leaves in pinned host memory (spinning on them
 * costs a few microseconds, a stream synchronisation about 25); after 200 us without them,
 * synchronise the stream instead, which also reports a failed launch. */
static int k1_wait_report(mcx_ctx *ctx, uint32_t seq, unsigned long long *total, uint32_t *overflows)
{
    volatile unsigned long long *h = (volatile unsigned long long *)(ctx->h_pinned + MCX_PINNED_LIST_WORD);
    const unsigned long long most = (1ull << K1_REPORT_BITS) - 1ull;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (uint32_t spins = 0;; ++spins) {
        const unsigned long long a = h[0], b = h[1];
        if ((a >> K1_REPORT_BITS) == seq && (b >> K1_REPORT_BITS) == seq) {
            *total = a & most;
            *overflows = (uint32_t)(b & most);
            return MCX_OK;
        }
        if (synced) {
            ctx->err = "the match lists' report did not arrive";
            return MCX_E_HIP;
        }
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            synced = true;
        }
    }
}

/* the tile kernel in one of its list modes over the whole batch */
static int launch_tiles_list(mcx_ctx *ctx, int mode, const uint8_t *d_S, const uint32_t *d_s_off, uint32_t N,
                             uint32_t *d_first, uint32_t *d_count, const k1_list_t &L)
{
    k1_table_t T;
    k1_fill_table(ctx, T);
    const uint32_t spb = k1t_strings_per_block(N);
    const uint32_t blocks = (N + spb - 1u) / spb;
    const uint32_t s_aligned = ((uintptr_t)d_S & 3u) == 0u ? 1u : 0u;
    if (mode == 1)
        k1_match_tiles<1><<<blocks, K1T_THREADS, 0, ctx->stream>>>(d_S, d_s_off, N, spb, s_aligned, T, d_first,
                                                                    d_count, L);
    else
        k1_match_tiles<2><<<blocks, K1T_THREADS, 0, ctx->stream>>>(d_S, d_s_off, N, spb, s_aligned, T, nullptr,
                                                                    nullptr, L);
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}

int mcx_alias_match_list_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, uint32_t K,
                             void *d_list_off, void *d_list_idx, uint64_t cap, uint64_t *total_out)
{
    if (!ctx || !d_s_off || !d_list_off || !total_out || (cap && !d_list_idx))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    *total_out = 0;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    if (N == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_list_off, 0, 4, ctx->stream));
        return MCX_OK;
    }
    const uint32_t nb = (N + SCAN_ITEMS - 1u) / SCAN_ITEMS, grid = (N + 255u) / 256u;
    const uint32_t spb = k1t_strings_per_block(N), ntiles = (N + spb - 1u) / spb;
    /* scratch: first, count, number of matching texts, offsets of the full lists (fallback),
     * block sums and total of its scan, the tiles' sums, overflows and offsets, and the short lists */
    const size_t words = up256(4 * ((size_t)N + 1));
    const size_t o_count = words, o_ns = 2 * words, o_full = 3 * words, o_bsum = 4 * words;
    const size_t o_total = o_bsum + up256(4 * (size_t)nb), o_tsum = o_total + 256;
    const size_t o_tover = o_tsum + up256(8 * (size_t)ntiles), o_toff = o_tover + up256(4 * (size_t)ntiles);
    const size_t o_texts = o_toff + up256(8 * ((size_t)ntiles + 2));
    void *base;
    int rc = mcx_scratch(ctx, o_texts + up256(8 * (size_t)N * K1T_LIST_CAP), &base);
    if (rc != MCX_OK)
        return rc;
    uint32_t *d_first = (uint32_t *)base, *d_count = (uint32_t *)((char *)base + o_count);
    uint32_t *d_ns = (uint32_t *)((char *)base + o_ns);
    uint32_t *d_full = (uint32_t *)((char *)base + o_full), *d_bsum = (uint32_t *)((char *)base + o_bsum);
    unsigned long long *d_total = (unsigned long long *)((char *)base + o_total);
    unsigned long long *d_tsum = (unsigned long long *)((char *)base + o_tsum);
    uint32_t *d_tover = (uint32_t *)((char *)base + o_tover);
    unsigned long long *d_toff = (unsigned long long *)((char *)base + o_toff);
    uint2 *d_texts = (uint2 *)((char *)base + o_texts);
    unsigned long long *h_total = (unsigned long long *)(ctx->h_pinned + MCX_PINNED_LIST_WORD);
    k1_list_t L{nullptr, nullptr, ctx->d_dup_start, ctx->d_dup_idx, d_ns, d_texts, d_tsum, d_tover, K};
    /* one pass: counts, first matches, the short lists and what every tile's lists hold together;
     * then one launch that places the lists and writes them.  It does so before the host has seen
     * the total (and keeps inside the caller's array by itself): the device does not sit idle
     * through the host's round trip. */
    if ((rc = launch_tiles_list(ctx, 1, (const uint8_t *)d_S, (const uint32_t *)d_s_off, N, d_first, d_count, L)) !=
        MCX_OK)
        return rc;
    /* the pinned words as the device addresses them */
    unsigned long long *dh_total = (unsigned long long *)(ctx->d_pinned + MCX_PINNED_LIST_WORD);
    const bool direct = ntiles <= K1_TILES_DIRECT;
    /* the report's words start out unmarked (they serve other calls' read-backs too); the mark is never 0 */
    const uint32_t seq = (mcx_next_seq(ctx) & 0xFFFFFFu) | 0x800000u;
    h_total[0] = 0ull;
    h_total[1] = 0ull;
    if (!direct)
        k1_tile_offsets<<<1, 1024, 0, ctx->stream>>>(d_tsum, d_tover, ntiles, d_toff);
    const bool emitted = !ctx->list_two_pass && d_list_idx != nullptr;
    k1_list_place_emit<<<ntiles, K1_EMIT_THREADS, 0, ctx->stream>>>(
        d_count, d_ns, d_texts, ctx->d_dup_start, ctx->d_dup_idx, N, spb, K, d_tsum, d_tover,
        direct ? nullptr : d_toff, (uint32_t *)d_list_off, emitted ? (uint32_t *)d_list_idx : nullptr,
        (unsigned long long)cap, dh_total, seq);
    HIPCHK(ctx, hipGetLastError());
    /* the totals are out as soon as the last tile's lists have their places: the lists themselves
     * follow in stream order, like the results of every other *_dev call */
    unsigned long long kept = 0;
    uint32_t overflows = 0;
    if ((rc = k1_wait_report(ctx, seq, &kept, &overflows)) != MCX_OK)
        return rc;
    *total_out = kept;
    if (kept > 0xFFFFFFFFull) {
        ctx->err = "more than 2^32 - 1 matches: split the batch";
        return MCX_E_TOO_LARGE;
    }
    if (kept > cap) {
        ctx->err = "list_idx too small: " + std::to_string(kept) + " entries needed";
        return MCX_E_TOO_LARGE;
    }
    if (kept == 0)
        return MCX_OK;
    if (overflows == 0u && emitted)
        return MCX_OK; /* written already */
    /* some string is matched by more pattern texts than a short list holds (or the test seam
     * asks for it): the lists are filled by a second pass of the kernel and sorted */
    uint32_t *full_off = (uint32_t *)d_list_off, *full_idx = (uint32_t *)d_list_idx;
    if (K) {
        mcx_scan_enqueue(ctx->stream, d_count, N, d_bsum, d_total, d_full);
        k1_publish2<<<1, 64, 0, ctx->stream>>>(d_total, dh_total);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const unsigned long long all = h_total[0];
        if (all > 0xFFFFFFFFull) {
            ctx->err = "more than 2^32 - 1 matches: split the batch";
            return MCX_E_TOO_LARGE;
        }
        if (hipMalloc((void **)&full_idx, 4 * (size_t)all) != hipSuccess) {
            ctx->err = "hipMalloc of the full match lists failed";
            return MCX_E_NOMEM;
        }
        full_off = d_full;
    }
    L.off = full_off;
    L.idx = full_idx;
    rc = launch_tiles_list(ctx, 2, (const uint8_t *)d_S, (const uint32_t *)d_s_off, N, nullptr, nullptr, L);
    if (rc == MCX_OK) {
        k1_list_sort<<<grid, 256, 0, ctx->stream>>>(full_off, N, full_idx);
        k1_list_sort_long<<<N < 4096u ? N : 4096u, 256, 0, ctx->stream>>>(full_off, N, full_idx);
        if (K)
            k1_list_take<<<grid, 256, 0, ctx->stream>>>(full_off, full_idx, (const uint32_t *)d_list_off, N,
                                                        (uint32_t *)d_list_idx);
        if (hipGetLastError() != hipSuccess)
            rc = MCX_E_HIP;
    }
    if (K) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(full_idx);
    }
    return rc;
}

int mcx_alias_match_list(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, uint32_t K,
                         uint32_t *list_off, uint32_t *list_idx, uint64_t cap, uint64_t *total_out)
{
    if (!ctx || !s_off || !list_off || !total_out || (cap && !list_idx))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    *total_out = 0;
    list_off[0] = 0;
    if (N == 0)
        return MCX_OK;
    if (!S && s_off[N] != 0)
        return MCX_E_BAD_ARG;
    for (uint32_t i = 0; i < N; ++i)
        if (s_off[i + 1] < s_off[i]) {
            ctx->err = "s_off not monotonic";
            return MCX_E_BAD_ARG;
        }
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    const size_t sbytes = s_off[N];
    const size_t o_off = up256(sbytes ? sbytes : 1), o_loff = o_off + up256(4 * ((size_t)N + 1));
    const size_t o_idx = o_loff + up256(4 * ((size_t)N + 1)), bytes = o_idx + up256(4 * (size_t)(cap ? cap : 1));
    /* strings, offsets and both result arrays in the context's staging area (kept between calls) */
    char *d = nullptr;
    {
        uint8_t *base = nullptr;
        const int rc_io = mcx_io_reserve(ctx, bytes, &base);
        if (rc_io != MCX_OK)
            return rc_io;
        d = (char *)base;
    }
    hipError_t e = hipSuccess;
    if (sbytes)
        e = hipMemcpyAsync(d, S, sbytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + o_off, s_off, 4 * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream);
    int rc = MCX_OK;
    if (e != hipSuccess) {
        ctx->err = std::string("copy in: ") + hipGetErrorString(e);
        rc = MCX_E_HIP;
    }
    if (rc == MCX_OK)
        rc = mcx_alias_match_list_dev(ctx, d, d + o_off, N, K, d + o_loff, cap ? d + o_idx : nullptr, cap, total_out);
    /* the offsets are valid even when list_idx was too small */
    if (rc == MCX_OK || (rc == MCX_E_TOO_LARGE && *total_out > cap && *total_out <= 0xFFFFFFFFull)) {
        e = hipMemcpyAsync(list_off, d + o_loff, 4 * ((size_t)N + 1), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == MCX_OK && *total_out)
            e = hipMemcpyAsync(list_idx, d + o_idx, 4 * (size_t)*total_out, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->err = std::string("copy back: ") + hipGetErrorString(e);
            rc = MCX_E_HIP;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

int mcx_alias_driver_mod_dev(mcx_ctx *ctx, const void *d_first_idx, uint32_t N, void *d_driver_mod)
{
    if (!ctx || (N && (!d_first_idx || !d_driver_mod)))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    if (N == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    k1_driver_mod<<<(N + 255) / 256, 256, 0, ctx->stream>>>((const uint32_t *)d_first_idx, ctx->d_amod, N,
                                                          (uint32_t *)d_driver_mod);
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}

int mcx_alias_hw_flags(mcx_ctx *ctx, uint32_t n_modules, uint8_t *hw_alias)
{
    if (!ctx || (n_modules && !hw_alias))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded)
        return MCX_E_NOT_LOADED;
    if (n_modules == 0)
        return MCX_OK;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    uint8_t *d_hw = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_hw, n_modules));
    int rc = mcx_k1_hw_flags_dev(ctx, n_modules, d_hw);
    hipError_t e = hipSuccess;
    if (rc == MCX_OK)
        e = hipMemcpyAsync(hw_alias, d_hw, n_modules, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_hw);
    if (rc != MCX_OK)
        return rc;
    if (e != hipSuccess) {
        ctx->err = hipGetErrorString(e);
        return MCX_E_HIP;
    }
    return MCX_OK;
}

} /* extern "C" */

int mcx_k1_hw_flags_dev(mcx_ctx *ctx, uint32_t n_modules, uint8_t *d_hw)
{
    if (n_modules == 0)
        return MCX_OK;
    HIPCHK(ctx, hipMemsetAsync(d_hw, 0, n_modules, ctx->stream));
    if (ctx->P)
        k1_hw_flags<<<(ctx->P + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_hwtag, ctx->d_amod, ctx->P, n_modules,
                                                               d_hw);
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}
