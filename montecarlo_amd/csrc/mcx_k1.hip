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

/* ------------------------------------------------------------------------- */
/* K1: alias match                                                            */
/* ------------------------------------------------------------------------- */

#define K1_THREADS 256
#define K1_MAX_SPB 16                       /* strings per block, at most      */
#define K1_OCC_WORDS 512                     /* 256 x 64-bit position sets      */
#define K1_LDS_STR_BYTES 16384              /* staging buffer for a tile       */
#define K1_MAX_HITS (K1_MAX_SPB * (MCX_KEY_MAX + 1))

/* what the match kernel needs from the loaded table; passed by value, so the
 * key lengths sit in the kernel argument segment */
struct k1_table_t {
    const uint32_t *slot;   /* n_indexed x MCX_SLOT_WORDS                       */
    const uint32_t *ext;    /* overflow programs                               */
    const uint2 *sigorig;   /* n_indexed x {gram signature, table position}    */
    const mcx_slot_t *slots;
    const uint32_t *sets;
    uint32_t slot_mask;
    uint32_t nlens;
    uint32_t lens[MCX_KEY_MAX + 1];
};

typedef const __attribute__((address_space(3))) uint32_t *lds_words_t;

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
 * byte reads, bytes past the end read as 0 */
struct GlobalStr {
    const uint8_t *p;
    uint32_t len;
    __device__ __forceinline__ uint32_t byte(uint32_t q) const { return q < len ? (uint32_t)p[q] : 0u; }
    __device__ __forceinline__ uint32_t word(uint32_t q) const
    {
        return byte(q) | (byte(q + 1) << 8) | (byte(q + 2) << 16) | (byte(q + 3) << 24);
    }
};

struct GlobalProg {
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
struct LdsBloom {
    lds_words_t w;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return w[i]; }
};

#define K1_RING 1024u       /* survivors waiting for a program run, per block    */
#define K1_UNIT 256u        /* patterns screened per wave step (4 per lane)       */
#define K1_WAVES (K1_THREADS / 64)

struct __attribute__((aligned(8))) k1_sig4_t {
    uint2 v[4]; /* {signature, table position} of four consecutive patterns */
};

/*
 * One block (4 waves) = `spb` consecutive strings x the whole table.
 *  1. the tile's bytes are staged in LDS and every string gets its 256-bit
 *     gram set;
 *  2. one lane per (string, key length) hashes the string's prefix of that
 *     length and probes the prefix index; a hit names a group of patterns that
 *     all start with exactly those literal bytes;
 *  3. hits are cut into units of up to 256 consecutive patterns; the waves
 *     share the units, four patterns per lane, and keep the patterns whose
 *     gram signature is inside the string's gram set: survivors are compacted
 *     by ballot into one ring per block;
 *  4. survivors are run 64 at a time by as few waves as it takes (a wave with
 *     four live lanes costs as much as one with 64): every lane copies one
 *     survivor's program slot into LDS and runs it on its string; matches go
 *     to per-string min / count cells in LDS, stored at the end.
 * The block owns its strings: no global atomics, no pre-initialised output.
 */
__global__ void __launch_bounds__(K1_THREADS)
k1_match_indexed(const uint8_t *__restrict__ S, const uint32_t *__restrict__ s_off, uint32_t N, uint32_t spb,
                 uint32_t str_cap, uint32_t s_aligned, const k1_table_t T, uint32_t *__restrict__ first_idx,
                 uint32_t *__restrict__ count
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
#else
#define K1_STAMP(i) do { } while (0)
#endif
    K1_STAMP(0);
    /* dynamic LDS, sized by the launch for its strings-per-block:
     * [spb occurrence tables][str_cap/4 + 5 words of string bytes][hit_start][hit_slot][hit_cnt][hit_upref + 1] */
    extern __shared__ __attribute__((aligned(16))) uint32_t k1_lds[];
    __shared__ uint32_t ring[K1_RING], ring_orig[K1_RING];
    __shared__ uint32_t bloom[K1_MAX_SPB][MCX_BLOOM_WORDS];
    __shared__ uint32_t res_first[K1_MAX_SPB], res_count[K1_MAX_SPB];
    __shared__ uint32_t str_at[K1_MAX_SPB], str_len[K1_MAX_SPB];
    __shared__ uint32_t nhits, ring_tail, ring_head;
    const uint32_t max_hits = spb * T.nlens;
    uint32_t *occ_w = k1_lds;
    uint32_t *str_w = occ_w + spb * K1_OCC_WORDS;
    uint32_t *hit_start = str_w + str_cap / 4u + 5u;
    uint32_t *hit_slot = hit_start + max_hits;
    uint32_t *hit_cnt = hit_slot + max_hits;
    uint32_t *hit_upref = hit_cnt + max_hits;

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

    for (uint32_t i = tid; i < nstr; i += K1_THREADS) {
        uint32_t so = s_off[t0 + i];
        str_at[i] = so - gbase + shift;
        str_len[i] = s_off[t0 + i + 1] - so;
        res_first[i] = MCX_NO_MATCH;
        res_count[i] = 0;
    }
    for (uint32_t w = tid; w < nstr * MCX_BLOOM_WORDS; w += K1_THREADS) {
        /* strings read in place get a full set: their patterns are not screened */
        uint32_t v = use_lds ? 0u : 0xFFFFFFFFu;
        if ((w & (MCX_BLOOM_WORDS - 1u)) == (MCX_BLOOM_FREE >> 5))
            v |= 1u << (MCX_BLOOM_FREE & 31u);
        bloom[w / MCX_BLOOM_WORDS][w & (MCX_BLOOM_WORDS - 1u)] = v;
    }
    if (tid == 0) {
        nhits = 0;
        ring_tail = 0;
        ring_head = 0;
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
    K1_STAMP(1);

    /* ---- gram sets: one wave per string, one lane per start position ------- */
    if (use_lds) {
        for (uint32_t slot = wave; slot < nstr; slot += K1_WAVES) {
            const uint32_t len = str_len[slot];
            LdsStr str{(lds_words_t)str_w, str_at[slot]};
            for (uint32_t q = lane; q + 4u <= len; q += 64u) {
                uint32_t bit = mcx_gram_bit(str.word(q));
                atomicOr(&bloom[slot][bit >> 5], 1u << (bit & 31u));
            }
            /* occurrence table, for strings the bit-parallel matcher takes */
            if (len <= MCX_OCC_MAX_LEN && lane < len)
                atomicOr(&occ_w[slot * K1_OCC_WORDS + 2u * str.byte(lane) + (lane >> 5)], 1u << (lane & 31u));
        }
    }

    /* ---- probe: last wave first, so that it overlaps the gram sets -------- */
    const uint32_t nprobe = nstr * T.nlens;
    for (uint32_t pr = (K1_THREADS - 1u) - tid; pr < nprobe; pr += K1_THREADS) {
        const uint32_t slot = pr / T.nlens, L = T.lens[pr - slot * T.nlens];
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
            GlobalStr str{S + gbase + str_at[slot], len};
#pragma unroll
            for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k)
                kw[k] = (4u * k < L) ? str.word(4u * k) : 0u;
        }
        /* zero the bytes past L in the last word */
#pragma unroll
        for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k)
            if ((L >> 2) == k && (L & 3u))
                kw[k] &= 0xFFFFFFFFu >> (8u * (4u - (L & 3u)));
        uint32_t a, b;
        mcx_key_hash(kw, L, a, b);
        for (uint32_t at = a & T.slot_mask;; at = (at + 1u) & T.slot_mask) {
            const mcx_slot_t sl = T.slots[at];
            if (sl.count == 0u)
                break;
            bool same = sl.h1 == a && sl.h2 == b;
#pragma unroll
            for (uint32_t k = 0; k < MCX_KEY_MAX / 4; ++k)
                same = same && sl.key[k] == kw[k];
            if (same) {
                uint32_t h = atomicAdd(&nhits, 1u);
                hit_start[h] = sl.start;
                hit_slot[h] = slot;
                hit_cnt[h] = sl.count;
                break;
            }
        }
    }
    __syncthreads();
    K1_STAMP(2);

    /* ---- exclusive prefix sum of the hits' unit counts (first wave) ------- */
    const uint32_t nh = nhits;
    if (wave == 0) {
        const uint32_t per = (nh + 63u) / 64u;
        const uint32_t lo = lane * per, hi = min(lo + per, nh);
        uint32_t sum = 0;
        for (uint32_t h = lo; h < hi; ++h)
            sum += (hit_cnt[h] + K1_UNIT - 1u) / K1_UNIT;
        uint32_t incl = sum;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d)
                incl += o;
        }
        uint32_t run = incl - sum;
        if (lane == 0)
            hit_upref[0] = 0;
        for (uint32_t h = lo; h < hi; ++h) {
            run += (hit_cnt[h] + K1_UNIT - 1u) / K1_UNIT;
            hit_upref[h + 1] = run;
        }
    }
    __syncthreads();

    /* ---- units -> survivors ----------------------------------------------- */
    K1_STAMP(3);
    const uint32_t nunits = hit_upref[nh];

    /* run the n survivors ring[from .. from+n), one per lane of this wave */
    auto run_survivors = [&](uint32_t from, uint32_t n) {
        const bool have = lane < n;
        const uint32_t e = have ? ring[(from + lane) & (K1_RING - 1u)] : 0u;
        const uint32_t orig = have ? ring_orig[(from + lane) & (K1_RING - 1u)] : 0u;
        const uint32_t k = e & 0x03FFFFFFu, slot = e >> 26;
        K1_STAMP(7);
        if (have) {
            /* the program is read in place: its slot is one 128-byte line */
            const uint32_t len = str_len[slot];
            const GlobalProg lp{T.slot + (size_t)k * MCX_SLOT_WORDS};
            const uint32_t hdr = lp[0];
            const bool is_ext = (MCX_HDR_FLAGS(hdr) & MCX_PF_EXT) != 0u;
            const GlobalProg gp{is_ext ? T.ext : lp.p};
            const uint32_t at = is_ext ? lp[1] : 0u;
            bool ok;
            if (use_lds && len <= MCX_OCC_MAX_LEN) {
                /* the probe compared the key's bytes: skip them */
                LdsOcc occ{(lds_words_t)occ_w + slot * K1_OCC_WORDS};
                LdsStr str{(lds_words_t)str_w, str_at[slot]};
                ok = mcx_prog_match_occ(gp, at, str, occ, len, MCX_HDR_KEYLEN(hdr), T.sets);
            } else if (use_lds) {
                LdsStr str{(lds_words_t)str_w, str_at[slot]};
                ok = mcx_prog_match(gp, at, str, len, T.sets);
            } else {
                GlobalStr str{S + gbase + str_at[slot], len};
                ok = mcx_prog_match(gp, at, str, len, T.sets);
            }
            K1_STAMP(8);
            if (ok) {
                atomicMin(&res_first[slot], orig);
                atomicAdd(&res_count[slot], 1u);
            }
        }
        K1_STAMP(9);
        __builtin_amdgcn_wave_barrier();
    };

    /* the block walks the units in rounds of K1_WAVES (one unit per wave);
     * after every round the survivors beyond a multiple of 64 stay in the
     * ring, the others are run */
    const uint32_t nrounds = (nunits + K1_WAVES - 1u) / K1_WAVES;
    uint32_t h = 0;
    for (uint32_t r = 0; r < nrounds; ++r) {
        const uint32_t u = r * K1_WAVES + wave;
        if (u < nunits) {
            while (hit_upref[h + 1] <= u)
                ++h;
            const uint32_t j = u - hit_upref[h];
            const uint32_t k0 = hit_start[h] + K1_UNIT * j;
            const uint32_t n = min(K1_UNIT, hit_cnt[h] - K1_UNIT * j);
            const uint32_t slot = hit_slot[h];
            k1_sig4_t cur;
#pragma unroll
            for (uint32_t jj = 0; jj < 4; ++jj)
                cur.v[jj] = make_uint2(0u, 0u);
            if (4u * lane + 4u <= n) {
                cur = *(const k1_sig4_t *)(T.sigorig + k0 + 4u * lane);
            } else {
#pragma unroll
                for (uint32_t jj = 0; jj < 4; ++jj)
                    if (4u * lane + jj < n)
                        cur.v[jj] = T.sigorig[k0 + 4u * lane + jj];
            }
            LdsBloom bl{(lds_words_t)bloom[slot]};
            bool pass[4];
            uint32_t npass = 0;
#pragma unroll
            for (uint32_t jj = 0; jj < 4; ++jj) {
                pass[jj] = (4u * lane + jj < n) && mcx_sig_pass(cur.v[jj].x, bl);
                npass += pass[jj] ? 1u : 0u;
            }
            /* wave-wide exclusive scan of npass, one reservation per wave */
            uint32_t incl = npass;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d)
                    incl += o;
            }
            const uint32_t wtotal = __shfl(incl, 63, 64);
            if (wtotal) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&ring_tail, wtotal);
                base = __shfl(base, 0, 64);
                uint32_t at = base + incl - npass;
#pragma unroll
                for (uint32_t jj = 0; jj < 4; ++jj)
                    if (pass[jj]) {
                        ring[at & (K1_RING - 1u)] = (slot << 26) | (k0 + 4u * lane + jj);
                        ring_orig[at & (K1_RING - 1u)] = cur.v[jj].y;
                        ++at;
                    }
            }
        }
        __syncthreads();
        {
            /* full groups of 64 go now; at most K1_WAVES x 256 new survivors a
             * round plus 63 left over fit the ring */
            const uint32_t head = ring_head, avail = ring_tail - head;
            const uint32_t groups = avail / 64u;
            for (uint32_t g = wave; g < groups; g += K1_WAVES)
                run_survivors(head + 64u * g, 64u);
            __syncthreads();
            if (tid == 0)
                ring_head = head + 64u * groups;
            __syncthreads();
        }
    }
    K1_STAMP(4);
    {
        /* fewer than 64 are left: split them over the waves, a wave's time
         * follows its slowest lane, not its lane count */
        const uint32_t head = ring_head, avail = ring_tail - head;
        const uint32_t share = (avail + K1_WAVES - 1u) / K1_WAVES;
        const uint32_t from = wave * share;
        if (from < avail)
            run_survivors(head + from, min(share, avail - from));
    }
    __syncthreads();
    K1_STAMP(5);
    for (uint32_t i = tid; i < nstr; i += K1_THREADS) {
        first_idx[t0 + i] = res_first[i];
        count[t0 + i] = res_count[i];
    }
    K1_STAMP(6);
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
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    mcx_compiled_table t;
    std::string err;
    int rc = mcx_compile_table(arena, a_off, P, (flags & MCX_TABLE_NORMALIZE_DASH) != 0, t, err);
    if (rc != 0) {
        ctx->err = err;
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    mcx_free_table(ctx);
    if ((rc = mcx_dev_upload(ctx, &ctx->d_slot, t.slot.data(), t.slot.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_ext, t.ext.data(), t.ext.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_sigorig, t.sigorig.data(), t.sigorig.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_slots, t.slots.data(), t.slots.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_sets, t.sets.data(), t.sets.size())) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_amod, a_mod, (size_t)P)) != 0 ||
        (rc = mcx_dev_upload(ctx, &ctx->d_hwtag, t.hw_tag.data(), t.hw_tag.size())) != 0) {
        mcx_free_table(ctx);
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->P = P;
    ctx->n_indexed = (uint32_t)(t.sigorig.size() / 2);
    ctx->slot_mask = (uint32_t)t.slots.size() - 1u;
    ctx->nlens = (uint32_t)t.lens.size();
    for (uint32_t k = 0; k < ctx->nlens; ++k)
        ctx->lens[k] = t.lens[k];
    ctx->norm_dash = (flags & MCX_TABLE_NORMALIZE_DASH) != 0;
    ctx->table_bytes = 4 * (uint64_t)(t.slot.size() + t.ext.size() + t.sigorig.size() + t.sets.size()) +
                       sizeof(mcx_slot_t) * (uint64_t)t.slots.size() + 4 * (uint64_t)P + t.hw_tag.size();
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

static int launch_alias_match(mcx_ctx *ctx, const uint8_t *d_S, const uint32_t *d_s_off, uint32_t N,
                              uint32_t *d_first, uint32_t *d_count)
{
    if (N == 0)
        return MCX_OK;
    /* few strings: one per block, so that the whole chip has work;
     * many strings: up to 64 per block */
    uint32_t spb = (N + 2047u) / 2048u;
    if (spb > K1_MAX_SPB)
        spb = K1_MAX_SPB;
    uint32_t blocks = (N + spb - 1u) / spb;
    k1_table_t T;
    T.slot = ctx->d_slot;
    T.ext = ctx->d_ext;
    T.sigorig = (const uint2 *)ctx->d_sigorig;
    T.slots = ctx->d_slots;
    T.sets = ctx->d_sets;
    T.slot_mask = ctx->slot_mask;
    T.nlens = ctx->nlens;
    for (uint32_t k = 0; k <= MCX_KEY_MAX; ++k)
        T.lens[k] = k < ctx->nlens ? ctx->lens[k] : 0u;
    /* staging buffer: 256 bytes per string of the tile, between 1 KiB and 16 KiB;
     * a tile that needs more is read in place */
    uint32_t str_cap = spb * 256u;
    if (str_cap < 1024u)
        str_cap = 1024u;
    if (str_cap > K1_LDS_STR_BYTES)
        str_cap = K1_LDS_STR_BYTES;
    size_t lds = 4 * ((size_t)spb * K1_OCC_WORDS + (size_t)str_cap / 4 + 5 + 4 * (size_t)spb * T.nlens + 1);
    const uint32_t s_aligned = ((uintptr_t)d_S & 3u) == 0u ? 1u : 0u;
#ifdef MCX_K1_STAMPS
    if (!g_k1_stamps)
        (void)hipMalloc((void **)&g_k1_stamps, 16u * 8u * 65536u);
    g_k1_stamp_blocks = blocks < 65536u ? blocks : 65536u;
    k1_match_indexed<<<blocks, K1_THREADS, lds, ctx->stream>>>(d_S, d_s_off, N, spb, str_cap, s_aligned, T,
                                                               d_first, d_count, g_k1_stamps);
#else
    k1_match_indexed<<<blocks, K1_THREADS, lds, ctx->stream>>>(d_S, d_s_off, N, spb, str_cap, s_aligned, T,
                                                               d_first, d_count);
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
    if (ctx->norm_dash) {
        ctx->err = "MCX_TABLE_NORMALIZE_DASH needs the host-buffer entry point";
        return MCX_E_BAD_ARG;
    }
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
    std::vector<char> norm;
    const char *src = S;
    if (ctx->norm_dash) {
        /* kmod applies the '-' -> '_' rule to the queried alias as well; a
         * query it refuses (unmatched bracket) resolves to nothing, so the
         * results of such strings are cleared after the match. */
        norm.assign(S, S + s_off[N]);
        src = norm.data();
    }
    std::vector<uint8_t> refused;
    if (ctx->norm_dash) {
        refused.assign(N, 0);
        for (uint32_t i = 0; i < N; ++i) {
            char *p = norm.data();
            for (uint32_t k = s_off[i]; k < s_off[i + 1]; ++k) {
                if (p[k] == '-') {
                    p[k] = '_';
                } else if (p[k] == ']') {
                    refused[i] = 1;
                    break;
                } else if (p[k] == '[') {
                    while (k < s_off[i + 1] && p[k] != ']')
                        ++k;
                    if (k >= s_off[i + 1]) {
                        refused[i] = 1;
                        break;
                    }
                }
            }
        }
    }
    uint8_t *d_S = nullptr;
    uint32_t *d_off = nullptr, *d_first = nullptr, *d_count = nullptr;
    int rc = MCX_OK;
    hipError_t e;
    size_t sbytes = s_off[N];
    if ((e = hipMalloc((void **)&d_S, sbytes ? sbytes : 1)) != hipSuccess ||
        (e = hipMalloc((void **)&d_off, 4 * ((size_t)N + 1))) != hipSuccess ||
        (e = hipMalloc((void **)&d_first, 4 * (size_t)N)) != hipSuccess ||
        (e = hipMalloc((void **)&d_count, 4 * (size_t)N)) != hipSuccess) {
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        rc = MCX_E_NOMEM;
    }
    if (rc == MCX_OK && sbytes &&
        (e = hipMemcpyAsync(d_S, src, sbytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = MCX_E_HIP;
    if (rc == MCX_OK &&
        (e = hipMemcpyAsync(d_off, s_off, 4 * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = MCX_E_HIP;
    if (rc == MCX_OK)
        rc = launch_alias_match(ctx, d_S, d_off, N, d_first, d_count);
    if (rc == MCX_OK &&
        ((e = hipMemcpyAsync(first_idx, d_first, 4 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
         (e = hipMemcpyAsync(count, d_count, 4 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
         (e = hipStreamSynchronize(ctx->stream)) != hipSuccess)) {
        ctx->err = std::string("copy back: ") + hipGetErrorString(e);
        rc = MCX_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_S);
    (void)hipFree(d_off);
    (void)hipFree(d_first);
    (void)hipFree(d_count);
    if (rc == MCX_OK && ctx->norm_dash)
        for (uint32_t i = 0; i < N; ++i)
            if (refused[i]) {
                first_idx[i] = MCX_NO_MATCH;
                count[i] = 0;
            }
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
