/*
 * mcx_device.hip - HIP kernels (gfx950) and the C ABI of include/mcx.h.
 *
 * K1  alias_match : device modalias strings x alias patterns, fnmatch flags 0.
 * K2a dep_flags   : has-holders / in-use / root bits over the dependency CSR.
 * K2b dep_closure : frontier BFS over the (reverse) dependency CSR.
 *
 * There is no CPU path in this file: every entry point needs a HIP device.
 */
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/mcx.h"
#include "glob_compile.h"
#include "mcx_prog.h"

/* ------------------------------------------------------------------------- */
/* context                                                                    */
/* ------------------------------------------------------------------------- */

struct mcx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    /* alias table */
    uint32_t P = 0;
    bool table_loaded = false;
    bool norm_dash = false;
    uint8_t *d_prog = nullptr;
    uint32_t *d_prog_off = nullptr;
    mcx_prefix_t *d_prefix = nullptr;
    uint32_t *d_meta = nullptr;
    uint32_t *d_sets = nullptr;
    uint32_t *d_amod = nullptr;
    uint8_t *d_hwtag = nullptr;
    uint64_t table_bytes = 0;

    /* dependency graph: forward CSR as given, reverse CSR built at load */
    uint32_t V = 0, E = 0;
    bool graph_loaded = false;
    uint32_t *d_rowptr = nullptr, *d_col = nullptr;
    uint32_t *d_rrowptr = nullptr, *d_rcol = nullptr;
    uint32_t *d_holders = nullptr; /* V counters                              */
    uint32_t *d_queue[2] = {nullptr, nullptr};
    uint32_t *d_counters = nullptr; /* [0],[1] queue sizes, [2..3] edges (u64) */
};

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return MCX_E_HIP;                                                          \
        }                                                                              \
    } while (0)

static int set_device(mcx_ctx *ctx)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return 0;
}

template <class T>
static int dev_upload(mcx_ctx *ctx, T **d_out, const T *src, size_t count)
{
    size_t bytes = sizeof(T) * (count ? count : 1);
    HIPCHK(ctx, hipMalloc((void **)d_out, bytes));
    if (count)
        HIPCHK(ctx, hipMemcpyAsync(*d_out, src, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

/* ------------------------------------------------------------------------- */
/* K1: alias match                                                            */
/* ------------------------------------------------------------------------- */

#define K1_THREADS 256
#define K1_WAVES 4
#define K1_TILE_STR 64
#define K1_LDS_BYTES 16384

typedef const __attribute__((address_space(3))) uint8_t *lds_bytes_t;
struct LdsStr {
    lds_bytes_t p;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[i]; }
};
struct GlobalStr {
    const uint8_t *p;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[i]; }
};

__global__ void __launch_bounds__(256) k1_init(uint32_t *__restrict__ first, uint32_t *__restrict__ count, uint32_t N)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        first[i] = MCX_NO_MATCH;
        count[i] = 0;
    }
}

/*
 * One block = 64 strings (one per lane, the same 64 in each of the 4 waves)
 * x one tile of patterns, a quarter of the tile per wave.  The pattern index
 * is wave-uniform: prefilter records, meta words and program offsets come in
 * through scalar loads and the prefilter compares a lane's first 16 string
 * bytes against SGPR operands.  Lanes that pass run the segment program.
 */
__global__ void __launch_bounds__(K1_THREADS)
k1_match_tiled(const uint8_t *__restrict__ S, const uint32_t *__restrict__ s_off, uint32_t N,
               const mcx_prefix_t *__restrict__ prefix, const uint32_t *__restrict__ meta,
               const uint32_t *__restrict__ prog_off, const uint8_t *__restrict__ prog,
               const uint32_t *__restrict__ sets, uint32_t P, uint32_t pat_tile,
               uint32_t *__restrict__ first_idx, uint32_t *__restrict__ count)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[K1_LDS_BYTES];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t t0 = blockIdx.y * K1_TILE_STR;
    const uint32_t nstr = min((uint32_t)K1_TILE_STR, N - t0);
    const uint32_t base = s_off[t0];
    /* a tile whose strings do not fit the staging buffer is read in place */
    const uint32_t total = s_off[t0 + nstr] - base;
    const bool USE_LDS = total <= K1_LDS_BYTES;

    if (USE_LDS) {
        for (uint32_t b = threadIdx.x; b < total; b += K1_THREADS)
            lds[b] = S[base + b];
        __syncthreads();
    }

    const bool live = lane < nstr;
    const uint32_t i = t0 + (live ? lane : 0u);
    const uint32_t so = s_off[i];
    const uint32_t len = live ? s_off[i + 1] - so : 0u;
    const LdsStr str_lds{(lds_bytes_t)lds + (so - base)};
    const GlobalStr str_glb{S + so};

    /* first 16 bytes, zero padded: no string byte is NUL, so padding can never
     * equal a literal */
    uint32_t h[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        uint32_t c = 0u;
        if (k < len)
            c = USE_LDS ? str_lds[k] : str_glb[k];
        h[k >> 2] |= c << ((k & 3u) * 8u);
    }

    const uint32_t j_tile = blockIdx.x * pat_tile;
    const uint32_t per_wave = (pat_tile + K1_WAVES - 1) / K1_WAVES;
    const uint32_t j_begin = j_tile + wave * per_wave;
    const uint32_t j_end = min(min(j_begin + per_wave, j_tile + pat_tile), P);

    uint32_t first = MCX_NO_MATCH, cnt = 0;
    for (uint32_t j = j_begin; j < j_end; ++j) {
        const mcx_prefix_t pf = prefix[j];
        const uint32_t m = meta[j];
        uint32_t d = ((h[0] ^ pf.pfx[0]) & pf.mask[0]) | ((h[1] ^ pf.pfx[1]) & pf.mask[1]) |
                     ((h[2] ^ pf.pfx[2]) & pf.mask[2]) | ((h[3] ^ pf.pfx[3]) & pf.mask[3]);
        bool pass = live && d == 0u && len >= MCX_META_MINLEN(m);
        if (pass) {
            bool ok;
            if (USE_LDS)
                ok = mcx_prog_match(prog + prog_off[j], str_lds, len, sets);
            else
                ok = mcx_prog_match(prog + prog_off[j], str_glb, len, sets);
            if (ok) {
                first = min(first, j);
                ++cnt;
            }
        }
    }
    if (cnt) {
        atomicMin(&first_idx[i], first);
        atomicAdd(&count[i], cnt);
    }
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

/* ------------------------------------------------------------------------- */
/* K2a: dependency flags                                                      */
/* ------------------------------------------------------------------------- */

__global__ void __launch_bounds__(256)
k2_count_holders(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                 const uint8_t *__restrict__ in_flags, uint32_t V, uint32_t *__restrict__ holders)
{
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= V || !(in_flags[u] & MCX_F_LOADED))
        return;
    uint32_t k0 = rowptr[u], k1 = rowptr[u + 1];
    for (uint32_t k = k0; k < k1; ++k) {
        uint32_t v = col[k];
        if (in_flags[v] & MCX_F_LOADED)
            atomicAdd(&holders[v], 1u);
    }
}

__global__ void __launch_bounds__(256)
k2_flags(const uint8_t *__restrict__ in_flags, const uint32_t *__restrict__ holders, uint32_t V,
         uint8_t *__restrict__ out_flags, uint32_t *__restrict__ refcnt)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V)
        return;
    uint32_t f = in_flags[v], hcount = holders[v], o = 0;
    if (f & MCX_F_LOADED) {
        if (hcount)
            o |= MCX_O_HAS_HOLDERS;
        if ((f & MCX_F_BOUND) || hcount)
            o |= MCX_O_IN_USE;
        if ((f & MCX_F_SAFE) && !hcount)
            o |= MCX_O_ROOT;
    }
    out_flags[v] = (uint8_t)o;
    if (refcnt)
        refcnt[v] = (f & MCX_F_LOADED) ? hcount + ((f & MCX_F_BOUND) ? 1u : 0u) : 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------- */
/* K2b: closure (level-synchronous frontier BFS)                              */
/* ------------------------------------------------------------------------- */

__global__ void __launch_bounds__(256)
k2b_seed(const uint32_t *__restrict__ seed, const uint8_t *__restrict__ in_flags, uint32_t V,
         uint32_t *__restrict__ out, uint32_t *__restrict__ queue, uint32_t *__restrict__ qsize)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V)
        return;
    if ((seed[v >> 5] >> (v & 31u) & 1u) && (in_flags[v] & MCX_F_LOADED)) {
        atomicOr(&out[v >> 5], 1u << (v & 31u));
        queue[atomicAdd(qsize, 1u)] = v;
    }
}

__global__ void __launch_bounds__(256)
k2b_expand(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
           const uint8_t *__restrict__ in_flags, const uint32_t *__restrict__ frontier,
           const uint32_t *__restrict__ fsize, uint32_t *__restrict__ out, uint32_t *__restrict__ next,
           uint32_t *__restrict__ nsize, unsigned long long *__restrict__ edges)
{
    uint32_t n = *fsize;
    unsigned long long scanned = 0;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        uint32_t v = frontier[t];
        uint32_t k0 = rowptr[v], k1 = rowptr[v + 1];
        scanned += k1 - k0;
        for (uint32_t k = k0; k < k1; ++k) {
            uint32_t u = col[k];
            if (!(in_flags[u] & MCX_F_LOADED))
                continue;
            uint32_t bit = 1u << (u & 31u);
            if (out[u >> 5] & bit)
                continue;
            uint32_t old = atomicOr(&out[u >> 5], bit);
            if (!(old & bit))
                next[atomicAdd(nsize, 1u)] = u;
        }
    }
    if (scanned)
        atomicAdd(edges, scanned);
}

/* ------------------------------------------------------------------------- */
/* C ABI                                                                      */
/* ------------------------------------------------------------------------- */

extern "C" {

const char *mcx_strerror(int rc)
{
    switch (rc) {
    case MCX_OK: return "ok";
    case MCX_E_NO_DEVICE: return "no usable HIP device";
    case MCX_E_BAD_ARG: return "bad argument";
    case MCX_E_HIP: return "HIP runtime error";
    case MCX_E_NOMEM: return "out of memory";
    case MCX_E_PATTERN: return "pattern outside the supported grammar";
    case MCX_E_TOO_LARGE: return "size beyond a documented limit";
    case MCX_E_NOT_LOADED: return "table or graph not loaded";
    default: return "unknown error";
    }
}

const char *mcx_last_error(mcx_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : "";
}

int mcx_ctx_create(int device_ordinal, void *stream, mcx_ctx **out)
{
    if (!out)
        return MCX_E_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return MCX_E_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return MCX_E_BAD_ARG;
    if (hipSetDevice(device_ordinal) != hipSuccess)
        return MCX_E_NO_DEVICE;
    mcx_ctx *ctx = new (std::nothrow) mcx_ctx();
    if (!ctx)
        return MCX_E_NOMEM;
    ctx->device = device_ordinal;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return MCX_E_NO_DEVICE;
        }
        ctx->own_stream = true;
    }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc((void **)&ctx->d_counters, 64) != hipSuccess) {
        mcx_ctx_destroy(ctx);
        return MCX_E_NO_DEVICE;
    }
    *out = ctx;
    return MCX_OK;
}

static void free_table(mcx_ctx *ctx)
{
    (void)hipFree(ctx->d_prog);
    (void)hipFree(ctx->d_prog_off);
    (void)hipFree(ctx->d_prefix);
    (void)hipFree(ctx->d_meta);
    (void)hipFree(ctx->d_sets);
    (void)hipFree(ctx->d_amod);
    (void)hipFree(ctx->d_hwtag);
    ctx->d_prog = nullptr;
    ctx->d_prog_off = nullptr;
    ctx->d_prefix = nullptr;
    ctx->d_meta = nullptr;
    ctx->d_sets = nullptr;
    ctx->d_amod = nullptr;
    ctx->d_hwtag = nullptr;
    ctx->table_loaded = false;
    ctx->P = 0;
    ctx->table_bytes = 0;
}

static void free_graph(mcx_ctx *ctx)
{
    (void)hipFree(ctx->d_rowptr);
    (void)hipFree(ctx->d_col);
    (void)hipFree(ctx->d_rrowptr);
    (void)hipFree(ctx->d_rcol);
    (void)hipFree(ctx->d_holders);
    (void)hipFree(ctx->d_queue[0]);
    (void)hipFree(ctx->d_queue[1]);
    ctx->d_rowptr = ctx->d_col = ctx->d_rrowptr = ctx->d_rcol = ctx->d_holders = nullptr;
    ctx->d_queue[0] = ctx->d_queue[1] = nullptr;
    ctx->graph_loaded = false;
    ctx->V = ctx->E = 0;
}

void mcx_ctx_destroy(mcx_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    free_table(ctx);
    free_graph(ctx);
    (void)hipFree(ctx->d_counters);
    if (ctx->ev0)
        (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1)
        (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mcx_sync(mcx_ctx *ctx)
{
    if (!ctx)
        return MCX_E_BAD_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MCX_OK;
}

void *mcx_ctx_stream(mcx_ctx *ctx)
{
    return ctx ? (void *)ctx->stream : nullptr;
}

int mcx_dev_alloc(mcx_ctx *ctx, size_t bytes, void **d_out)
{
    if (!ctx || !d_out)
        return MCX_E_BAD_ARG;
    if (set_device(ctx))
        return MCX_E_HIP;
    HIPCHK(ctx, hipMalloc(d_out, bytes ? bytes : 1));
    return MCX_OK;
}

int mcx_dev_free(mcx_ctx *ctx, void *d_ptr)
{
    if (!ctx)
        return MCX_E_BAD_ARG;
    HIPCHK(ctx, hipFree(d_ptr));
    return MCX_OK;
}

int mcx_h2d(mcx_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !src)))
        return MCX_E_BAD_ARG;
    HIPCHK(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MCX_OK;
}

int mcx_d2h(mcx_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !d_src)))
        return MCX_E_BAD_ARG;
    HIPCHK(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MCX_OK;
}

int mcx_timer_begin(mcx_ctx *ctx)
{
    if (!ctx)
        return MCX_E_BAD_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return MCX_OK;
}

int mcx_timer_end(mcx_ctx *ctx, float *ms_out)
{
    if (!ctx || !ms_out)
        return MCX_E_BAD_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
    return MCX_OK;
}

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
    if (set_device(ctx))
        return MCX_E_HIP;
    mcx_compiled_table t;
    std::string err;
    int rc = mcx_compile_table(arena, a_off, P, (flags & MCX_TABLE_NORMALIZE_DASH) != 0, t, err);
    if (rc != 0) {
        ctx->err = err;
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_table(ctx);
    if ((rc = dev_upload(ctx, &ctx->d_prog, t.prog.data(), t.prog.size())) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_prog_off, t.prog_off.data(), t.prog_off.size())) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_prefix, t.prefix.data(), t.prefix.size())) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_meta, t.meta.data(), t.meta.size())) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_sets, t.sets.data(), t.sets.size())) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_amod, a_mod, (size_t)P)) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_hwtag, t.hw_tag.data(), t.hw_tag.size())) != 0) {
        free_table(ctx);
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->P = P;
    ctx->norm_dash = (flags & MCX_TABLE_NORMALIZE_DASH) != 0;
    ctx->table_bytes = t.prog.size() + 4 * t.prog_off.size() + sizeof(mcx_prefix_t) * t.prefix.size() +
                       4 * t.meta.size() + 4 * t.sets.size() + 4 * (uint64_t)P + t.hw_tag.size();
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

static int launch_alias_match(mcx_ctx *ctx, const uint8_t *d_S, const uint32_t *d_s_off, uint32_t N,
                              uint32_t *d_first, uint32_t *d_count)
{
    if (N == 0)
        return MCX_OK;
    k1_init<<<(N + 255) / 256, 256, 0, ctx->stream>>>(d_first, d_count, N);
    if (ctx->P == 0)
        return MCX_OK;
    uint32_t tiles_s = (N + K1_TILE_STR - 1) / K1_TILE_STR;
    uint32_t want_p = tiles_s >= 4096 ? 1u : 4096u / tiles_s;
    uint32_t pat_tile = (ctx->P + want_p - 1) / want_p;
    if (pat_tile < 64)
        pat_tile = 64;
    pat_tile = (pat_tile + 3u) & ~3u;
    uint32_t tiles_p = (ctx->P + pat_tile - 1) / pat_tile;
    if (tiles_s > 65535u) {
        ctx->err = "more than 65535*64 strings in one call";
        return MCX_E_TOO_LARGE;
    }
    dim3 grid(tiles_p, tiles_s);
    k1_match_tiled<<<grid, K1_THREADS, 0, ctx->stream>>>(d_S, d_s_off, N, ctx->d_prefix, ctx->d_meta,
                                                          ctx->d_prog_off, ctx->d_prog, ctx->d_sets, ctx->P,
                                                          pat_tile, d_first, d_count);
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
    if (set_device(ctx))
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
    if (set_device(ctx))
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
    if (set_device(ctx))
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
    if (set_device(ctx))
        return MCX_E_HIP;
    uint8_t *d_hw = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_hw, n_modules));
    (void)hipMemsetAsync(d_hw, 0, n_modules, ctx->stream);
    if (ctx->P)
        k1_hw_flags<<<(ctx->P + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_hwtag, ctx->d_amod, ctx->P, n_modules,
                                                               d_hw);
    hipError_t e = hipMemcpyAsync(hw_alias, d_hw, n_modules, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_hw);
    if (e != hipSuccess) {
        ctx->err = hipGetErrorString(e);
        return MCX_E_HIP;
    }
    return MCX_OK;
}

/* ---- K2 ----------------------------------------------------------------- */

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
    if (set_device(ctx))
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_graph(ctx);
    int rc;
    if ((rc = dev_upload(ctx, &ctx->d_rowptr, rowptr, (size_t)V + 1)) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_col, col, (size_t)E)) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_rrowptr, rrow.data(), (size_t)V + 1)) != 0 ||
        (rc = dev_upload(ctx, &ctx->d_rcol, rcol.data(), (size_t)E)) != 0) {
        free_graph(ctx);
        return rc;
    }
    size_t vb = 4 * ((size_t)V ? V : 1);
    if (hipMalloc((void **)&ctx->d_holders, vb) != hipSuccess ||
        hipMalloc((void **)&ctx->d_queue[0], vb) != hipSuccess ||
        hipMalloc((void **)&ctx->d_queue[1], vb) != hipSuccess) {
        free_graph(ctx);
        ctx->err = "hipMalloc of graph scratch failed";
        return MCX_E_NOMEM;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->V = V;
    ctx->E = E;
    ctx->graph_loaded = true;
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
    if (set_device(ctx))
        return MCX_E_HIP;
    uint32_t V = ctx->V, blocks = (V + 255) / 256;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_holders, 0, 4 * (size_t)V, ctx->stream));
    k2_count_holders<<<blocks, 256, 0, ctx->stream>>>(ctx->d_rowptr, ctx->d_col, (const uint8_t *)d_in_flags, V,
                                                     ctx->d_holders);
    k2_flags<<<blocks, 256, 0, ctx->stream>>>((const uint8_t *)d_in_flags, ctx->d_holders, V,
                                             (uint8_t *)d_out_flags, (uint32_t *)d_refcnt);
    HIPCHK(ctx, hipGetLastError());
    return MCX_OK;
}

int mcx_dep_flags(mcx_ctx *ctx, const uint8_t *in_flags, uint8_t *out_flags, uint32_t *refcnt)
{
    if (!ctx || !in_flags || !out_flags)
        return MCX_E_BAD_ARG;
    if (!ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    uint32_t V = ctx->V;
    if (V == 0)
        return MCX_OK;
    if (set_device(ctx))
        return MCX_E_HIP;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    uint32_t *d_ref = nullptr;
    int rc = MCX_OK;
    if (hipMalloc((void **)&d_in, V) != hipSuccess || hipMalloc((void **)&d_out, V) != hipSuccess ||
        (refcnt && hipMalloc((void **)&d_ref, 4 * (size_t)V) != hipSuccess)) {
        ctx->err = "hipMalloc failed";
        rc = MCX_E_NOMEM;
    }
    if (rc == MCX_OK && hipMemcpyAsync(d_in, in_flags, V, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
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
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_ref);
    return rc;
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
    uint32_t V = ctx->V;
    if (V == 0)
        return MCX_OK;
    if (set_device(ctx))
        return MCX_E_HIP;
    const uint32_t *rp = direction == MCX_CLOSURE_HOLDERS ? ctx->d_rrowptr : ctx->d_rowptr;
    const uint32_t *cl = direction == MCX_CLOSURE_HOLDERS ? ctx->d_rcol : ctx->d_col;
    uint32_t W = (V + 31) / 32;
    uint32_t *cnt = ctx->d_counters; /* [0] |queue0|, [1] |queue1|, [2..3] edges */
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, 16, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_out_bitmap, 0, 4 * (size_t)W, ctx->stream));
    k2b_seed<<<(V + 255) / 256, 256, 0, ctx->stream>>>((const uint32_t *)d_seed_bitmap,
                                                      (const uint8_t *)d_in_flags, V, (uint32_t *)d_out_bitmap,
                                                      ctx->d_queue[0], &cnt[0]);
    uint32_t lv = 0, cur = 0;
    for (;;) {
        uint32_t fsize = 0;
        HIPCHK(ctx, hipMemcpyAsync(&fsize, &cnt[cur], 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (fsize == 0)
            break;
        ++lv;
        HIPCHK(ctx, hipMemsetAsync(&cnt[cur ^ 1], 0, 4, ctx->stream));
        uint32_t blocks = (fsize + 255) / 256;
        if (blocks > 4096)
            blocks = 4096;
        k2b_expand<<<blocks, 256, 0, ctx->stream>>>(rp, cl, (const uint8_t *)d_in_flags, ctx->d_queue[cur],
                                                   &cnt[cur], (uint32_t *)d_out_bitmap, ctx->d_queue[cur ^ 1],
                                                   &cnt[cur ^ 1], (unsigned long long *)&cnt[2]);
        cur ^= 1;
    }
    HIPCHK(ctx, hipGetLastError());
    if (edges_scanned) {
        unsigned long long e = 0;
        HIPCHK(ctx, hipMemcpyAsync(&e, &cnt[2], 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *edges_scanned = e;
    }
    if (levels)
        *levels = lv;
    return MCX_OK;
}

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
    if (set_device(ctx))
        return MCX_E_HIP;
    size_t wb = 4 * (size_t)((V + 31) / 32);
    uint8_t *d_in = nullptr;
    uint32_t *d_seed = nullptr, *d_out = nullptr;
    int rc = MCX_OK;
    if (hipMalloc((void **)&d_in, V) != hipSuccess || hipMalloc((void **)&d_seed, wb) != hipSuccess ||
        hipMalloc((void **)&d_out, wb) != hipSuccess) {
        ctx->err = "hipMalloc failed";
        rc = MCX_E_NOMEM;
    }
    if (rc == MCX_OK && (hipMemcpyAsync(d_in, in_flags, V, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                         hipMemcpyAsync(d_seed, seed_bitmap, wb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess))
        rc = MCX_E_HIP;
    if (rc == MCX_OK)
        rc = mcx_dep_closure_dev(ctx, d_in, d_seed, d_out, direction, edges_scanned, levels);
    if (rc == MCX_OK && hipMemcpyAsync(out_bitmap, d_out, wb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = MCX_E_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MCX_OK)
        rc = MCX_E_HIP;
    (void)hipFree(d_in);
    (void)hipFree(d_seed);
    (void)hipFree(d_out);
    return rc;
}

} /* extern "C" */
