/*
 * mcx_core.hip - context, error strings, device-memory and timing helpers of
 * the C ABI in include/mcx.h.
 */
#include <new>
#include <string.h>

#include "mcx_internal.h"

/* ------------------------------------------------------------------------- */
/* C ABI                                                                      */
/* ------------------------------------------------------------------------- */

/* release everything a loaded alias table / dependency graph holds */
void mcx_free_table(mcx_ctx *ctx)
{
    (void)hipFree(ctx->d_slot);
    (void)hipFree(ctx->d_ext);
    (void)hipFree(ctx->d_sig);
    (void)hipFree(ctx->d_lens);
    (void)hipFree(ctx->d_slots);
    (void)hipFree(ctx->d_tags);
    (void)hipFree(ctx->d_sub);
    (void)hipFree(ctx->d_sub_tags);
    (void)hipFree(ctx->d_sets);
    (void)hipFree(ctx->d_amod);
    (void)hipFree(ctx->d_hwtag);
    (void)hipFree(ctx->d_dup_start);
    (void)hipFree(ctx->d_dup_idx);
    ctx->d_dup_start = ctx->d_dup_idx = nullptr;
    ctx->d_slot = nullptr;
    ctx->d_ext = nullptr;
    ctx->d_sig = nullptr;
    ctx->d_lens = nullptr;
    ctx->d_slots = nullptr;
    ctx->d_tags = nullptr;
    ctx->d_sub = nullptr;
    ctx->d_sub_tags = nullptr;
    ctx->d_sets = nullptr;
    ctx->d_amod = nullptr;
    ctx->d_hwtag = nullptr;
    ctx->table_loaded = false;
    ctx->P = 0;
    ctx->n_indexed = 0;
    ctx->nlens = 0;
    ctx->table_bytes = 0;
}

void mcx_free_graph(mcx_ctx *ctx)
{
    (void)hipFree(ctx->d_rowptr);
    (void)hipFree(ctx->d_col);
    (void)hipFree(ctx->d_rrowptr);
    (void)hipFree(ctx->d_rcol);
    (void)hipFree(ctx->d_part);
    (void)hipFree(ctx->d_slices);
    (void)hipFree(ctx->d_rpk);
    ctx->d_slices = nullptr;
    ctx->d_rpk = nullptr;
    ctx->n_slices = 0;
    (void)hipFree(ctx->d_queue[0]);
    (void)hipFree(ctx->d_queue[1]);
    (void)hipFree(ctx->d_bfs);
    (void)hipFree(ctx->d_lvl_order);
    ctx->d_lvl_order = nullptr;
    ctx->lvl_ptr.clear();
    ctx->lvl_maxdeg.clear();
    ctx->d_rowptr = ctx->d_col = ctx->d_rrowptr = ctx->d_rcol = ctx->d_part = ctx->d_bfs = nullptr;
    ctx->d_queue[0] = ctx->d_queue[1] = nullptr;
    ctx->n_part = 0;
    ctx->graph_loaded = false;
    ctx->V = ctx->E = 0;
}

int mcx_pinned_io(mcx_ctx *ctx, uint8_t **host, uint8_t **dev)
{
    if (!ctx->h_io) {
        if (hipHostMalloc(&ctx->h_io, MCX_ZERO_COPY_MAX, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
            ctx->h_io = nullptr;
            ctx->err = "hipHostMalloc of the pinned staging area failed";
            return MCX_E_NOMEM;
        }
        ctx->h_io_bytes = MCX_ZERO_COPY_MAX;
        HIPCHK(ctx, hipHostGetDevicePointer(&ctx->h_io_dev, ctx->h_io, 0));
    }
    *host = (uint8_t *)ctx->h_io;
    *dev = (uint8_t *)ctx->h_io_dev;
    return MCX_OK;
}

/* one wave: an optional copy of a few words into pinned memory, then the sequence word, behind a
 * system-scope fence */
__global__ void mcx_done(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t copy_words,
                         uint32_t *__restrict__ seq_word, uint32_t seq)
{
    for (uint32_t i = threadIdx.x; i < copy_words; i += blockDim.x)
        dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int mcx_signal_and_wait(mcx_ctx *ctx, const uint32_t *src, uint32_t *dst, uint32_t copy_words)
{
    const uint32_t seq = mcx_next_seq(ctx);
    mcx_done<<<1, copy_words > 64u ? 1024 : 64, 0, ctx->stream>>>(src, dst, copy_words,
                                                                 ctx->d_pinned + MCX_PINNED_SEQ_WORD, seq);
    HIPCHK(ctx, hipGetLastError());
    return mcx_wait_seq(ctx, seq) ? MCX_E_HIP : MCX_OK;
}

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
        hipHostMalloc((void **)&ctx->h_pinned, 4 * MCX_PINNED_WORDS, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&ctx->d_pinned, ctx->h_pinned, 0) != hipSuccess) {
        mcx_ctx_destroy(ctx);
        return MCX_E_NO_DEVICE;
    }
    memset(ctx->h_pinned, 0, 4 * MCX_PINNED_WORDS);
    if (mcx_k2_static_init(ctx) != MCX_OK) {
        mcx_ctx_destroy(ctx);
        return MCX_E_NO_DEVICE;
    }
    *out = ctx;
    return MCX_OK;
}

void mcx_ctx_destroy(mcx_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    mcx_free_table(ctx);
    mcx_free_graph(ctx);
    (void)hipFree(ctx->d_fleet);
    (void)hipFree(ctx->d_inv);
    (void)hipFree(ctx->d_io);
    if (ctx->h_io)
        (void)hipHostFree(ctx->h_io);
    if (ctx->h_pinned)
        (void)hipHostFree(ctx->h_pinned);
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

int mcx_ctx_tune(mcx_ctx *ctx, int knob, uint64_t value)
{
    if (!ctx)
        return MCX_E_BAD_ARG;
    switch (knob) {
    case MCX_TUNE_COMPILE_THREADS:
        if (value > 64u) {
            ctx->err = "MCX_TUNE_COMPILE_THREADS: at most 64";
            return MCX_E_BAD_ARG;
        }
        ctx->tune_threads = (uint32_t)value;
        return MCX_OK;
    case MCX_TUNE_SUB_MIN:
        if (value > 0xFFFFFFFFull) {
            ctx->err = "MCX_TUNE_SUB_MIN: does not fit 32 bits";
            return MCX_E_BAD_ARG;
        }
        ctx->tune_sub_min = (uint32_t)value;
        return MCX_OK;
    case MCX_TUNE_TILES_MIN:
        ctx->tune_tiles_min_set = value != MCX_TUNE_AUTO;
        ctx->tune_tiles_min = value > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)value;
        return MCX_OK;
    case MCX_TUNE_LIST_TWO_PASS:
        ctx->tune_two_pass = value != 0;
        return MCX_OK;
    default:
        ctx->err = "unknown tuning knob";
        return MCX_E_BAD_ARG;
    }
}

int mcx_dev_alloc(mcx_ctx *ctx, size_t bytes, void **d_out)
{
    if (!ctx || !d_out)
        return MCX_E_BAD_ARG;
    if (mcx_set_device(ctx))
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

} /* extern "C" */
