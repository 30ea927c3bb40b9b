/*
 * mcx_internal.h - the context behind include/mcx.h and small helpers shared by
 * the translation units of libmontecarlo.so (not installed, not part of the ABI).
 */
#ifndef MCX_INTERNAL_H
#define MCX_INTERNAL_H

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/mcx.h"
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
    uint32_t n_indexed = 0;          /* patterns that can match anything       */
    uint32_t *d_slot = nullptr;      /* n_indexed program slots, index order   */
    uint32_t *d_ext = nullptr;       /* overflow programs                      */
    uint32_t *d_sig = nullptr;       /* n_indexed gram signatures (+ padding)  */
    uint32_t *d_dup_start = nullptr; /* per slot: where its copies start in d_dup_idx (mult > 1 only) */
    uint32_t *d_dup_idx = nullptr;   /* table positions of repeated pattern texts */
    uint32_t *d_lens = nullptr;      /* nlens distinct key lengths             */
    mcx_slot_t *d_slots = nullptr;   /* prefix index                           */
    uint8_t *d_tags = nullptr;       /* one tag byte per index slot            */
    mcx_sub_t *d_sub = nullptr;      /* second index of the large groups       */
    uint8_t *d_sub_tags = nullptr;
    uint32_t sub_mask = 0, n_sub_groups = 0, max_group = 0;
    uint32_t slot_mask = 0;
    uint32_t nlens = 0;
    uint32_t *d_sets = nullptr;
    uint32_t *d_amod = nullptr;
    uint8_t *d_hwtag = nullptr;
    uint64_t table_bytes = 0;
    /* what mcx_ctx_tune left for the next table load, and what the loaded table took */
    uint32_t tune_threads = 0, tune_sub_min = 0;
    bool tune_tiles_min_set = false, tune_two_pass = false;
    uint32_t tune_tiles_min = 0;
    bool bulk_min_set = false;       /* the loaded table has a batch size of its own for the tile kernel */
    uint32_t bulk_min = 0;
    bool list_two_pass = false;      /* match lists of the loaded table by count + fill always */

    /* dependency graph: forward CSR as given, reverse CSR built at load */
    uint32_t V = 0, E = 0;
    bool graph_loaded = false;
    uint32_t *d_rowptr = nullptr, *d_col = nullptr;
    uint32_t *d_rrowptr = nullptr, *d_rcol = nullptr;
    uint32_t *d_part = nullptr;    /* row blocks of the reverse CSR (n_part + 1) */
    uint32_t n_part = 0;
    /* one slice of rows per CU for the variant of K2a that keeps the whole
     * LOADED bitmap in LDS (n_slices = 0: the graph is too large for it) */
    void *d_slices = nullptr;      /* n_slices records (r0, r1, e0, ne, first group of packed ids) */
    uint32_t *d_rpk = nullptr;     /* the slices' holder ids, pk_bits (20 or 32) bits each, in the kernel's layout */
    uint32_t n_slices = 0, pk_bits = 0, cu_rows_max = 0, cu_edges_max = 0;
    uint32_t *d_queue[2] = {nullptr, nullptr}; /* BFS frontiers, V (first entry, end) pairs each */
    uint32_t *d_bfs = nullptr;     /* BFS counters, then two hub lists of hub_cap pairs, then the bitmap of loaded
                                      modules not reached yet */
    bool bfs_clean = false;        /* the counters are zero (the last search ended and cleared them) */
    uint32_t n_hub[2] = {0, 0};    /* modules with a long adjacency list, per closure direction */
    uint32_t hub_cap = 0;
    uint32_t *h_pinned = nullptr;  /* MCX_PINNED_WORDS words of pinned host memory for small read-backs */
    uint32_t *d_pinned = nullptr;  /* the same words as the device addresses them */
    uint32_t pinned_seq = 0;       /* last value a kernel was asked to leave in h_pinned[MCX_PINNED_SEQ_WORD] */
    std::vector<uint32_t> bfs_sizes[2]; /* per direction: the frontier sizes of the last closure's per-level
                                           launches (0xFFFFFFFF: a level with hubs) */
    uint32_t bfs_depth[2] = {0, 0}; /* per-level launches the last closure needed per direction behind its
                                       one-workgroup part (0: none, or no search yet) */

    /* holder levels of the graph (fleet pipeline): level 0 = modules nobody
     * needs, level k = modules whose holders all sit in levels < k.  lvl_order
     * lists the modules level by level, lvl_ptr[k]..lvl_ptr[k+1] is level k.
     * Empty lvl_ptr: the graph has a cycle and has no such order. */
    uint32_t *d_lvl_order = nullptr;
    std::vector<uint32_t> lvl_ptr;
    std::vector<uint32_t> lvl_maxdeg; /* per level: the most holders one of its modules has */

    /* staging of the host-buffer alias match, grown on demand */
    void *d_io = nullptr;
    size_t io_bytes = 0;
    void *h_io = nullptr;          /* pinned mirror of d_io for small host-buffer calls */
    void *h_io_dev = nullptr;      /* h_io as the device addresses it */
    size_t h_io_bytes = 0;

    /* packed strings of the listed devices (inventory pipeline), grown on demand */
    void *d_inv = nullptr;
    size_t inv_bytes = 0;

    /* scratch of the fleet pipeline, grown on demand, freed with the context */
    void *d_fleet = nullptr;
    size_t fleet_bytes = 0;
};

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return MCX_E_HIP;                                                          \
        }                                                                              \
    } while (0)

static inline int mcx_set_device(mcx_ctx *ctx)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return 0;
}

/* the context's scratch for fleet-sized work, grown on demand (the stream is
 * drained before the old block is freed), freed with the context */
static inline int mcx_scratch(mcx_ctx *ctx, size_t bytes, void **out)
{
    if (ctx->fleet_bytes < bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_fleet);
        ctx->d_fleet = nullptr;
        ctx->fleet_bytes = 0;
        if (hipMalloc(&ctx->d_fleet, bytes) != hipSuccess) {
            ctx->err = "fleet scratch of " + std::to_string(bytes) +
                       " bytes does not fit the device: split the fleet into groups of hosts";
            return MCX_E_NOMEM;
        }
        ctx->fleet_bytes = bytes;
    }
    *out = ctx->d_fleet;
    return MCX_OK;
}

/*
 * The words of h_pinned (MCX_PINNED_WORDS of them), by user:
 *   0 .. 35   closure search: sizes of a batch of levels, edge count, levels handed over
 *             (mcx_k2.hip); the fleet calls and the device filter: a 64-bit total in 0..1
 *   40        sequence word: what a read-back kernel stores last (grows with every use)
 *   44 .. 49  the closure's early report (a one-workgroup launch that ended the search)
 *   52 .. 55  the match lists' report: total and overflow count, each marked with the call's
 *             number (mcx_k1.hip)
 *   64 .. 79  the closure's early report, second part: the frontier sizes of the batch's levels
 * A closure call may return on the early report while the rest of its batch is still queued:
 * those launches find the search over, write zeros into 0..35 and the batch's number into 40
 * after the call has returned.  They are stream-ordered in front of every later kernel, and
 * host code trusts a word only for its mark (a sequence number it waits for, or a total a
 * stream synchronisation stands behind), never for being unchanged since it last looked;
 * the protocols that spin on marks of their own have words of their own.
 */
#define MCX_PINNED_SEQ_WORD 40u
#define MCX_PINNED_LIST_WORD 52u
#define MCX_PINNED_HIST_WORD 64u
#define MCX_PINNED_WORDS 96u

/* the next value of the sequence word (never 0: kernels take 0 as "do not report") */
static inline uint32_t mcx_next_seq(mcx_ctx *ctx)
{
    if (++ctx->pinned_seq == 0u)
        ctx->pinned_seq = 1u;
    return ctx->pinned_seq;
}

/* Wait for a read-back kernel on the context's stream: spin on the sequence word it stores
 * last, behind a system-scope fence, into pinned host memory (a few microseconds against
 * about 25 for a stream synchronisation); after 200 us without it, synchronise the stream
 * instead (which also reports a failed launch). */
static inline int mcx_wait_seq(mcx_ctx *ctx, uint32_t want)
{
    volatile uint32_t *seq = ctx->h_pinned + MCX_PINNED_SEQ_WORD;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (*seq == want) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return 0;
        }
        if ((spins & 63u) == 63u &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200))
            break;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (*seq != want) {
        ctx->err = "read-back kernel did not finish";
        return MCX_E_HIP;
    }
    return 0;
}

/* The same for two kernels of which either may be the one that has the answer: returns as soon as
 * the word holds `early` or `last` (the kernel storing `last` runs behind the other, so after the
 * fallback synchronisation the word holds `last`). */
static inline int mcx_wait_seq2(mcx_ctx *ctx, uint32_t early, uint32_t last)
{
    volatile uint32_t *seq = ctx->h_pinned + MCX_PINNED_SEQ_WORD;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        const uint32_t now = *seq;
        if (now == early || now == last) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return 0;
        }
        if ((spins & 63u) == 63u &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200))
            break;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (*seq != last) {
        ctx->err = "read-back kernel did not finish";
        return MCX_E_HIP;
    }
    return 0;
}

/* Small host-buffer calls use no copies: their inputs are put into pinned host memory kept by the
 * context (MCX_ZERO_COPY_MAX bytes, allocated at the first such call), the kernels read them there
 * over the bus and write the results there.  mcx_pinned_io hands out that area as the host and as
 * the device address it; mcx_signal_and_wait enqueues a one-wave launch that leaves a fresh
 * sequence number in the report area behind everything the stream has done, and waits for it
 * (mcx_wait_seq).  `copy_words` > 0: that launch first copies so many words from device memory
 * `src` to `dst` (pinned, device address) - results that kernels reach with atomics stay in
 * device memory until then.  (mcx_core.hip) */
#define MCX_ZERO_COPY_MAX ((size_t)2048 << 10)
int mcx_pinned_io(mcx_ctx *ctx, uint8_t **host, uint8_t **dev);
int mcx_signal_and_wait(mcx_ctx *ctx, const uint32_t *src = nullptr, uint32_t *dst = nullptr, uint32_t copy_words = 0);

static inline size_t up256(size_t x)
{
    return (x + 255) & ~(size_t)255;
}

/* The context's device staging area for host-buffer calls, grown on demand (one
 * allocation kept between calls: a hipMalloc / hipFree pair per call costs more
 * than the small kernels it would serve).  Calls on one context are serial. */
static inline int mcx_io_reserve(mcx_ctx *ctx, size_t need, uint8_t **base)
{
    if (ctx->io_bytes < need) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess)
            return MCX_E_HIP;
        (void)hipFree(ctx->d_io);
        ctx->d_io = nullptr;
        ctx->io_bytes = 0;
        const size_t want = need + need / 2;
        if (hipMalloc(&ctx->d_io, want) != hipSuccess) {
            ctx->err = "hipMalloc of " + std::to_string(want) + " bytes of staging failed";
            return MCX_E_NOMEM;
        }
        ctx->io_bytes = want;
    }
    *base = (uint8_t *)ctx->d_io;
    return MCX_OK;
}

template <class T>
static int mcx_dev_upload(mcx_ctx *ctx, T **d_out, const T *src, size_t count)
{
    size_t bytes = sizeof(T) * (count ? count : 1);
    HIPCHK(ctx, hipMalloc((void **)d_out, bytes));
    if (count)
        HIPCHK(ctx, hipMemcpyAsync(*d_out, src, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

void mcx_free_table(mcx_ctx *ctx);
/* d_hw[m] = 1 iff module m < n_modules owns a hardware-bus alias; enqueues */
int mcx_k1_hw_flags_dev(mcx_ctx *ctx, uint32_t n_modules, uint8_t *d_hw);

/* Where the "(In Use)" byte of every listed module comes from (device pointers): per-host
 * lists of marked modules (inuse_off has H + 1 entries), and / or the records of all the
 * hosts' devices (rec_off: H + 1 offsets into rec) with the driver-number -> modules map of
 * mcx_driver_marks; root_in_use receives one byte per entry of root_ids. */
struct mcx_fleet_labels {
    const uint32_t *inuse_off = nullptr, *inuse_ids = nullptr;
    uint32_t Y = 0;
    const mcx_dev_rec_t *rec = nullptr;
    const uint32_t *rec_off = nullptr;
    uint32_t n_rec = 0;
    const uint32_t *mark_off = nullptr, *mark_ids = nullptr;
    uint32_t D = 0;
    uint8_t *root_in_use = nullptr;
};
/* the fleet pipeline behind every mcx_fleet_roots* entry point (mcx_fleet.hip) */
int mcx_fleet_roots_enqueue(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, const void *d_dev_off,
                            uint32_t H, const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                            const void *d_safe_name, const void *d_bound_mod, const mcx_fleet_labels *lab,
                            void *d_root_off, void *d_root_ids, uint64_t cap, uint64_t *total_out);
void mcx_free_graph(mcx_ctx *ctx);
/* device address of the static bitmap of mcx_dep_flags_noalloc_dev on the context's
 * device (which must be current); called once per context at creation */
int mcx_k2_static_init(mcx_ctx *ctx);

#endif /* MCX_INTERNAL_H */
