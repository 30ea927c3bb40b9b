/*
 * mcx_devfilter.hip - which devices of a fleet inventory a host's device list
 * would show, and the packing of their modalias strings for the alias match.
 *
 * Restates, over packed records, the tests that mc_list_all_devices
 * (reference montecarlo/libmontecarlo.c:233-440) and
 * mc_is_infrastructure_device (montecarlo/libmontecarlo.c:729-823) make on one
 * sysfs node at a time, plus mc_is_excluded_device
 * (montecarlo/libmontecarlo.c:826-874).  Byte and integer work, bound by the
 * stream of records and strings through HBM:
 *
 *   dv_names      one thread per driver name: the two name tests, once
 *   dv_classify   one thread per 16-byte record -> verdict byte
 *   dv_host<0>    one wave per host: rank of each passing device in its host,
 *                 the per-host limit, the host's count
 *   scan          counts -> kept_off
 *   dv_host<1>    one wave per host: source index and length of every kept string
 *   scan          lengths -> s_off_out
 *   dv_copy_*     16 lanes per kept string: the bytes, whole words where the arenas are aligned
 */
#include <string.h>

#include <unordered_map>

#include "mcx_internal.h"
#include "mcx_scan.h"
#include "tables.h"

#define DVN_INFRA 1u   /* one of the 8 PCI port / hot-plug drivers */
#define DVN_USBHOST 2u /* holds "hcd" or is "hub" */

/* infra: the names PCI devices are dropped for (montecarlo/libmontecarlo.c:778-785
 * unless a table file replaced them), packed: name k = infra[infra_off[k] .. infra_off[k+1]) */
__global__ void __launch_bounds__(FL_THREADS)
dv_names(const char *__restrict__ arena, const uint32_t *__restrict__ off, uint32_t D,
         const char *__restrict__ infra, const uint32_t *__restrict__ infra_off, uint32_t n_infra,
         uint8_t *__restrict__ bits)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D)
        return;
    const char *s = arena + off[d];
    const uint32_t n = off[d + 1] - off[d];
    uint32_t b = 0;
    for (uint32_t k = 0; k < n_infra; ++k) {
        const uint32_t a = infra_off[k];
        if (infra_off[k + 1] - a != n)
            continue;
        uint32_t i = 0;
        while (i < n && infra[a + i] == s[i])
            ++i;
        if (i == n)
            b |= DVN_INFRA;
    }
    /* strstr(name, "hcd") || strcmp(name, "hub") == 0   (libmontecarlo.c:413) */
    for (uint32_t i = 0; i + 3u <= n; ++i)
        if (s[i] == 'h' && s[i + 1] == 'c' && s[i + 2] == 'd')
            b |= DVN_USBHOST;
    if (n == 3u && s[0] == 'h' && s[1] == 'u' && s[2] == 'b')
        b |= DVN_USBHOST;
    bits[d] = (uint8_t)b;
}

__device__ __forceinline__ uint32_t dv_verdict(const uint4 r, const uint8_t *__restrict__ drv_bits, uint32_t D)
{
    const uint32_t sub = r.x & 0xFFu, devtype = (r.x >> 8) & 0xFFu, attrs = (r.x >> 16) & 0xFFu;
    const uint32_t iface_class = r.x >> 24, pci_class = r.y;
    const uint32_t parent_class = r.z & 0xFFu, own_class = (r.z >> 8) & 0xFFu, driver = r.w;
    const uint32_t nbits = driver < D ? drv_bits[driver] : 0u;

    /* mc_is_excluded_device: class 08 on the node, the interface or the parent */
    uint32_t v = 0;
    if (((attrs & MCX_DA_OWN_CLASS) && own_class == 8u) || ((attrs & MCX_DA_IFACE_CLASS) && iface_class == 8u) ||
        ((attrs & MCX_DA_PARENT) && (attrs & MCX_DA_PARENT_CLASS) && parent_class == 8u))
        v |= MCX_DV_EXCLUDED;

    uint32_t why = MCX_DR_NONE;
    if (sub == MCX_SUB_PCI) {
        const uint32_t base = (pci_class >> 16) & 0xFFu, subc = (pci_class >> 8) & 0xFFu;
        if ((attrs & MCX_DA_PCI_CLASS) && (base == 0x06u || (base == 0x0Cu && subc == 0x05u) || base == 0x08u))
            why = MCX_DR_PCI_CLASS;
        else if (driver == MCX_DRV_NOSLASH || (nbits & DVN_INFRA))
            why = MCX_DR_PCI_DRIVER;
    } else if (sub == MCX_SUB_SCSI) {
        if (devtype == MCX_DT_NONE || devtype == MCX_DT_SCSI_HOST || devtype == MCX_DT_SCSI_TARGET ||
            devtype == MCX_DT_SCSI_GENERIC || !(attrs & (MCX_DA_SCSI_MODEL | MCX_DA_SCSI_VENDOR)))
            why = MCX_DR_SCSI;
    } else if (sub == MCX_SUB_USB) {
        if (devtype != MCX_DT_USB_INTERFACE)
            why = MCX_DR_USB_NODE;
        else if ((attrs & MCX_DA_IFACE_CLASS) && iface_class == 9u)
            why = MCX_DR_USB_HUB;
        else if (!(attrs & MCX_DA_PARENT))
            why = MCX_DR_USB_NODE;
        else if ((attrs & MCX_DA_PARENT_CLASS) && parent_class == 9u)
            why = MCX_DR_USB_HUB;
        else if (nbits & DVN_USBHOST)
            why = MCX_DR_USB_HOST;
    } else if (sub != MCX_SUB_HID && sub != MCX_SUB_PCMCIA) {
        why = MCX_DR_SUBSYSTEM;
    }
    return why == MCX_DR_NONE ? (v | MCX_DV_LISTED) : (v | (why << 4));
}

__global__ void __launch_bounds__(FL_THREADS)
dv_classify(const uint4 *__restrict__ rec, uint32_t N, const uint8_t *__restrict__ drv_bits, uint32_t D,
            uint8_t *__restrict__ verdict)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N)
        verdict[i] = (uint8_t)dv_verdict(rec[i], drv_bits, D);
}

/*
 * One wave per host, its devices 64 at a time in enumeration order.
 * FILL = 0: apply the per-host limit to the verdicts and count the host's
 * listed devices.  FILL = 1: for the r-th listed device of host h write its
 * index and string length at kept_off[h] + r.
 */
template <int FILL>
__global__ void __launch_bounds__(FL_THREADS)
dv_host(const uint32_t *__restrict__ dev_off, uint32_t H, uint32_t max_per_host, uint8_t *verdict,
        uint32_t *__restrict__ cnt, const uint32_t *__restrict__ kept_off, const uint32_t *__restrict__ s_off,
        uint32_t *__restrict__ src, uint32_t *__restrict__ len)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (h >= H)
        return;
    const uint32_t b = dev_off[h], e = dev_off[h + 1];
    const uint32_t base = FILL ? kept_off[h] : 0u;
    uint32_t running = 0;
    for (uint32_t i0 = b; i0 < e; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t v = i < e ? verdict[i] : 0u;
        const bool keep = (v & MCX_DV_LISTED) != 0u;
        const unsigned long long m = __ballot(keep);
        const uint32_t r = running + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (FILL) {
            if (keep) {
                src[base + r] = i;
                len[base + r] = s_off[i + 1] - s_off[i];
            }
        } else if (keep && max_per_host && r >= max_per_host) {
            verdict[i] = (uint8_t)((v & MCX_DV_EXCLUDED) | (MCX_DR_OVER_MAX << 4));
        }
        running += (uint32_t)__popcll(m);
    }
    if (!FILL && lane == 0u)
        cnt[h] = (max_per_host && running > max_per_host) ? max_per_host : running;
}

/* 16 lanes per kept string, byte by byte: arenas that are not 4-byte aligned */
__global__ void __launch_bounds__(FL_THREADS)
dv_copy_bytes(const char *__restrict__ S, const uint32_t *__restrict__ s_off, const uint32_t *__restrict__ src,
              const uint32_t *__restrict__ s_off_out, uint32_t K, char *__restrict__ S_out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = t >> 4, sub = t & 15u;
    if (k >= K)
        return;
    const uint32_t from = s_off[src[k]], to = s_off_out[k], n = s_off_out[k + 1] - to;
    for (uint32_t b = sub; b < n; b += 16u)
        S_out[to + b] = S[from + b];
}

/*
 * 16 lanes per group of DV_BATCH kept strings, a destination word per lane and
 * string: the bytes in front of a string's first aligned destination word and
 * behind its last one go one by one, every word in between is put together
 * from the two aligned source words it straddles.  The index loads of the
 * batch are issued together, then its source words, then its stores, so that
 * a wave has DV_BATCH x 64 words in flight instead of 64.  Both arenas 4-byte
 * aligned; s_off[N] bounds the source reads.
 */
#define DV_BATCH 4u
__global__ void __launch_bounds__(FL_THREADS)
dv_copy_words(const uint32_t *__restrict__ Sw, uint32_t N, const uint32_t *__restrict__ s_off,
              const uint32_t *__restrict__ src, const uint32_t *__restrict__ s_off_out, uint32_t K,
              char *__restrict__ S_out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k0 = (t >> 4) * DV_BATCH, sub = t & 15u;
    if (k0 >= K)
        return;
    const char *S = (const char *)Sw;
    const uint32_t src_bytes = s_off[N];
    /* source word i; the arena's last, partly filled word is read byte by byte */
    auto word = [&](uint32_t i) __attribute__((always_inline)) {
        if (4u * i + 4u <= src_bytes)
            return Sw[i];
        uint32_t v = 0;
        for (uint32_t b = 4u * i; b < src_bytes; ++b)
            v |= (uint32_t)(uint8_t)S[b] << (8u * (b & 3u));
        return v;
    };
    uint32_t si[DV_BATCH], from[DV_BATCH], to[DV_BATCH], n[DV_BATCH];
#pragma unroll
    for (uint32_t q = 0; q < DV_BATCH; ++q)
        si[q] = k0 + q < K ? src[k0 + q] : 0u;
#pragma unroll
    for (uint32_t q = 0; q < DV_BATCH; ++q) {
        from[q] = s_off[si[q]];
        to[q] = s_off_out[min(k0 + q, K)];
        n[q] = k0 + q < K ? s_off_out[k0 + q + 1u] - to[q] : 0u;
    }
    uint32_t head[DV_BATCH], nwords[DV_BATCH], lo[DV_BATCH], hi[DV_BATCH];
    uint8_t hb[DV_BATCH], tb[DV_BATCH];
#pragma unroll
    for (uint32_t q = 0; q < DV_BATCH; ++q) {
        head[q] = min(n[q], (4u - (to[q] & 3u)) & 3u); /* bytes up to the first aligned destination word */
        nwords[q] = (n[q] - head[q]) >> 2;
        const uint32_t tail0 = head[q] + 4u * nwords[q];
        hb[q] = sub < head[q] ? (uint8_t)S[from[q] + sub] : (uint8_t)0;
        tb[q] = sub < n[q] - tail0 ? (uint8_t)S[from[q] + tail0 + sub] : (uint8_t)0;
        const uint32_t at = from[q] + head[q] + 4u * sub;
        lo[q] = sub < nwords[q] ? word(at >> 2) : 0u;
        hi[q] = (sub < nwords[q] && (at & 3u)) ? word((at >> 2) + 1u) : 0u;
    }
#pragma unroll
    for (uint32_t q = 0; q < DV_BATCH; ++q) {
        const uint32_t tail0 = head[q] + 4u * nwords[q];
        if (sub < head[q])
            S_out[to[q] + sub] = (char)hb[q];
        if (sub < n[q] - tail0)
            S_out[to[q] + tail0 + sub] = (char)tb[q];
        const uint32_t a = from[q] + head[q]; /* source byte of destination word 0 */
        uint32_t *dst = (uint32_t *)(S_out + to[q] + head[q]);
        if (sub < nwords[q])
            dst[sub] = __builtin_amdgcn_alignbit(hi[q], lo[q], ((a + 4u * sub) & 3u) * 8u);
        /* strings of more than 16 words: the rest, 16 words a step */
        for (uint32_t w = sub + 16u; w < nwords[q]; w += 16u) {
            const uint32_t at = a + 4u * w, i = at >> 2;
            const uint32_t l2 = word(i), h2 = (at & 3u) ? word(i + 1u) : 0u;
            dst[w] = __builtin_amdgcn_alignbit(h2, l2, (at & 3u) * 8u);
        }
    }
}

extern "C" {

int mcx_device_filter_dev(mcx_ctx *ctx, const void *d_rec, uint32_t N, const void *d_drv_arena,
                          const void *d_drv_off, uint32_t D, const void *d_dev_off, uint32_t H,
                          uint32_t max_per_host, void *d_verdict, void *d_kept_off, uint64_t *kept_total)
{
    if (!ctx || !d_dev_off || (N && (!d_rec || !d_verdict)) || (D && !d_drv_off))
        return MCX_E_BAD_ARG;
    if (kept_total)
        *kept_total = 0;
    if ((uintptr_t)d_rec & 15u) {
        ctx->err = "device records must be 16-byte aligned";
        return MCX_E_BAD_ARG;
    }
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    if (H == 0) {
        if (d_kept_off)
            HIPCHK(ctx, hipMemsetAsync(d_kept_off, 0, 4, ctx->stream));
        return MCX_OK;
    }
    const uint32_t nb = (H + SCAN_ITEMS - 1u) / SCAN_ITEMS;
    /* the active table's driver names, packed for the device */
    const mcx_tables_view *tv = mcx_tables_active();
    const uint32_t n_infra = tv->count[MCX_TS_INFRA];
    std::vector<uint32_t> infra_off(n_infra + 1, 0);
    std::string infra;
    for (uint32_t k = 0; k < n_infra; ++k) {
        infra += tv->item[MCX_TS_INFRA][k];
        infra_off[k + 1] = (uint32_t)infra.size();
    }
    /* scratch: name bits, per-host counts, block sums, offsets when the caller wants none, total, names */
    const size_t o_cnt = up256(D ? D : 1), o_bsum = o_cnt + up256(4 * (size_t)H);
    const size_t o_off = o_bsum + up256(4 * (size_t)nb), o_total = o_off + up256(4 * ((size_t)H + 1));
    const size_t o_ioff = o_total + 256, o_infra = o_ioff + up256(4 * infra_off.size());
    void *base;
    int rc = mcx_scratch(ctx, o_infra + up256(infra.size() + 1), &base);
    if (rc != MCX_OK)
        return rc;
    /* pageable sources: the copies are done when the calls return */
    HIPCHK(ctx, hipMemcpyAsync((char *)base + o_ioff, infra_off.data(), 4 * infra_off.size(), hipMemcpyHostToDevice,
                               ctx->stream));
    if (!infra.empty())
        HIPCHK(ctx, hipMemcpyAsync((char *)base + o_infra, infra.data(), infra.size(), hipMemcpyHostToDevice,
                                   ctx->stream));
    uint8_t *d_bits = (uint8_t *)base;
    uint32_t *d_cnt = (uint32_t *)((char *)base + o_cnt), *d_bsum = (uint32_t *)((char *)base + o_bsum);
    uint32_t *d_off = d_kept_off ? (uint32_t *)d_kept_off : (uint32_t *)((char *)base + o_off);
    unsigned long long *d_total = (unsigned long long *)((char *)base + o_total);
    if (D)
        dv_names<<<(D + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(
            (const char *)d_drv_arena, (const uint32_t *)d_drv_off, D, (const char *)base + o_infra,
            (const uint32_t *)((char *)base + o_ioff), n_infra, d_bits);
    if (N)
        dv_classify<<<(N + FL_THREADS - 1) / FL_THREADS, FL_THREADS, 0, ctx->stream>>>(
            (const uint4 *)d_rec, N, d_bits, D, (uint8_t *)d_verdict);
    const uint32_t host_blocks = (uint32_t)(((uint64_t)H * 64u + FL_THREADS - 1) / FL_THREADS);
    dv_host<0><<<host_blocks, FL_THREADS, 0, ctx->stream>>>((const uint32_t *)d_dev_off, H, max_per_host,
                                                           (uint8_t *)d_verdict, d_cnt, nullptr, nullptr, nullptr,
                                                           nullptr);
    mcx_scan_enqueue(ctx->stream, d_cnt, H, d_bsum, d_total, d_off);
    HIPCHK(ctx, hipGetLastError());
    unsigned long long *h_total = (unsigned long long *)ctx->h_pinned;
    HIPCHK(ctx, hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (kept_total)
        *kept_total = *h_total;
    return MCX_OK;
}

int mcx_fleet_compact_dev(mcx_ctx *ctx, const void *d_verdict, const void *d_S, const void *d_s_off, uint32_t N,
                          const void *d_dev_off, uint32_t H, const void *d_kept_off, uint32_t kept_total,
                          void *d_S_out, void *d_s_off_out, void *d_src, uint64_t *bytes_out)
{
    if (!ctx || !d_s_off || !d_dev_off || !d_kept_off || !d_s_off_out || (N && !d_verdict) || kept_total > N)
        return MCX_E_BAD_ARG;
    if (bytes_out)
        *bytes_out = 0;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    const uint32_t K = kept_total;
    if (K == 0 || H == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_s_off_out, 0, 4, ctx->stream));
        return MCX_OK;
    }
    const uint32_t nb = (K + SCAN_ITEMS - 1u) / SCAN_ITEMS;
    const size_t o_len = up256(4 * (size_t)K), o_bsum = 2 * o_len, o_total = o_bsum + up256(4 * (size_t)nb);
    void *base;
    int rc = mcx_scratch(ctx, o_total + 256, &base);
    if (rc != MCX_OK)
        return rc;
    uint32_t *src = d_src ? (uint32_t *)d_src : (uint32_t *)base;
    uint32_t *len = (uint32_t *)((char *)base + o_len), *d_bsum = (uint32_t *)((char *)base + o_bsum);
    unsigned long long *d_total = (unsigned long long *)((char *)base + o_total);
    const uint32_t host_blocks = (uint32_t)(((uint64_t)H * 64u + FL_THREADS - 1) / FL_THREADS);
    dv_host<1><<<host_blocks, FL_THREADS, 0, ctx->stream>>>((const uint32_t *)d_dev_off, H, 0u,
                                                           (uint8_t *)const_cast<void *>(d_verdict), nullptr,
                                                           (const uint32_t *)d_kept_off, (const uint32_t *)d_s_off,
                                                           src, len);
    mcx_scan_enqueue(ctx->stream, len, K, d_bsum, d_total, (uint32_t *)d_s_off_out);
    if (d_S_out && d_S) {
        const uint64_t threads = (uint64_t)K * 16u;
        const uint32_t blocks = (uint32_t)((threads + FL_THREADS - 1) / FL_THREADS);
        if ((((uintptr_t)d_S | (uintptr_t)d_S_out) & 3u) == 0u)
            dv_copy_words<<<(blocks + DV_BATCH - 1u) / DV_BATCH, FL_THREADS, 0, ctx->stream>>>((const uint32_t *)d_S, N,
                                                                  (const uint32_t *)d_s_off, src,
                                                                  (const uint32_t *)d_s_off_out, K, (char *)d_S_out);
        else
            dv_copy_bytes<<<blocks, FL_THREADS, 0, ctx->stream>>>((const char *)d_S, (const uint32_t *)d_s_off, src,
                                                                  (const uint32_t *)d_s_off_out, K, (char *)d_S_out);
    }
    HIPCHK(ctx, hipGetLastError());
    if (bytes_out) {
        unsigned long long *h_total = (unsigned long long *)ctx->h_pinned;
        HIPCHK(ctx, hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *bytes_out = *h_total;
    }
    return MCX_OK;
}

/* bound[k] = module named like the driver of the k-th listed device (drv_mod by driver number) */
__global__ void __launch_bounds__(256)
dv_bound_of_kept(const mcx_dev_rec_t *__restrict__ rec, const uint32_t *__restrict__ src, uint32_t K,
                 const uint32_t *__restrict__ drv_mod, uint32_t D, uint32_t *__restrict__ bound)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K)
        return;
    const uint32_t d = rec[src[k]].driver;
    bound[k] = d < D ? drv_mod[d] : MCX_NO_MATCH;
}

/* d_drv_mod == nullptr: used drivers from the alias match; otherwise from the records' bound drivers.
 * d_root_in_use != nullptr (bound mode only): the "(In Use)" byte of every listed module, from the
 * bound drivers of ALL the hosts' records on the pci / usb / pcmcia buses through d_mark_off / d_mark_ids */
static int inventory_roots(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off, uint32_t N,
                           const void *d_drv_arena, const void *d_drv_off, uint32_t D, const void *d_drv_mod,
                           const void *d_mark_off, const void *d_mark_ids,
                           const void *d_dev_off, uint32_t H, uint32_t max_per_host, const void *d_extra_off,
                           const void *d_extra_ids, uint32_t X, const void *d_safe_name, void *d_verdict,
                           void *d_root_off, void *d_root_ids, void *d_root_in_use, uint64_t cap, uint64_t *total_out)
{
    if (!ctx || !d_s_off || !d_dev_off || !d_root_off || !total_out || (N && !d_verdict))
        return MCX_E_BAD_ARG;
    if (!ctx->table_loaded || !ctx->graph_loaded)
        return MCX_E_NOT_LOADED;
    *total_out = 0;
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    /* the listed devices' strings live in a buffer of their own, kept by the context and
     * grown on demand: the stages share the context's scratch, which may move when it grows */
    uint32_t *h_bytes = ctx->h_pinned;
    HIPCHK(ctx, hipMemcpyAsync(h_bytes, (const uint32_t *)d_s_off + N, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t s_bytes = *h_bytes;
    const size_t o_soff = up256(s_bytes ? s_bytes : 1), o_kept = o_soff + up256(4 * ((size_t)N + 1));
    const size_t o_src = o_kept + up256(4 * ((size_t)H + 1)), o_bound = o_src + up256(4 * (size_t)(N ? N : 1));
    const size_t need = o_bound + up256(4 * (size_t)(N ? N : 1));
    if (ctx->inv_bytes < need) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_inv);
        ctx->d_inv = nullptr;
        ctx->inv_bytes = 0;
        if (hipMalloc(&ctx->d_inv, need) != hipSuccess) {
            ctx->err = "hipMalloc of the packed strings failed";
            return MCX_E_NOMEM;
        }
        ctx->inv_bytes = need;
    }
    char *tmp = (char *)ctx->d_inv;
    uint64_t kept = 0;
    int rc = mcx_device_filter_dev(ctx, d_rec, N, d_drv_arena, d_drv_off, D, d_dev_off, H, max_per_host, d_verdict,
                                   tmp + o_kept, &kept);
    if (rc == MCX_OK)
        rc = mcx_fleet_compact_dev(ctx, d_verdict, d_S, d_s_off, N, d_dev_off, H, tmp + o_kept, (uint32_t)kept, tmp,
                                   tmp + o_soff, d_drv_mod ? tmp + o_src : nullptr, nullptr);
    if (rc != MCX_OK)
        return rc;
    if (!d_drv_mod)
        return mcx_fleet_roots_dev(ctx, tmp, tmp + o_soff, (uint32_t)kept, tmp + o_kept, H, d_extra_off, d_extra_ids,
                                   X, d_safe_name, d_root_off, d_root_ids, cap, total_out);
    if (kept)
        dv_bound_of_kept<<<((uint32_t)kept + 255u) / 256u, 256, 0, ctx->stream>>>(
            (const mcx_dev_rec_t *)d_rec, (const uint32_t *)(tmp + o_src), (uint32_t)kept,
            (const uint32_t *)d_drv_mod, D, (uint32_t *)(tmp + o_bound));
    HIPCHK(ctx, hipGetLastError());
    mcx_fleet_labels lab;
    lab.rec = (const mcx_dev_rec_t *)d_rec; /* every record, listed or not */
    lab.rec_off = (const uint32_t *)d_dev_off;
    lab.n_rec = N;
    lab.mark_off = (const uint32_t *)d_mark_off;
    lab.mark_ids = (const uint32_t *)d_mark_ids;
    lab.D = D;
    lab.root_in_use = (uint8_t *)d_root_in_use;
    return mcx_fleet_roots_enqueue(ctx, tmp, tmp + o_soff, (uint32_t)kept, tmp + o_kept, H, d_extra_off, d_extra_ids,
                                   X, d_safe_name, kept ? tmp + o_bound : nullptr, d_root_in_use ? &lab : nullptr,
                                   d_root_off, d_root_ids, cap, total_out);
}

int mcx_fleet_inventory_roots_dev(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off,
                                  uint32_t N, const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                                  const void *d_dev_off, uint32_t H, uint32_t max_per_host,
                                  const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                                  const void *d_safe_name, void *d_verdict, void *d_root_off, void *d_root_ids,
                                  uint64_t cap, uint64_t *total_out)
{
    return inventory_roots(ctx, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D, nullptr, nullptr, nullptr, d_dev_off,
                           H, max_per_host, d_extra_off, d_extra_ids, X, d_safe_name, d_verdict, d_root_off,
                           d_root_ids, nullptr, cap, total_out);
}

int mcx_fleet_inventory_roots_bound_dev(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off,
                                        uint32_t N, const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                                        const void *d_drv_mod, const void *d_dev_off, uint32_t H,
                                        uint32_t max_per_host, const void *d_extra_off, const void *d_extra_ids,
                                        uint32_t X, const void *d_safe_name, void *d_verdict, void *d_root_off,
                                        void *d_root_ids, uint64_t cap, uint64_t *total_out)
{
    if (D && !d_drv_mod)
        return MCX_E_BAD_ARG;
    /* D == 0: every device is unbound; any non-null pointer selects the mode, nothing is read through it */
    return inventory_roots(ctx, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D, d_drv_mod ? d_drv_mod : d_rec,
                           nullptr, nullptr, d_dev_off, H, max_per_host, d_extra_off, d_extra_ids, X, d_safe_name,
                           d_verdict, d_root_off, d_root_ids, nullptr, cap, total_out);
}

int mcx_fleet_inventory_roots_labeled_dev(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off,
                                          uint32_t N, const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                                          const void *d_drv_mod, const void *d_mark_off, const void *d_mark_ids,
                                          const void *d_dev_off, uint32_t H, uint32_t max_per_host,
                                          const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                                          const void *d_safe_name, void *d_verdict, void *d_root_off,
                                          void *d_root_ids, void *d_root_in_use, uint64_t cap, uint64_t *total_out)
{
    if ((D && (!d_drv_mod || !d_mark_off || !d_mark_ids)) || (N && !d_rec) || (cap && !d_root_in_use))
        return MCX_E_BAD_ARG;
    return inventory_roots(ctx, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D, d_drv_mod ? d_drv_mod : d_rec,
                           d_mark_off, d_mark_ids, d_dev_off, H, max_per_host, d_extra_off, d_extra_ids, X,
                           d_safe_name, d_verdict, d_root_off, d_root_ids, cap ? d_root_in_use : nullptr, cap,
                           total_out);
}

/* host: the modules a bound driver's name stands for in mc_driver_is_in_use
 * (reference montecarlo/libmontecarlo.c:573-725): that function looks for device links under
 * /sys/bus/{pci,usb,pcmcia}/drivers/<n> with n = the module's name and, through get_driver_names
 * (:538-566), its name variant - rtw88_X -> rtw_X, else rtlX -> rtw_X (the active table's
 * [name_variant] rules; names cut at 127 bytes as the reference's buffers cut them).  Seen from
 * the driver: driver name d marks module m iff d is m's name or m's variant; a module without a
 * name marks nothing (:575).  CSR by driver number, module ids ascending within a driver. */
int mcx_driver_marks(const char *drv_arena, const uint32_t *drv_off, uint32_t D, const char *mod_names,
                     const uint32_t *mod_off, uint32_t V, uint32_t *mark_off, uint32_t *mark_ids, uint64_t cap,
                     uint64_t *total_out)
{
    if (!mark_off || !total_out || (D && !drv_off) || (V && !mod_off) || (cap && !mark_ids))
        return MCX_E_BAD_ARG;
    for (uint32_t d = 0; d < D; ++d)
        if (drv_off[d + 1] < drv_off[d])
            return MCX_E_BAD_ARG;
    for (uint32_t v = 0; v < V; ++v)
        if (mod_off[v + 1] < mod_off[v])
            return MCX_E_BAD_ARG;
    if ((D && drv_off[D] && !drv_arena) || (V && mod_off[V] && !mod_names))
        return MCX_E_BAD_ARG;
    const mcx_tables_view *tv = mcx_tables_active();
    std::unordered_map<std::string, std::vector<uint32_t>> by_name; /* a name a driver may carry -> modules */
    by_name.reserve(2 * (size_t)V);
    for (uint32_t v = 0; v < V; ++v) {
        std::string name(mod_names + mod_off[v], mod_off[v + 1] - mod_off[v]);
        if (name.empty())
            continue;
        std::string first = name.substr(0, 127);
        by_name[first].push_back(v);
        for (uint32_t k = 0; k + 1 < tv->count[MCX_TS_VARIANT]; k += 2) {
            const char *prefix = tv->item[MCX_TS_VARIANT][k];
            const size_t pl = strlen(prefix);
            if (name.compare(0, pl, prefix) == 0) {
                std::string second = (std::string(tv->item[MCX_TS_VARIANT][k + 1]) + name.substr(pl)).substr(0, 127);
                if (second != first)
                    by_name[second].push_back(v);
                break; /* the first rule that fits, no other */
            }
        }
    }
    uint64_t total = 0;
    mark_off[0] = 0;
    for (uint32_t d = 0; d < D; ++d) {
        auto it = by_name.find(std::string(drv_arena + drv_off[d], drv_off[d + 1] - drv_off[d]));
        if (it != by_name.end())
            for (uint32_t v : it->second) { /* pushed in ascending module order */
                if (total < cap)
                    mark_ids[total] = v;
                ++total;
            }
        if (total > 0xFFFFFFFFull)
            return MCX_E_TOO_LARGE;
        mark_off[d + 1] = (uint32_t)total;
    }
    *total_out = total;
    return total > cap ? MCX_E_TOO_LARGE : MCX_OK;
}

/* host: drv_mod[d] = id of the module whose name is byte for byte the d-th driver name
 * (the comparison of reference desktop/ui.py:1567: no '-' / '_' folding), MCX_NO_MATCH if none */
int mcx_driver_modules(const char *drv_arena, const uint32_t *drv_off, uint32_t D, const char *mod_names,
                       const uint32_t *mod_off, uint32_t V, uint32_t *drv_mod)
{
    if ((D && (!drv_off || !drv_mod)) || (V && !mod_off))
        return MCX_E_BAD_ARG;
    for (uint32_t d = 0; d < D; ++d)
        if (drv_off[d + 1] < drv_off[d])
            return MCX_E_BAD_ARG;
    for (uint32_t v = 0; v < V; ++v)
        if (mod_off[v + 1] < mod_off[v])
            return MCX_E_BAD_ARG;
    if ((D && drv_off[D] && !drv_arena) || (V && mod_off[V] && !mod_names))
        return MCX_E_BAD_ARG;
    std::unordered_map<std::string, uint32_t> by_name;
    by_name.reserve(V);
    for (uint32_t v = 0; v < V; ++v) /* the first module of a name keeps it */
        by_name.emplace(std::string(mod_names + mod_off[v], mod_off[v + 1] - mod_off[v]), v);
    for (uint32_t d = 0; d < D; ++d) {
        auto it = by_name.find(std::string(drv_arena + drv_off[d], drv_off[d + 1] - drv_off[d]));
        drv_mod[d] = it == by_name.end() ? MCX_NO_MATCH : it->second;
    }
    return MCX_OK;
}

int mcx_device_filter(mcx_ctx *ctx, const mcx_dev_rec_t *rec, uint32_t N, const char *drv_arena,
                      const uint32_t *drv_off, uint32_t D, const uint32_t *dev_off, uint32_t H,
                      uint32_t max_per_host, uint8_t *verdict, uint32_t *kept_off)
{
    if (!ctx || !dev_off || (N && (!rec || !verdict)) || (D && (!drv_off || (drv_off[D] && !drv_arena))))
        return MCX_E_BAD_ARG;
    if (dev_off[0] != 0 || dev_off[H] != N) {
        ctx->err = "dev_off must run from 0 to N";
        return MCX_E_BAD_ARG;
    }
    for (uint32_t h = 0; h < H; ++h)
        if (dev_off[h + 1] < dev_off[h]) {
            ctx->err = "dev_off not monotonic";
            return MCX_E_BAD_ARG;
        }
    for (uint32_t d = 0; d < D; ++d)
        if (drv_off[d + 1] < drv_off[d]) {
            ctx->err = "drv_off not monotonic";
            return MCX_E_BAD_ARG;
        }
    for (uint32_t i = 0; i < N; ++i)
        if (rec[i].driver >= D && rec[i].driver < MCX_DRV_NOSLASH) {
            ctx->err = "record " + std::to_string(i) + " names a driver outside the table";
            return MCX_E_BAD_ARG;
        }
    if (mcx_set_device(ctx))
        return MCX_E_HIP;
    const size_t nb_arena = D ? drv_off[D] : 0;
    const size_t o_arena = up256(16 * (size_t)(N ? N : 1)), o_doff = o_arena + up256(nb_arena ? nb_arena : 1);
    const size_t o_dev = o_doff + up256(4 * ((size_t)D + 1)), o_ver = o_dev + up256(4 * ((size_t)H + 1));
    const size_t o_kept = o_ver + up256(N ? N : 1), bytes = o_kept + up256(4 * ((size_t)H + 1));
    char *d = nullptr;
    if (hipMalloc((void **)&d, bytes) != hipSuccess) {
        ctx->err = "hipMalloc of the staging buffers failed";
        return MCX_E_NOMEM;
    }
    hipError_t e = hipSuccess;
    if (N)
        e = hipMemcpyAsync(d, rec, 16 * (size_t)N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nb_arena)
        e = hipMemcpyAsync(d + o_arena, drv_arena, nb_arena, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && D)
        e = hipMemcpyAsync(d + o_doff, drv_off, 4 * ((size_t)D + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + o_dev, dev_off, 4 * ((size_t)H + 1), hipMemcpyHostToDevice, ctx->stream);
    int rc = MCX_OK;
    if (e != hipSuccess) {
        ctx->err = std::string("copy in: ") + hipGetErrorString(e);
        rc = MCX_E_HIP;
    }
    uint64_t total = 0;
    if (rc == MCX_OK)
        rc = mcx_device_filter_dev(ctx, d, N, d + o_arena, d + o_doff, D, d + o_dev, H, max_per_host, d + o_ver,
                                   d + o_kept, &total);
    if (rc == MCX_OK) {
        if (N)
            e = hipMemcpyAsync(verdict, d + o_ver, N, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && kept_off)
            e = hipMemcpyAsync(kept_off, d + o_kept, 4 * ((size_t)H + 1), hipMemcpyDeviceToHost, ctx->stream);
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

} /* extern "C" */
