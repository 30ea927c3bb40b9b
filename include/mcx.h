/*
 * mcx.h - batch (GPU) entry points of the MI355X build of libmontecarlo.
 *
 * The reference library answers "which driver serves this device" and "which
 * loaded modules are roots / in use" one object at a time, by shelling out to
 * kmod and walking sysfs:
 *   - alias -> module resolution happens inside `modprobe`,
 *       reference montecarlo/libmontecarlo.c:115-127 (mc_try_load_driver),
 *       montecarlo/helper.c:76-82, and inside `modinfo -F alias`,
 *       reference desktop/ui.py:1526-1531;
 *   - holders / in-use / root filtering is one opendir per module,
 *       reference montecarlo/libmontecarlo.c:446-473 (mc_module_has_holders),
 *       montecarlo/libmontecarlo.c:573-725 (mc_driver_is_in_use),
 *       desktop/ui.py:1565-1599 (root-module filter loop).
 * A scalar per-string ABI cannot feed a GPU, so the same questions are
 * exposed here over whole tables.  All entry points are plain C: pointers and
 * sizes only.  Every function returns 0 on success or a negative MCX_E_* code;
 * nothing is computed on the CPU as a fallback: without a usable HIP device
 * mcx_ctx_create() fails with MCX_E_NO_DEVICE.
 *
 * Buffers named d_* are device pointers (hipMalloc memory or any pointer the
 * HIP runtime accepts on the context's device); all others are host pointers.
 * A context owns (or borrows) one HIP stream; *_dev entry points only enqueue
 * work on it and return, except the ones that hand a total or a count back to
 * the host (mcx_dep_closure_dev, mcx_alias_match_list_dev, the fleet calls, the
 * device filter: each says so), which wait for that number; host-buffer entry
 * points synchronise before returning.
 * A context is not thread safe: use one per thread.
 */
#ifndef MCX_H
#define MCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcx_ctx mcx_ctx;

enum {
    MCX_OK = 0,
    MCX_E_NO_DEVICE = -1,     /* no HIP device / HIP runtime error at create   */
    MCX_E_BAD_ARG = -2,       /* NULL pointer, bad size, table not loaded ...  */
    MCX_E_HIP = -3,           /* a HIP call failed; see mcx_last_error()       */
    MCX_E_NOMEM = -4,
    MCX_E_PATTERN = -5,       /* pattern outside the supported grammar         */
    MCX_E_TOO_LARGE = -6,     /* size beyond a documented limit                */
    MCX_E_NOT_LOADED = -7     /* alias table / dep graph not loaded yet        */
};

#define MCX_NO_MATCH 0xFFFFFFFFu

/* flags for mcx_alias_table_load */
#define MCX_TABLE_NORMALIZE_DASH 1u  /* map '-' to '_' outside [...] in patterns
                                        and in queried strings, as kmod does
                                        before it calls fnmatch               */

/* in_flags bits (per module) for the dependency kernels */
#define MCX_F_LOADED 1u
#define MCX_F_BOUND  2u
#define MCX_F_SAFE   4u
/* out_flags bits (per module) */
#define MCX_O_HAS_HOLDERS 1u
#define MCX_O_IN_USE      2u
#define MCX_O_ROOT        4u

/* direction for mcx_dep_closure */
#define MCX_CLOSURE_HOLDERS 0  /* everything that (transitively) needs a seed  */
#define MCX_CLOSURE_DEPS    1  /* everything a seed (transitively) needs       */

/* ---- context ------------------------------------------------------------ */

/* stream: a hipStream_t to borrow (e.g. the caller's current stream), or NULL
 * to let the context create and own one. */
int  mcx_ctx_create(int device_ordinal, void *stream, mcx_ctx **out);
void mcx_ctx_destroy(mcx_ctx *ctx);
int  mcx_sync(mcx_ctx *ctx);
void *mcx_ctx_stream(mcx_ctx *ctx);
const char *mcx_strerror(int rc);
/* text of the last failure on this context (HIP error string, offending
 * pattern index ...); never NULL. */
const char *mcx_last_error(mcx_ctx *ctx);

/*
 * Tuning knobs.  None of them changes a result; they choose between kernels and set the host
 * threads of the table compiler, and they apply to tables loaded after the call.  The library
 * reads no environment variable: a process that wants a knob turned calls this.
 *   MCX_TUNE_COMPILE_THREADS  host threads mcx_alias_table_load may use (0 = one per core,
 *                             at most 16; the table's arrays do not depend on the number)
 *   MCX_TUNE_SUB_MIN          group size above which a group of the prefix index gets the
 *                             second (anchor-gram) index (0 = default, 128)
 *   MCX_TUNE_TILES_MIN        batch size from which the tile kernel (large batches) runs
 *                             instead of the small-batch kernel; MCX_TUNE_AUTO = the library's
 *                             own rule (more than 8192 strings, earlier for tables with very
 *                             large groups)
 *   MCX_TUNE_LIST_TWO_PASS    1: match lists by a counting and a filling pass always (the path
 *                             otherwise taken only when a string is matched by more than 16
 *                             distinct pattern texts)
 */
#define MCX_TUNE_COMPILE_THREADS 1
#define MCX_TUNE_SUB_MIN 2
#define MCX_TUNE_TILES_MIN 3
#define MCX_TUNE_LIST_TWO_PASS 4
#define MCX_TUNE_AUTO 0xFFFFFFFFFFFFFFFFull
int mcx_ctx_tune(mcx_ctx *ctx, int knob, uint64_t value);

/* device memory helpers for callers that have no HIP binding of their own */
int mcx_dev_alloc(mcx_ctx *ctx, size_t bytes, void **d_out);
int mcx_dev_free(mcx_ctx *ctx, void *d_ptr);
int mcx_h2d(mcx_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int mcx_d2h(mcx_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* stream-ordered timing: begin/end bracket work enqueued on the context's
 * stream with HIP events; end returns the elapsed milliseconds. */
int mcx_timer_begin(mcx_ctx *ctx);
int mcx_timer_end(mcx_ctx *ctx, float *ms_out);

/* ---- K1: modalias x alias-pattern glob match ---------------------------- */

/*
 * Load a modules.alias-style table: P patterns packed in `arena`, pattern j is
 * arena[a_off[j] .. a_off[j+1]) (no terminator needed, no NUL inside), owned
 * by module id a_mod[j].  Table order is file order.  Replaces the alias
 * lookup kmod performs for the reference at montecarlo/libmontecarlo.c:123.
 * Semantics of a pattern are those of glibc fnmatch(pattern, string, 0) in
 * the C locale.  Supported grammar: literals, '?', '*', '\' escapes and
 * bracket expressions with literals, ranges, leading '!' or '^' and leading
 * ']'.  POSIX [:class:], [=x=] and [.x.] sub-expressions are rejected with
 * MCX_E_PATTERN (no modules.alias line uses them).  At most 2^26 patterns per
 * table (MCX_E_BAD_ARG beyond; larger tables are sharded over contexts).
 */
int mcx_alias_table_load(mcx_ctx *ctx, const char *arena, const uint32_t *a_off,
                         const uint32_t *a_mod, uint32_t P, uint32_t flags);

/*
 * Match N strings (string i = S[s_off[i] .. s_off[i+1])) against the loaded
 * table.  first_idx[i] = smallest j with fnmatch(pattern_j, string_i, 0) == 0,
 * or MCX_NO_MATCH; count[i] = number of matching patterns.  Host buffers.
 */
int mcx_alias_match(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N,
                    uint32_t *first_idx, uint32_t *count);
/* Same with everything resident on the device; enqueues and returns.  The
 * host-buffer call checks that s_off is non-decreasing; here the offsets are
 * out of the host's sight and are trusted: d_s_off must hold N + 1
 * non-decreasing offsets into d_S (4-byte aligned d_S is read by words). */
int mcx_alias_match_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N,
                        void *d_first_idx, void *d_count);
/*
 * Every matching table position per string, ascending - the order in which
 * kmod lists the modules of a modalias (all matching lines of modules.alias,
 * in file order).  CSR output: list_off[i] .. list_off[i+1] index list_idx;
 * with K > 0 only the first K positions of a string are kept (K = 0: all).
 * *total_out = list_off[N]; when list_idx (cap entries) is too small the call
 * returns MCX_E_TOO_LARGE with *total_out and list_off filled in (list_idx then
 * holds an unspecified part of the lists, never more than cap entries).  Costs
 * one pass of the match and one more launch (a second pass only when some
 * string is matched by more than 16 distinct pattern texts).  Strings that kmod would refuse
 * (MCX_TABLE_NORMALIZE_DASH tables) have empty lists.
 * The device call returns as soon as the total is known; list_off and list_idx
 * are complete in stream order, like the results of every other *_dev call.
 */
int mcx_alias_match_list(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N, uint32_t K,
                         uint32_t *list_off, uint32_t *list_idx, uint64_t cap, uint64_t *total_out);
int mcx_alias_match_list_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N, uint32_t K,
                             void *d_list_off, void *d_list_idx, uint64_t cap, uint64_t *total_out);

/* driver_mod[i] = a_mod[first_idx[i]] (MCX_NO_MATCH if none); device buffers. */
int mcx_alias_driver_mod_dev(mcx_ctx *ctx, const void *d_first_idx, uint32_t N,
                             void *d_driver_mod);
/* hw_alias[m] = 1 iff module m owns a pattern whose text starts with one of
 * pci: usb: platform: hid: serio: of:  (reference desktop/ui.py:1540-1549);
 * n_modules bytes are written to the host buffer. */
int mcx_alias_hw_flags(mcx_ctx *ctx, uint32_t n_modules, uint8_t *hw_alias);
/* number of patterns / bytes of device memory held by the table */
int mcx_alias_table_info(mcx_ctx *ctx, uint32_t *P_out, uint64_t *device_bytes_out);

/* ---- K2: module dependency graph ---------------------------------------- */

/*
 * Load the direct-dependency graph in CSR form: edge u -> col[k] for
 * k in [rowptr[u], rowptr[u+1]) means "u needs col[k]" (col[k] is held by u,
 * the relation sysfs shows as /sys/module/<col[k]>/holders/<u>, read by the
 * reference at montecarlo/libmontecarlo.c:446-473).
 */
int mcx_dep_graph_load(mcx_ctx *ctx, const uint32_t *rowptr, const uint32_t *col,
                       uint32_t V, uint32_t E);
/*
 * out_flags[v]:  HAS_HOLDERS = LOADED[v] and some LOADED u has edge u -> v
 *                IN_USE      = LOADED[v] and (BOUND[v] or HAS_HOLDERS[v])
 *                              (montecarlo/libmontecarlo.c:573-725)
 *                ROOT        = LOADED[v] and SAFE[v] and not HAS_HOLDERS[v]
 *                              (desktop/ui.py:1565-1577)
 * refcnt (optional, may be NULL): number of LOADED holders of v plus 1 if
 * BOUND[v]; 0xFFFFFFFF for modules that are not loaded, the batch analogue of
 * the -1 of mc_get_module_refcount (montecarlo/libmontecarlo.c:476-495).
 */
int mcx_dep_flags(mcx_ctx *ctx, const uint8_t *in_flags, uint8_t *out_flags,
                  uint32_t *refcnt);
int mcx_dep_flags_dev(mcx_ctx *ctx, const void *d_in_flags, void *d_out_flags,
                      void *d_refcnt);
/*
 * mcx_dep_flags_dev for callers that cannot allocate (code under HIP graph
 * capture, serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  Same arguments and results; the LOADED bitmap
 * the kernels gather from lives in a device array of the library's code object
 * (MCX_NOALLOC_MAX_MODULES / 8 = 512 KiB on every GPU the library runs on, held
 * until the process ends) instead of memory allocated at graph load, and the
 * call itself allocates and frees nothing.  Graphs of more than
 * MCX_NOALLOC_MAX_MODULES modules are refused with MCX_E_TOO_LARGE.  Every call
 * on one GPU uses that GPU's array: calls that overlap on one GPU (two streams,
 * two contexts) race; calls on one stream are safe.
 */
#define MCX_NOALLOC_MAX_MODULES (4u << 20)
int mcx_dep_flags_noalloc_dev(mcx_ctx *ctx, const void *d_in_flags, void *d_out_flags,
                              void *d_refcnt);
/*
 * Reachability restricted to LOADED modules.  Bitmaps are ceil(V/32) words,
 * bit v = word v/32, bit v%32.  direction MCX_CLOSURE_HOLDERS: out = seeds'
 * transitive holders (what must be unloaded before the seeds, the situation
 * the reference reports at desktop/ui.py:1679-1680); MCX_CLOSURE_DEPS: out =
 * everything the seeds transitively need.  Seeds that are LOADED are part of
 * the result.  edges_scanned / levels are optional outputs.  There is no limit
 * on the depth of a search.  Unlike the other *_dev entry points, the _dev form
 * does not return before the search is over: its length is only known on the
 * device, so the call enqueues levels a batch at a time and waits for each
 * batch's frontier sizes (the out bitmap is complete when it returns).
 */
int mcx_dep_closure(mcx_ctx *ctx, const uint8_t *in_flags, const uint32_t *seed_bitmap,
                    uint32_t *out_bitmap, int direction,
                    uint64_t *edges_scanned, uint32_t *levels);
int mcx_dep_closure_dev(mcx_ctx *ctx, const void *d_in_flags, const void *d_seed_bitmap,
                        void *d_out_bitmap, int direction,
                        uint64_t *edges_scanned, uint32_t *levels);

/* ---- safe-module filter and the fleet pipeline --------------------------- */

/*
 * Name stages of the reference's is_safe_module (desktop/ui.py:1391-1518):
 * safe_out[i] = 0 when name i is one of the 113 core module names
 * (ui.py:1402-1470) or starts with one of the 56 prefixes (ui.py:1486-1514),
 * else 1.  Names are packed like patterns: name i = names[name_off[i] ..
 * name_off[i+1]).  Host only; needs no context and no device.  The third
 * stage of the filter ("owns a hardware alias") is mcx_alias_hw_flags.
 */
int mcx_safe_module_names(const char *names, const uint32_t *name_off, uint32_t n,
                          uint8_t *safe_out);
int mcx_safe_table_sizes(uint32_t *n_core, uint32_t *n_prefix);

/*
 * The dashboard's "loaded modules" list (desktop/ui.py:1565-1599) for H hosts
 * at once, against the loaded alias table and dependency graph (module ids of
 * the table's a_mod and of the graph are the same ids).
 *
 * Host h owns the devices dev_off[h] .. dev_off[h+1] of the string batch
 * (S, s_off, N; dev_off[H] = N) and the extra loaded modules
 * extra_ids[extra_off[h] .. extra_off[h+1]) (modules loaded by hand or at
 * boot, without a device).  Per host:
 *   used    = { a_mod[first_idx[i]] : i a device of h with a match }
 *             (the drivers the device list shows, ui.py:1567)
 *   loaded  = used + extras, closed under "needs" (what /proc/modules holds
 *             once those modules are in)
 *   held[v] = some loaded u needs v (mc_module_has_holders,
 *             montecarlo/libmontecarlo.c:446-473, asked at ui.py:1575)
 *   listed  = loaded, not used, safe_name[v] and hw_alias[v] (is_safe_module,
 *             ui.py:1571), not held.
 * Output in CSR form: root_off[h] .. root_off[h+1] index root_ids, module ids
 * ascending within a host.  *total_out = root_off[H].  When root_ids (cap
 * entries) is too small the call returns MCX_E_TOO_LARGE with *total_out and
 * root_off filled in, so the caller can allocate and call again.  Module ids
 * >= V in a_mod or extra_ids are ignored.  A graph with a dependency cycle is
 * refused (MCX_E_BAD_ARG).
 * Device memory: 8 * V * ceil(H/32) bytes of scratch; a fleet that does not fit
 * is split by the caller into groups of hosts (hosts are independent).
 * In this mode (used = what the modaliases resolve to) a listed module drives no
 * device the batch knows of and has no holder.  The reference goes on to label
 * every listed module "(In Use)" or "(Idle)" with mc_driver_is_in_use
 * (desktop/ui.py:1579-1590), which looks at ALL device links of the host; the
 * *_labeled entry points below return that byte.
 */
int mcx_fleet_roots(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N,
                    const uint32_t *dev_off, uint32_t H,
                    const uint32_t *extra_off, const uint32_t *extra_ids,
                    const uint8_t *safe_name,
                    uint32_t *root_off, uint32_t *root_ids, uint64_t cap,
                    uint64_t *total_out);
/* Same with device buffers; X = extra_off[H].  Enqueues, then waits once for
 * the total (it decides whether root_ids is large enough). */
int mcx_fleet_roots_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N,
                        const void *d_dev_off, uint32_t H,
                        const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                        const void *d_safe_name,
                        void *d_root_off, void *d_root_ids, uint64_t cap,
                        uint64_t *total_out);

/*
 * The same lists with the reference's own notion of a used driver: the driver
 * BOUND to each listed device (desktop/ui.py:1358-1359 collects the `driver`
 * field of every record, the base name of the sysfs driver link,
 * montecarlo/libmontecarlo.c:400-424), compared byte for byte with the module
 * names (ui.py:1567: no '-' / '_' folding, so a device bound to "virtio-pci" does
 * not make module virtio_pci a used driver).  bound_mod[i] = id of the module
 * named exactly like device i's driver, MCX_NO_MATCH for an unbound device or a
 * driver name that is no module name (mcx_driver_modules builds that map).  Per host:
 *   used   = { bound_mod[i] }
 *   loaded = used + { a_mod[first_idx[i]] } + extras, closed under "needs"
 * and the rest as above.  A device whose modalias resolves to a module that did not
 * bind - the case the reference's brute-force loop looks for, ui.py:1854-1920 -
 * leaves that module loaded and unused: it is listed if nothing holds it.
 */
int mcx_fleet_roots_bound(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N,
                          const uint32_t *dev_off, uint32_t H,
                          const uint32_t *extra_off, const uint32_t *extra_ids,
                          const uint8_t *safe_name, const uint32_t *bound_mod,
                          uint32_t *root_off, uint32_t *root_ids, uint64_t cap,
                          uint64_t *total_out);
int mcx_fleet_roots_bound_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N,
                              const void *d_dev_off, uint32_t H,
                              const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                              const void *d_safe_name, const void *d_bound_mod,
                              void *d_root_off, void *d_root_ids, uint64_t cap,
                              uint64_t *total_out);
/* drv_mod[d] = id of the module whose name equals driver name d byte for byte, else
 * MCX_NO_MATCH; names packed like patterns.  Host only, no context needed. */
int mcx_driver_modules(const char *drv_arena, const uint32_t *drv_off, uint32_t D,
                       const char *mod_names, const uint32_t *mod_off, uint32_t V,
                       uint32_t *drv_mod);

/*
 * The "(In Use)" / "(Idle)" label of the dashboard's module rows.  The reference asks
 * mc_driver_is_in_use(module) for every module it lists (desktop/ui.py:1579-1590).  That
 * function (montecarlo/libmontecarlo.c:573-725) answers 1 when a device link stands under
 * /sys/bus/pci/drivers/<n>, /sys/bus/usb/drivers/<n> or /sys/bus/pcmcia/drivers/<n> for n =
 * the module's name or its name variant (get_driver_names, libmontecarlo.c:538-566:
 * rtw88_X -> rtw_X, else rtlX -> rtw_X), or when the module has holders (a listed module
 * has none).  It looks at every device of the host: also the bridges, hubs and host
 * controllers mc_list_all_devices leaves out and the devices behind the list's limit, none
 * of which reach used_drivers.  So a listed module is "(In Use)" when its driver is bound
 *   - under the variant name (module rtw88_8821cu, driver rtw_8821cu: the names differ, so
 *     ui.py:1567 does not take the module for a used driver),
 *   - only to devices the list does not show.
 *
 * mcx_driver_marks: the map from driver names to the modules they mark, CSR by driver
 * number (mark_off: D + 1 entries; module ids ascending within a driver): driver d marks
 * module m iff d's name equals m's name or m's name variant (the [name_variant] rules of
 * the active tables; names cut at 127 bytes like the reference's buffers; a module without
 * a name marks nothing).  *total_out = mark_off[D]; MCX_E_TOO_LARGE (offsets and total
 * filled in) when mark_ids (cap entries) is too small.  Host only, no context needed.
 */
int mcx_driver_marks(const char *drv_arena, const uint32_t *drv_off, uint32_t D,
                     const char *mod_names, const uint32_t *mod_off, uint32_t V,
                     uint32_t *mark_off, uint32_t *mark_ids, uint64_t cap, uint64_t *total_out);
/*
 * mcx_fleet_roots_bound with labels: host h's marked modules are
 * inuse_ids[inuse_off[h] .. inuse_off[h+1]) - for every device of the host on the pci, usb
 * or pcmcia bus that has a driver bound, listed or not, the modules mcx_driver_marks names
 * for that driver (repeats allowed, ids >= V ignored).  root_in_use[k] = 1 iff module
 * root_ids[k] is marked on its host, else 0; needs cap bytes.  Y = inuse_off[H].
 */
int mcx_fleet_roots_labeled(mcx_ctx *ctx, const char *S, const uint32_t *s_off, uint32_t N,
                            const uint32_t *dev_off, uint32_t H,
                            const uint32_t *extra_off, const uint32_t *extra_ids,
                            const uint8_t *safe_name, const uint32_t *bound_mod,
                            const uint32_t *inuse_off, const uint32_t *inuse_ids,
                            uint32_t *root_off, uint32_t *root_ids, uint8_t *root_in_use, uint64_t cap,
                            uint64_t *total_out);
int mcx_fleet_roots_labeled_dev(mcx_ctx *ctx, const void *d_S, const void *d_s_off, uint32_t N,
                                const void *d_dev_off, uint32_t H,
                                const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                                const void *d_safe_name, const void *d_bound_mod,
                                const void *d_inuse_off, const void *d_inuse_ids, uint32_t Y,
                                void *d_root_off, void *d_root_ids, void *d_root_in_use, uint64_t cap,
                                uint64_t *total_out);

/* ---- fleet inventory: which devices a host would list --------------------- */

/*
 * mc_list_all_devices (montecarlo/libmontecarlo.c:233-440) walks one host's
 * sysfs and drops bridges, hubs, host controllers and SCSI scaffolding
 * (mc_is_infrastructure_device, montecarlo/libmontecarlo.c:729-823) before a
 * device reaches the list; mc_is_excluded_device
 * (montecarlo/libmontecarlo.c:826-874) marks USB mass storage.  For an
 * inventory of many hosts the same tests run over packed records, one per
 * device, holding what those functions read from sysfs.
 */
#define MCX_SUB_USB 0u     /* subsystem, the five of libmontecarlo.c:247 */
#define MCX_SUB_PCI 1u
#define MCX_SUB_HID 2u
#define MCX_SUB_SCSI 3u
#define MCX_SUB_PCMCIA 4u
#define MCX_SUB_OTHER 255u /* anything else: never enumerated */

#define MCX_DT_NONE 0u /* udev devtype */
#define MCX_DT_USB_DEVICE 1u
#define MCX_DT_USB_INTERFACE 2u
#define MCX_DT_SCSI_HOST 3u
#define MCX_DT_SCSI_TARGET 4u
#define MCX_DT_SCSI_GENERIC 5u
#define MCX_DT_OTHER 6u /* e.g. scsi_device */

#define MCX_DA_PCI_CLASS 0x01u    /* sysfs "class" is readable: pci_class is valid  */
#define MCX_DA_IFACE_CLASS 0x02u  /* bInterfaceClass readable: iface_class valid    */
#define MCX_DA_PARENT 0x04u       /* has a usb / usb_device parent                  */
#define MCX_DA_PARENT_CLASS 0x08u /* the parent's bDeviceClass: parent_class valid  */
#define MCX_DA_OWN_CLASS 0x10u    /* the node's own bDeviceClass: own_class valid   */
#define MCX_DA_SCSI_MODEL 0x20u   /* sysfs "model" readable                         */
#define MCX_DA_SCSI_VENDOR 0x40u  /* sysfs "vendor" readable                        */

#define MCX_DRV_NONE 0xFFFFFFFFu    /* no driver link: the list shows "None"         */
#define MCX_DRV_NOSLASH 0xFFFFFFFEu /* a link whose target holds no '/'              */

typedef struct mcx_dev_rec {
    uint8_t subsystem;    /* MCX_SUB_*                                              */
    uint8_t devtype;      /* MCX_DT_*                                               */
    uint8_t attrs;        /* MCX_DA_* presence bits                                 */
    uint8_t iface_class;  /* bInterfaceClass, the value of its two hex digits       */
    uint32_t pci_class;   /* the 24-bit class code of sysfs "class" (0xBBSSII)      */
    uint8_t parent_class; /* bDeviceClass of the usb_device parent                  */
    uint8_t own_class;    /* bDeviceClass of the node itself                        */
    uint16_t reserved;
    uint32_t driver;      /* index into the driver-name table, or MCX_DRV_*         */
} mcx_dev_rec_t;          /* 16 bytes */

/* verdict byte per device */
#define MCX_DV_LISTED 0x01u   /* mc_list_all_devices would put it in the list       */
#define MCX_DV_EXCLUDED 0x02u /* mc_is_excluded_device: USB mass storage            */
#define MCX_DV_REASON(v) ((v) >> 4) /* why a device is not listed: MCX_DR_*         */
#define MCX_DR_NONE 0u
#define MCX_DR_SUBSYSTEM 1u   /* not one of the five subsystems (libmontecarlo.c:393-398) */
#define MCX_DR_PCI_CLASS 2u   /* bridge, SMBus, system peripheral (:757-762)         */
#define MCX_DR_PCI_DRIVER 3u  /* one of the 8 port / hot-plug drivers (:773-785)     */
#define MCX_DR_SCSI 4u        /* host, target, generic, or nameless (:790-818)       */
#define MCX_DR_USB_NODE 5u    /* not a usb_interface, or without parent (:293-314)   */
#define MCX_DR_USB_HUB 6u     /* interface or parent class 09 (:301-321)             */
#define MCX_DR_USB_HOST 7u    /* driver name holds "hcd" or is "hub" (:412-417)      */
#define MCX_DR_OVER_MAX 8u    /* the host's list was already full (:259-260)         */

/*
 * verdict[i] for N device records of H hosts (host h owns records
 * dev_off[h] .. dev_off[h+1]; the order within a host is the enumeration
 * order).  Driver names are packed like patterns (drv_arena, drv_off, D names).
 * max_per_host = the `max` argument of mc_list_all_devices (the reference's UI
 * passes 64, desktop/ui.py:1335; 0 = no limit): only the first max_per_host
 * devices of a host that pass are listed.  kept_off (H + 1 entries, optional):
 * prefix sums of the listed devices per host.  The *_dev form takes device
 * pointers, enqueues, and waits once for *kept_total.
 */
int mcx_device_filter(mcx_ctx *ctx, const mcx_dev_rec_t *rec, uint32_t N,
                      const char *drv_arena, const uint32_t *drv_off, uint32_t D,
                      const uint32_t *dev_off, uint32_t H, uint32_t max_per_host,
                      uint8_t *verdict, uint32_t *kept_off);
int mcx_device_filter_dev(mcx_ctx *ctx, const void *d_rec, uint32_t N,
                          const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                          const void *d_dev_off, uint32_t H, uint32_t max_per_host,
                          void *d_verdict, void *d_kept_off, uint64_t *kept_total);
/*
 * The listed devices' modalias strings, packed in order, ready for
 * mcx_alias_match_dev / mcx_fleet_roots_dev with d_kept_off as their dev_off:
 * string k of the output is string i of the input for the k-th i with
 * MCX_DV_LISTED set.  d_S_out needs s_off[N] bytes, d_s_off_out kept_total + 1
 * entries, d_src (optional) kept_total entries and receives the i's.
 * All pointers are device pointers; d_verdict and d_kept_off are the outputs
 * of mcx_device_filter_dev for the same records.
 */
int mcx_fleet_compact_dev(mcx_ctx *ctx, const void *d_verdict, const void *d_S, const void *d_s_off,
                          uint32_t N, const void *d_dev_off, uint32_t H, const void *d_kept_off,
                          uint32_t kept_total, void *d_S_out, void *d_s_off_out, void *d_src,
                          uint64_t *bytes_out);

/*
 * The whole of configs "inventory -> module lists" in one call, device buffers:
 * mcx_device_filter_dev (verdicts are left in d_verdict), the packing of the
 * listed devices' strings, and mcx_fleet_roots_dev over them; arguments as
 * in those three.  The packed strings live in a buffer the context keeps.
 */
int mcx_fleet_inventory_roots_dev(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off,
                                  uint32_t N, const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                                  const void *d_dev_off, uint32_t H, uint32_t max_per_host,
                                  const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                                  const void *d_safe_name, void *d_verdict,
                                  void *d_root_off, void *d_root_ids, uint64_t cap, uint64_t *total_out);

/* The same with the used drivers taken from the records' bound drivers
 * (mcx_fleet_roots_bound_dev): d_drv_mod holds D module ids, the output of
 * mcx_driver_modules for the driver-name table, on the device. */
int mcx_fleet_inventory_roots_bound_dev(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off,
                                        uint32_t N, const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                                        const void *d_drv_mod,
                                        const void *d_dev_off, uint32_t H, uint32_t max_per_host,
                                        const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                                        const void *d_safe_name, void *d_verdict,
                                        void *d_root_off, void *d_root_ids, uint64_t cap, uint64_t *total_out);

/* The same with the "(In Use)" byte of every listed module (see mcx_driver_marks): the marks
 * come from the records themselves - every record of a host on the pci, usb or pcmcia bus
 * with a driver bound, whether the device list shows it or not - through d_mark_off /
 * d_mark_ids, the output of mcx_driver_marks for the driver-name table, on the device. */
int mcx_fleet_inventory_roots_labeled_dev(mcx_ctx *ctx, const void *d_rec, const void *d_S, const void *d_s_off,
                                          uint32_t N, const void *d_drv_arena, const void *d_drv_off, uint32_t D,
                                          const void *d_drv_mod, const void *d_mark_off, const void *d_mark_ids,
                                          const void *d_dev_off, uint32_t H, uint32_t max_per_host,
                                          const void *d_extra_off, const void *d_extra_ids, uint32_t X,
                                          const void *d_safe_name, void *d_verdict,
                                          void *d_root_off, void *d_root_ids, void *d_root_in_use, uint64_t cap,
                                          uint64_t *total_out);

/* ---- the library's name tables as data ------------------------------------ */

/*
 * The literals the reference spreads over its sources - the 113 core module
 * names and 56 prefixes of is_safe_module (desktop/ui.py:1402-1470, 1486-1514),
 * the 6 bus tags of a hardware alias (desktop/ui.py:1540), the 8 PCI port
 * driver names (montecarlo/libmontecarlo.c:778-780), the rtw88 / rtl name
 * variants (montecarlo/libmontecarlo.c:552-565) - are one table set here.
 * The compiled-in defaults are the reference's values; a text file
 * ("format 1", sections [core] [prefix] [hw_alias_prefix] [infra_driver]
 * [name_variant], one entry per line, '#' comments) replaces them for the
 * whole process: mcx_safe_module_names, the hardware-alias flags of tables
 * loaded afterwards, mcx_device_filter, mc_is_infrastructure_device and
 * mc_driver_is_in_use follow the active set.  Replace it before other
 * threads use the library.  mcx_tables_dump writes the active set in the file
 * format (buf may be NULL to learn the size); counts: core, prefix,
 * hw_alias_prefix, infra_driver, name_variant rules.
 */
int mcx_tables_load_file(const char *path);
int mcx_tables_load_text(const char *text, size_t len);
void mcx_tables_reset(void);
int mcx_tables_dump(char *buf, size_t cap, size_t *need);
int mcx_tables_info(uint32_t *format, uint32_t counts[5], int *is_default);

/* ---- text tables: modules.alias, modules.dep, /proc/modules ------------- */

/*
 * The reference leaves these files to kmod (modprobe at
 * montecarlo/libmontecarlo.c:123, `modinfo -F alias` at desktop/ui.py:1527)
 * and reads /proc/modules token by token
 * (montecarlo/libmontecarlo.c:499-532).  A mcx_modtab parses their text into
 * the arrays the entry points above take.  Module ids are handed out in order
 * of first appearance, over all texts added to one table; a module's name is
 * the base name of its file up to the first '.', with '-' written as '_'
 * (the rule at desktop/ui.py:481-482).  Host only, no context needed.
 */
typedef struct mcx_modtab mcx_modtab;

mcx_modtab *mcx_modtab_create(void);
void mcx_modtab_destroy(mcx_modtab *t);
const char *mcx_modtab_last_error(mcx_modtab *t);
/* "alias <glob> <module>" lines; '#' comments, blank lines and lines with
 * other keywords are skipped.  Patterns keep file order. */
int mcx_modtab_add_alias_text(mcx_modtab *t, const char *text, size_t len);
/* "<path>.ko[.xz|.zst|.gz]: <dep path> ..." lines: everything the module needs,
 * directly or not, as depmod writes it. */
int mcx_modtab_add_dep_text(mcx_modtab *t, const char *text, size_t len);
/* both files of one /lib/modules/<version> directory */
int mcx_modtab_load_dir(mcx_modtab *t, const char *dir);
int mcx_modtab_counts(mcx_modtab *t, uint32_t *patterns, uint32_t *modules, uint64_t *alias_bytes);
/* the alias table as mcx_alias_table_load takes it; pointers stay valid until
 * the table is added to or destroyed */
int mcx_modtab_alias_arrays(mcx_modtab *t, const char **arena, const uint32_t **a_off,
                            const uint32_t **a_mod);
/* dependency CSR over module ids, rows sorted.  direct = 0: the lists as the
 * file has them (transitive); direct = 1: without the entries another entry of
 * the same list already needs (the transitive reduction), which is what
 * mcx_dep_graph_load wants.  /sys/module/<m>/holders lists symbol-level users:
 * a module that uses both B and C, where B uses C too, holds C there but has no
 * edge to C here.  The flag bits of mcx_dep_flags are the same either way (a
 * loaded holder implies the loaded chain below it); its reference counts and
 * the closure's edges_scanned are lower bounds of what sysfs would count. */
int mcx_modtab_dep_arrays(mcx_modtab *t, int direct, const uint32_t **rowptr, const uint32_t **col,
                          uint32_t *edges);
const char *mcx_modtab_module_name(mcx_modtab *t, uint32_t id);   /* NULL if out of range */
uint32_t mcx_modtab_module_id(mcx_modtab *t, const char *name);    /* MCX_NO_MATCH if unknown */
/* all names packed for mcx_safe_module_names: name_off needs modules + 1
 * entries; call with names = NULL to get the size in *bytes_out */
int mcx_modtab_module_names(mcx_modtab *t, char *names, uint64_t cap, uint32_t *name_off,
                            uint64_t *bytes_out);
/* /proc/modules text: in_flags[id] |= MCX_F_LOADED for every first token that
 * names a module of the table; *unknown counts the ones that do not */
int mcx_modtab_mark_loaded(mcx_modtab *t, const char *proc_modules, size_t len, uint8_t *in_flags,
                           uint32_t *unknown);
/* mcx_alias_table_load / mcx_dep_graph_load (direct edges) from a table */
int mcx_alias_table_load_modtab(mcx_ctx *ctx, mcx_modtab *t, uint32_t flags);
int mcx_dep_graph_load_modtab(mcx_ctx *ctx, mcx_modtab *t);

#ifdef __cplusplus
}
#endif
#endif /* MCX_H */
