/*
 * mcx_oracle.c - CPU oracle for the batch hot path.  TEST INFRASTRUCTURE ONLY:
 * nothing under oracle/ is linked, imported or called by the product library;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * as the checker.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference holds no
 * tests, fixtures or golden vectors for this path, and it computes neither the
 * alias match nor the dependency closure itself: it shells out to kmod
 * (`modprobe`, reference montecarlo/libmontecarlo.c:123,133 and
 * montecarlo/helper.c:76-82; `modinfo -F alias`, reference
 * desktop/ui.py:1526-1531).  kmod is a third-party dependency absent from the
 * reference tree (Debian/Ubuntu package kmod 29 in this image); it decides an
 * alias match with glibc fnmatch(alias_pattern, modalias, 0) after mapping '-'
 * to '_' outside brackets.  The oracle therefore
 *   - K1: calls glibc's own fnmatch(3) (flags 0, C locale forced), i.e. the
 *     function kmod executes, not a re-derivation of it; tests/golden holds
 *     outputs of kmod 29's `modprobe -R` produced in this image for a spot
 *     check of table order and multi-match behaviour;
 *   - K2: restates what the reference reads from sysfs one module at a time,
 *     mc_module_has_holders (reference montecarlo/libmontecarlo.c:446-473),
 *     mc_driver_is_in_use (573-725), mc_get_module_refcount (476-495) and the
 *     root-module loop (reference desktop/ui.py:1565-1599), over a CSR graph,
 *     plus a queue BFS for the closure;
 *   - safe-module name filter: restates the tables of is_safe_module
 *     (reference desktop/ui.py:1391-1518) literally.
 */
#define _GNU_SOURCE
#include <fnmatch.h>
#include <locale.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NO_MATCH 0xFFFFFFFFu

/* ---- K1 ------------------------------------------------------------------ */

typedef struct {
    locale_t c_loc, old;
} orc_locale_t;

static void orc_enter_c_locale(orc_locale_t *l)
{
    l->c_loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    l->old = l->c_loc ? uselocale(l->c_loc) : (locale_t)0;
}

static void orc_leave_c_locale(orc_locale_t *l)
{
    if (l->c_loc) {
        uselocale(l->old);
        freelocale(l->c_loc);
    }
}

/* One pair, for differential tests: fnmatch(pattern, string, 0) in the C
 * locale; 1 = match, 0 = no match. */
int orc_fnmatch(const char *pattern, const char *string)
{
    orc_locale_t l;
    orc_enter_c_locale(&l);
    int r = fnmatch(pattern, string, 0) == 0;
    orc_leave_c_locale(&l);
    return r;
}

/*
 * strings i in [i0, i1) of the batch against all P patterns.
 * first_idx[i] = smallest matching pattern index or ORC_NO_MATCH, count[i] =
 * number of matching patterns.  normalize_dash reproduces kmod's alias
 * normalisation on both sides ('-' -> '_' outside [...]).
 * Returns the number of fnmatch calls made.
 */
/* kmod's rule (libkmod-util.c, alias_normalize / underscores): '-' becomes
 * '_' outside [...]; an unmatched bracket is an error, and kmod then drops
 * the alias line or refuses the query.  Returns 0 on such an error. */
static int orc_normalize(char *s)
{
    for (size_t i = 0; s[i]; ++i) {
        if (s[i] == '-') {
            s[i] = '_';
        } else if (s[i] == ']') {
            return 0;
        } else if (s[i] == '[') {
            while (s[i] && s[i] != ']')
                ++i;
            if (!s[i])
                return 0;
        }
    }
    return 1;
}

uint64_t orc_alias_match(const char *arena, const uint32_t *a_off, uint32_t P,
                         const char *S, const uint32_t *s_off, uint32_t i0, uint32_t i1,
                         int normalize_dash, uint32_t *first_idx, uint32_t *count)
{
    /* NUL-terminated copies: fnmatch takes C strings */
    size_t abytes = a_off[P] + (size_t)P + 1;
    char *pat = (char *)malloc(abytes);
    size_t *pstart = (size_t *)malloc(sizeof(size_t) * ((size_t)P + 1));
    size_t w = 0;
    for (uint32_t j = 0; j < P; ++j) {
        size_t n = a_off[j + 1] - a_off[j];
        pstart[j] = w;
        memcpy(pat + w, arena + a_off[j], n);
        pat[w + n] = 0;
        if (normalize_dash && !orc_normalize(pat + w))
            pstart[j] = (size_t)-1; /* kmod ignores such a line */
        w += n + 1;
    }
    size_t maxlen = 0;
    for (uint32_t i = i0; i < i1; ++i)
        if (s_off[i + 1] - s_off[i] > maxlen)
            maxlen = s_off[i + 1] - s_off[i];
    char *str = (char *)malloc(maxlen + 1);
    orc_locale_t l;
    orc_enter_c_locale(&l);
    uint64_t calls = 0;
    for (uint32_t i = i0; i < i1; ++i) {
        size_t n = s_off[i + 1] - s_off[i];
        memcpy(str, S + s_off[i], n);
        str[n] = 0;
        int ok = 1;
        if (normalize_dash)
            ok = orc_normalize(str);
        uint32_t first = ORC_NO_MATCH, cnt = 0;
        for (uint32_t j = 0; ok && j < P; ++j) {
            if (pstart[j] == (size_t)-1)
                continue;
            if (fnmatch(pat + pstart[j], str, 0) == 0) {
                if (first == ORC_NO_MATCH)
                    first = j;
                ++cnt;
            }
        }
        calls += P;
        first_idx[i] = first;
        count[i] = cnt;
    }
    orc_leave_c_locale(&l);
    free(str);
    free(pstart);
    free(pat);
    return calls;
}

/* ---- K2 ------------------------------------------------------------------ */

#define F_LOADED 1u
#define F_BOUND 2u
#define F_SAFE 4u
#define O_HAS_HOLDERS 1u
#define O_IN_USE 2u
#define O_ROOT 4u

/*
 * rowptr/col: u -> col[k] means "u needs col[k]" (col[k] has holder u).
 * HAS_HOLDERS follows mc_module_has_holders: /sys/module/<v>/holders lists the
 * loaded modules that use v (reference libmontecarlo.c:446-473); IN_USE
 * follows mc_driver_is_in_use: a bound device, else holders (573-725); ROOT is
 * the survivor of the dashboard loop: loaded, passes is_safe_module and is not
 * a used driver (folded into SAFE), has no holders (ui.py:1565-1577).
 * refcnt mirrors /sys/module/<v>/refcnt read by mc_get_module_refcount
 * (476-495): holders + bound devices, -1 (0xFFFFFFFF) when absent.
 */
void orc_dep_flags(const uint32_t *rowptr, const uint32_t *col, uint32_t V,
                   const uint8_t *in_flags, uint8_t *out_flags, uint32_t *refcnt)
{
    uint32_t *holders = (uint32_t *)calloc(V ? V : 1, sizeof(uint32_t));
    for (uint32_t u = 0; u < V; ++u) {
        if (!(in_flags[u] & F_LOADED))
            continue;
        for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
            uint32_t v = col[k];
            if (in_flags[v] & F_LOADED)
                holders[v]++;
        }
    }
    for (uint32_t v = 0; v < V; ++v) {
        uint8_t f = in_flags[v], o = 0;
        if (f & F_LOADED) {
            if (holders[v])
                o |= O_HAS_HOLDERS;
            if ((f & F_BOUND) || holders[v])
                o |= O_IN_USE;
            if ((f & F_SAFE) && !holders[v])
                o |= O_ROOT;
        }
        out_flags[v] = o;
        if (refcnt)
            refcnt[v] = (f & F_LOADED) ? holders[v] + ((f & F_BOUND) ? 1u : 0u) : 0xFFFFFFFFu;
    }
    free(holders);
}

/*
 * The same flags from several threads (the edge-parallel form of the CPU
 * baseline).  Counting into the held module from the holder's side needs an
 * atomic increment per edge, and the few modules everything depends on make
 * those collide (measured: slower than one thread); so this form pulls: given
 * the graph of holders (rrow/rcol: v is held by rcol[rrow[v] .. rrow[v+1]),
 * built once like the device's copy), every caller owns the modules [v0, v1).
 */
void orc_dep_flags_pull(const uint32_t *rrow, const uint32_t *rcol, const uint8_t *in_flags,
                        uint8_t *out_flags, uint32_t *refcnt, uint32_t v0, uint32_t v1)
{
    for (uint32_t v = v0; v < v1; ++v) {
        uint8_t f = in_flags[v], o = 0;
        uint32_t holders = 0;
        if (f & F_LOADED) {
            for (uint32_t k = rrow[v]; k < rrow[v + 1]; ++k)
                holders += in_flags[rcol[k]] & F_LOADED;
            if (holders)
                o |= O_HAS_HOLDERS;
            if ((f & F_BOUND) || holders)
                o |= O_IN_USE;
            if ((f & F_SAFE) && !holders)
                o |= O_ROOT;
        }
        out_flags[v] = o;
        if (refcnt)
            refcnt[v] = (f & F_LOADED) ? holders + ((f & F_BOUND) ? 1u : 0u) : 0xFFFFFFFFu;
    }
}

/*
 * direction 0: out = loaded modules from which a loaded seed is reachable
 * along "needs" edges through loaded modules only (the modules that must go
 * before the seeds can be unloaded, the condition the reference reports at
 * desktop/ui.py:1679-1680); direction 1: out = loaded modules reachable from
 * a loaded seed.  Bitmaps: bit v = word v/32, bit v%32.
 * Returns edges scanned; *levels = BFS depth (number of frontier expansions
 * that found at least one vertex to expand).
 */
uint64_t orc_dep_closure(const uint32_t *rowptr, const uint32_t *col, uint32_t V, uint32_t E,
                         const uint8_t *in_flags, const uint32_t *seed, uint32_t *out,
                         int direction, uint32_t *levels)
{
    const uint32_t *rp = rowptr, *cl = col;
    uint32_t *rrow = NULL, *rcol = NULL;
    if (direction == 0) {
        /* reverse CSR by counting sort */
        rrow = (uint32_t *)calloc((size_t)V + 2, sizeof(uint32_t));
        rcol = (uint32_t *)malloc(sizeof(uint32_t) * (E ? E : 1));
        for (uint32_t k = 0; k < E; ++k)
            rrow[col[k] + 2]++;
        for (uint32_t v = 0; v < V; ++v)
            rrow[v + 2] += rrow[v + 1];
        for (uint32_t u = 0; u < V; ++u)
            for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; ++k)
                rcol[rrow[col[k] + 1]++] = u;
        rp = rrow;
        cl = rcol;
    }
    uint32_t W = (V + 31) / 32;
    memset(out, 0, sizeof(uint32_t) * W);
    uint32_t *queue = (uint32_t *)malloc(sizeof(uint32_t) * (V ? V : 1));
    uint32_t head = 0, tail = 0;
    for (uint32_t v = 0; v < V; ++v)
        if ((seed[v >> 5] >> (v & 31) & 1u) && (in_flags[v] & F_LOADED)) {
            out[v >> 5] |= 1u << (v & 31);
            queue[tail++] = v;
        }
    uint64_t scanned = 0;
    uint32_t lv = 0;
    while (head < tail) {
        uint32_t level_end = tail;
        ++lv;
        while (head < level_end) {
            uint32_t v = queue[head++];
            for (uint32_t k = rp[v]; k < rp[v + 1]; ++k) {
                uint32_t u = cl[k];
                ++scanned;
                if (!(in_flags[u] & F_LOADED))
                    continue;
                if (out[u >> 5] >> (u & 31) & 1u)
                    continue;
                out[u >> 5] |= 1u << (u & 31);
                queue[tail++] = u;
            }
        }
    }
    if (levels)
        *levels = lv;
    free(queue);
    free(rrow);
    free(rcol);
    return scanned;
}

/* ---- safe-module name filter ---------------------------------------------
 * Literal restatement of the first two stages of is_safe_module
 * (reference desktop/ui.py:1391-1518): exact-name sets, then bare
 * startswith() over the pattern list in list order.  Returns 1 when the name
 * survives both stages (the modinfo hardware-alias stage is a separate
 * input). */
static const char *const ORC_CORE[] = {
    /* kernel_core, ui.py:1402-1408 */
    "cpuid", "msr", "acpi_pad", "acpi_cpufreq", "acpi_thermal", "dmi_sysfs", "dmi_notifier",
    "efi_pstore", "efivars", "efivarfs", "pstore", "pstore_blk", "pstore_ram",
    /* memory_block, ui.py:1411-1414 */
    "zram", "zsmalloc", "loop", "nbd",
    /* raid_dm, ui.py:1417-1421 */
    "raid0", "raid1", "raid10", "raid456", "raid6_pq", "dm_mod", "dm_crypt", "dm_mirror",
    "dm_snapshot", "md_mod", "linear", "multipath",
    /* filesystems, ui.py:1424-1432 */
    "ext4", "ext3", "ext2", "jbd2", "mbcache", "btrfs", "xfs", "jfs", "reiserfs", "minix",
    "vfat", "fat", "msdos", "ntfs", "ntfs3", "fuse", "fuseblk", "overlayfs", "squashfs",
    "iso9660", "udf", "nfs", "nfsd", "lockd", "exportfs", "cifs", "smb", "smbfs",
    /* sound_core, ui.py:1435-1442 */
    "soundcore", "snd", "snd_seq", "snd_seq_device", "snd_seq_midi", "snd_seq_midi_event",
    "snd_timer", "snd_pcm", "snd_rawmidi", "snd_hwdep", "snd_hda_core", "snd_hda_codec",
    "snd_hda_codec_generic", "snd_hda_intel",
    /* hid_input_core, ui.py:1445-1450 */
    "hid", "hid_generic", "uhid", "hidp", "usbhid", "usbkbd", "usbmouse", "joydev", "evdev",
    "mousedev", "input_leds", "led_class",
    /* virtualization, ui.py:1453-1459 */
    "kvm", "kvm_intel", "kvm_amd", "vboxdrv", "vboxnetflt", "vboxnetadp", "vboxpci",
    "vmw_balloon", "vmw_vmci", "vmw_vsock_vmci_transport", "virtio", "virtio_pci",
    "virtio_balloon", "virtio_blk", "virtio_net", "vhost", "vhost_net", "vhost_vsock",
    /* network_core, ui.py:1462-1465 */
    "bridge", "stp", "llc", "bonding", "8021q", "veth", "tun", "tap",
    /* legacy_subsystems, ui.py:1468-1470 */
    "parport", "parport_pc", "ppdev", "lp",
    NULL};

/* dangerous_patterns, ui.py:1486-1514, list order */
static const char *const ORC_PREFIX[] = {
    "xt_", "nf_", "nft_", "ip_", "ip6_", "ipt_", "ip6t_", "nfnetlink", "netfilter", "conntrack",
    "crypto_", "sha", "aes", "ghash", "crc32", "md5", "des", "ecb", "cbc", "gcm", "ccm", "ctr",
    "k10temp", "coretemp", "ssse3", "aesni", "cpu_", "cpufreq", "intel_", "amd_", "x86_pkg_temp",
    "acpi", "battery", "ac", "button", "fan", "thermal",
    "i2c_", "spi_", "smbus",
    "pcieport", "pci_bridge", "shpchp",
    "scsi_mod", "sd_mod", "sr_mod", "sg", "st",
    "nvme_core",
    "videodev", "videobuf", "v4l2_common",
    "snd_",
    "hid_",
    "vbox", "vmw_",
    NULL};

int orc_core_count(void)
{
    int n = 0;
    while (ORC_CORE[n])
        ++n;
    return n;
}

int orc_prefix_count(void)
{
    int n = 0;
    while (ORC_PREFIX[n])
        ++n;
    return n;
}

int orc_safe_name(const char *mod)
{
    for (int i = 0; ORC_CORE[i]; ++i)
        if (strcmp(mod, ORC_CORE[i]) == 0)
            return 0;
    for (int i = 0; ORC_PREFIX[i]; ++i)
        if (strncmp(mod, ORC_PREFIX[i], strlen(ORC_PREFIX[i])) == 0)
            return 0;
    return 1;
}

/* hardware-alias stage, ui.py:1537-1553: any line of the alias text, stripped,
 * starts with one of the six bus tags.  `aliases` is the text modinfo would
 * print (one alias per line); empty text means "no aliases" (ui.py:1533). */
int orc_has_hw_alias(const char *aliases)
{
    static const char *const tags[] = {"pci:", "usb:", "platform:", "hid:", "serio:", "of:", NULL};
    const char *p = aliases;
    while (*p) {
        const char *e = strchr(p, '\n');
        size_t n = e ? (size_t)(e - p) : strlen(p);
        size_t b = 0;
        while (b < n && (p[b] == ' ' || p[b] == '\t' || p[b] == '\r' || p[b] == '\v' || p[b] == '\f'))
            ++b;
        for (int t = 0; tags[t]; ++t) {
            size_t tl = strlen(tags[t]);
            if (n - b >= tl && memcmp(p + b, tags[t], tl) == 0)
                return 1;
        }
        if (!e)
            break;
        p = e + 1;
    }
    return 0;
}

/* ---- fleet inventory: the dashboard's module list, one host after another --
 * For every host the loop of reference desktop/ui.py:1565-1599 is restated over
 * tables instead of sysfs:
 *   used     drivers bound to the host's devices (ui.py builds used_drivers
 *            from mc_list_all_usb_devices records); here the module a device's
 *            modalias resolves to (driver_mod[i], 0xFFFFFFFF = none),
 *   loaded   /proc/modules of the host: the used drivers and the host's extra
 *            modules, closed under "needs" (loading a module loads what it
 *            needs, so /proc/modules is always closed),
 *   holders  /sys/module/<v>/holders: loaded modules that need v
 *            (libmontecarlo.c:446-473).
 * A module is listed when it is loaded, is not a used driver (ui.py:1567),
 * passes is_safe_module - name stages safe_name[v], hardware-alias stage
 * hw_alias[v] (ui.py:1571, 1391-1562) - and has no holders (ui.py:1575-1577).
 * Output: root_off[h]..root_off[h+1] index root_ids, ascending module ids per
 * host.  root_ids may be NULL to count only.  Returns the total.
 */
static int orc_cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* bound_mod == NULL: the used drivers are the modules the devices resolve to (above).
 * Otherwise they are what reference desktop/ui.py:1358-1359 collects: the driver bound
 * to each listed device, bound_mod[i] = the module named exactly like it (0xFFFFFFFF:
 * unbound, or a driver name that is no module name - ui.py:1567 compares the strings as
 * they are); the module a device resolves to is then loaded but not used. */
static uint64_t fleet_roots_impl(const uint32_t *rowptr, const uint32_t *col, uint32_t V,
                                 const uint32_t *driver_mod, const uint32_t *bound_mod,
                                 const uint32_t *dev_off, uint32_t H,
                                 const uint32_t *extra_off, const uint32_t *extra_ids,
                                 const uint8_t *safe_name, const uint8_t *hw_alias,
                                 uint32_t *root_off, uint32_t *root_ids)
{
    /* per-host marks are stamped with h + 1 so nothing is cleared between hosts */
    uint32_t *used = (uint32_t *)calloc(V ? V : 1, sizeof(uint32_t));
    uint32_t *loaded = (uint32_t *)calloc(V ? V : 1, sizeof(uint32_t));
    uint32_t *held = (uint32_t *)calloc(V ? V : 1, sizeof(uint32_t));
    uint32_t *list = (uint32_t *)malloc(sizeof(uint32_t) * (V ? V : 1));
    uint64_t total = 0;
    for (uint32_t h = 0; h < H; ++h) {
        const uint32_t stamp = h + 1;
        uint32_t n = 0;
        for (uint32_t i = dev_off[h]; i < dev_off[h + 1]; ++i) {
            uint32_t m = driver_mod[i];
            if (m < V) {
                if (!bound_mod)
                    used[m] = stamp;
                if (loaded[m] != stamp) {
                    loaded[m] = stamp;
                    list[n++] = m;
                }
            }
            uint32_t b = bound_mod ? bound_mod[i] : 0xFFFFFFFFu;
            if (b < V) {
                used[b] = stamp;
                if (loaded[b] != stamp) {
                    loaded[b] = stamp;
                    list[n++] = b;
                }
            }
        }
        for (uint32_t k = extra_off[h]; k < extra_off[h + 1]; ++k) {
            uint32_t m = extra_ids[k];
            if (m < V && loaded[m] != stamp) {
                loaded[m] = stamp;
                list[n++] = m;
            }
        }
        /* close under "needs"; list doubles as the work queue */
        for (uint32_t q = 0; q < n; ++q) {
            uint32_t u = list[q];
            for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
                uint32_t v = col[k];
                held[v] = stamp;
                if (loaded[v] != stamp) {
                    loaded[v] = stamp;
                    list[n++] = v;
                }
            }
        }
        root_off[h] = (uint32_t)total;
        qsort(list, n, sizeof(uint32_t), orc_cmp_u32);
        for (uint32_t q = 0; q < n; ++q) {
            uint32_t v = list[q];
            if (used[v] == stamp || !safe_name[v] || !hw_alias[v] || held[v] == stamp)
                continue;
            if (root_ids)
                root_ids[total] = v;
            ++total;
        }
    }
    root_off[H] = (uint32_t)total;
    free(used);
    free(loaded);
    free(held);
    free(list);
    return total;
}

uint64_t orc_fleet_roots(const uint32_t *rowptr, const uint32_t *col, uint32_t V,
                         const uint32_t *driver_mod, const uint32_t *dev_off, uint32_t H,
                         const uint32_t *extra_off, const uint32_t *extra_ids,
                         const uint8_t *safe_name, const uint8_t *hw_alias,
                         uint32_t *root_off, uint32_t *root_ids)
{
    return fleet_roots_impl(rowptr, col, V, driver_mod, NULL, dev_off, H, extra_off, extra_ids, safe_name, hw_alias,
                            root_off, root_ids);
}

uint64_t orc_fleet_roots_bound(const uint32_t *rowptr, const uint32_t *col, uint32_t V,
                               const uint32_t *driver_mod, const uint32_t *bound_mod,
                               const uint32_t *dev_off, uint32_t H,
                               const uint32_t *extra_off, const uint32_t *extra_ids,
                               const uint8_t *safe_name, const uint8_t *hw_alias,
                               uint32_t *root_off, uint32_t *root_ids)
{
    return fleet_roots_impl(rowptr, col, V, driver_mod, bound_mod, dev_off, H, extra_off, extra_ids, safe_name,
                            hw_alias, root_off, root_ids);
}

/* ---- "(In Use)" / "(Idle)" of the dashboard's module rows ---------------------
 * reference desktop/ui.py:1579-1590: for every module it lists the dashboard asks
 * mc_driver_is_in_use(mod).  Restated from montecarlo/libmontecarlo.c:538-566
 * (get_driver_names) and :573-725 over records instead of sysfs: the directory
 * /sys/bus/<bus>/drivers/<name> holds a device link exactly when some device of the host
 * sits on <bus> with <name> bound, whatever the device list makes of that device.  The
 * holders part (:699-722) cannot fire for a listed module (ui.py:1575-1577 dropped those).
 */
static int orc_names_of_module(const char *module_name, char names[][128], int max_names)
{
    int count = 0;
    if (!module_name || count >= max_names)
        return 0;
    strncpy(names[count], module_name, 127); /* :548-550 */
    names[count][127] = 0;
    count++;
    if (strncmp(module_name, "rtw88_", 6) == 0 && count < max_names) { /* :553-559 */
        snprintf(names[count], 128, "rtw_%s", module_name + 6);
        count++;
    } else if (strncmp(module_name, "rtl", 3) == 0 && count < max_names) { /* :560-565 */
        snprintf(names[count], 128, "rtw_%s", module_name + 3);
        count++;
    }
    return count;
}

/* rec: all the hosts' device records (16 bytes each, layout of mcx_dev_rec_t: subsystem in
 * byte 0, driver number in bytes 12..15), host h owns rec_off[h] .. rec_off[h+1]; driver and
 * module names packed with offsets; in_use[k] for k < root_off[H]. */
void orc_fleet_in_use(const uint32_t *root_off, const uint32_t *root_ids, uint32_t H, const uint8_t *rec,
                      const uint32_t *rec_off, const char *drv_arena, const uint32_t *drv_off, uint32_t D,
                      const char *mod_names, const uint32_t *mod_off, uint32_t V, uint8_t *in_use)
{
    static const uint8_t buses[3] = {1 /* pci, :590-624 */, 0 /* usb, :626-660 */, 4 /* pcmcia, :662-696 */};
    for (uint32_t h = 0; h < H; ++h)
        for (uint32_t k = root_off[h]; k < root_off[h + 1]; ++k) {
            const uint32_t m = root_ids[k];
            in_use[k] = 0;
            if (m >= V)
                continue;
            char mod[256];
            size_t ml = mod_off[m + 1] - mod_off[m];
            if (ml > 255)
                ml = 255;
            memcpy(mod, mod_names + mod_off[m], ml);
            mod[ml] = 0;
            if (mod[0] == 0) /* :575-578 */
                continue;
            char names[4][128];
            const int n_names = orc_names_of_module(mod, names, 4);
            for (int n = 0; n < n_names && !in_use[k]; ++n)
                for (int b = 0; b < 3 && !in_use[k]; ++b)
                    for (uint32_t i = rec_off[h]; i < rec_off[h + 1]; ++i) {
                        const uint8_t *r = rec + 16 * (size_t)i;
                        uint32_t d;
                        memcpy(&d, r + 12, 4);
                        if (r[0] != buses[b] || d >= D)
                            continue;
                        const size_t dl = drv_off[d + 1] - drv_off[d];
                        if (dl == strlen(names[n]) && memcmp(drv_arena + drv_off[d], names[n], dl) == 0) {
                            in_use[k] = 1;
                            break;
                        }
                    }
        }
}

/*
 * All matching table positions per string, in table order (what kmod's
 * `modprobe -R` prints module by module), the first K of them when K > 0.
 * list_idx may be NULL to count only.  Returns list_off[N].
 */
uint64_t orc_alias_match_list(const char *arena, const uint32_t *a_off, uint32_t P, const char *S,
                              const uint32_t *s_off, uint32_t N, int normalize_dash, uint32_t K,
                              uint32_t *list_off, uint32_t *list_idx)
{
    size_t abytes = a_off[P] + (size_t)P + 1, maxlen = 0;
    char *pat = (char *)malloc(abytes);
    size_t *pstart = (size_t *)malloc(sizeof(size_t) * ((size_t)P + 1));
    size_t w = 0;
    for (uint32_t j = 0; j < P; ++j) {
        size_t n = a_off[j + 1] - a_off[j];
        pstart[j] = w;
        memcpy(pat + w, arena + a_off[j], n);
        pat[w + n] = 0;
        if (normalize_dash && !orc_normalize(pat + w))
            pstart[j] = (size_t)-1;
        w += n + 1;
    }
    for (uint32_t i = 0; i < N; ++i)
        if (s_off[i + 1] - s_off[i] > maxlen)
            maxlen = s_off[i + 1] - s_off[i];
    char *str = (char *)malloc(maxlen + 1);
    orc_locale_t l;
    orc_enter_c_locale(&l);
    uint64_t total = 0;
    list_off[0] = 0;
    for (uint32_t i = 0; i < N; ++i) {
        size_t n = s_off[i + 1] - s_off[i];
        memcpy(str, S + s_off[i], n);
        str[n] = 0;
        int ok = normalize_dash ? orc_normalize(str) : 1;
        uint32_t kept = 0;
        for (uint32_t j = 0; ok && j < P && (K == 0 || kept < K); ++j) {
            if (pstart[j] == (size_t)-1)
                continue;
            if (fnmatch(pat + pstart[j], str, 0) == 0) {
                if (list_idx)
                    list_idx[total + kept] = j;
                ++kept;
            }
        }
        total += kept;
        list_off[i + 1] = (uint32_t)total;
    }
    orc_leave_c_locale(&l);
    free(str);
    free(pstart);
    free(pat);
    return total;
}

/* ---- fleet inventory: the device list of each host ------------------------ */

/*
 * A packed device record (the layout of include/mcx.h, restated here so that
 * the oracle depends on nothing of the product).  The oracle turns a record
 * back into the strings a sysfs node would hold and then runs the reference's
 * tests on those strings the way the reference does: sscanf("0x%x") and the
 * class masks of mc_is_infrastructure_device (reference
 * montecarlo/libmontecarlo.c:748-762), strcmp against the 8 driver names
 * (778-785), the SCSI devtype tests (790-818), and in mc_list_all_devices the
 * usb_interface / hub / parent tests (291-321), strstr "hcd" / strcmp "hub"
 * (412-413) and the `count >= max` stop (259-260); mc_is_excluded_device's
 * three "08" tests (841-869).
 */
typedef struct {
    uint8_t subsystem, devtype, attrs, iface_class;
    uint32_t pci_class;
    uint8_t parent_class, own_class;
    uint16_t reserved;
    uint32_t driver;
} orc_dev_rec;

static const char *orc_subsystem_name(uint8_t s)
{
    static const char *names[] = {"usb", "pci", "hid", "scsi", "pcmcia"};
    return s < 5 ? names[s] : "other";
}

static const char *orc_devtype_name(uint8_t t)
{
    static const char *names[] = {NULL, "usb_device", "usb_interface", "scsi_host", "scsi_target", "scsi_generic",
                                  "scsi_device"};
    return t < 7 ? names[t] : "other";
}

/* mc_is_infrastructure_device over the strings of one node; driver_target is
 * the link target of <syspath>/driver or NULL */
static int orc_is_infrastructure(const char *subsystem, const char *class_str, const char *driver_target,
                                 const char *devtype, const char *model, const char *vendor)
{
    if (strcmp(subsystem, "pci") == 0) {
        if (class_str) {
            unsigned int class_code = 0;
            sscanf(class_str, "0x%x", &class_code);
            unsigned int base_class = (class_code >> 16) & 0xFF, sub_class = (class_code >> 8) & 0xFF;
            if (base_class == 0x06 || (base_class == 0x0c && sub_class == 0x05) || base_class == 0x08)
                return 1;
        }
        if (driver_target) {
            const char *name = strrchr(driver_target, '/');
            if (!name)
                return 1;
            ++name;
            static const char *infra[] = {"pcieport", "pci_bridge", "pciehp",  "pcie_aspm", "pcie_pme",
                                          "pcie_edr", "shpchp",     "piix4_smbus", NULL};
            for (int i = 0; infra[i]; ++i)
                if (strcmp(name, infra[i]) == 0)
                    return 1;
        }
    }
    if (strcmp(subsystem, "scsi") == 0) {
        if (!devtype)
            return 1;
        if (strcmp(devtype, "scsi_host") == 0 || strcmp(devtype, "scsi_target") == 0 ||
            strcmp(devtype, "scsi_generic") == 0)
            return 1;
        if (!model && !vendor)
            return 1;
    }
    return 0;
}

/*
 * verdict[i]: bit 0 = listed, bit 1 = excluded (mass storage); kept_off[h+1] -
 * kept_off[h] = listed devices of host h.  max_per_host 0 = no limit.
 * Driver d of the table is the string drv_arena[drv_off[d] .. drv_off[d+1]);
 * 0xFFFFFFFF = no link, 0xFFFFFFFE = a link target without '/'.
 */
void orc_device_filter(const orc_dev_rec *rec, uint32_t N, const char *drv_arena, const uint32_t *drv_off,
                       uint32_t D, const uint32_t *dev_off, uint32_t H, uint32_t max_per_host, uint8_t *verdict,
                       uint32_t *kept_off)
{
    (void)N;
    kept_off[0] = 0;
    for (uint32_t h = 0; h < H; ++h) {
        uint32_t count = 0;
        for (uint32_t i = dev_off[h]; i < dev_off[h + 1]; ++i) {
            const orc_dev_rec *r = &rec[i];
            char class_buf[16], ic[4], pc[4], oc[4], target[160];
            const char *subsystem = orc_subsystem_name(r->subsystem);
            const char *devtype = orc_devtype_name(r->devtype);
            const char *class_str = NULL, *iface = NULL, *pclass = NULL, *oclass = NULL, *driver_target = NULL;
            if (r->attrs & 0x01) {
                snprintf(class_buf, sizeof class_buf, "0x%06x", r->pci_class & 0xFFFFFFu);
                class_str = class_buf;
            }
            if (r->attrs & 0x02) {
                snprintf(ic, sizeof ic, "%02x", r->iface_class);
                iface = ic;
            }
            const int has_parent = (r->attrs & 0x04) != 0;
            if (has_parent && (r->attrs & 0x08)) {
                snprintf(pc, sizeof pc, "%02x", r->parent_class);
                pclass = pc;
            }
            if (r->attrs & 0x10) {
                snprintf(oc, sizeof oc, "%02x", r->own_class);
                oclass = oc;
            }
            const char *model = (r->attrs & 0x20) ? "model" : NULL, *vendor = (r->attrs & 0x40) ? "vendor" : NULL;
            if (r->driver == 0xFFFFFFFEu) {
                driver_target = "dangling";
            } else if (r->driver != 0xFFFFFFFFu && r->driver < D) {
                uint32_t n = drv_off[r->driver + 1] - drv_off[r->driver];
                if (n > 100)
                    n = 100;
                snprintf(target, sizeof target, "../../../bus/%s/drivers/%.*s", subsystem, (int)n,
                         drv_arena + drv_off[r->driver]);
                driver_target = target;
            }

            uint8_t v = 0;
            /* mc_is_excluded_device */
            if ((oclass && strcmp(oclass, "08") == 0) || (iface && strcmp(iface, "08") == 0) ||
                (pclass && strcmp(pclass, "08") == 0))
                v |= 2;

            /* mc_list_all_devices */
            int listed = 1;
            if (count >= max_per_host && max_per_host)
                listed = 0;
            else if (orc_is_infrastructure(subsystem, class_str, driver_target, devtype, model, vendor))
                listed = 0;
            else if (strcmp(subsystem, "usb") == 0) {
                if (!devtype || strcmp(devtype, "usb_interface") != 0)
                    listed = 0;
                else if (iface && strcmp(iface, "09") == 0)
                    listed = 0;
                else if (!has_parent)
                    listed = 0;
                else if (pclass && strcmp(pclass, "09") == 0)
                    listed = 0;
            } else if (strcmp(subsystem, "pci") != 0 && strcmp(subsystem, "hid") != 0 &&
                       strcmp(subsystem, "scsi") != 0 && strcmp(subsystem, "pcmcia") != 0)
                listed = 0;
            if (listed && driver_target) {
                const char *dname = strrchr(driver_target, '/');
                const char *final_driver = dname ? dname + 1 : "unknown";
                if (strcmp(subsystem, "usb") == 0 &&
                    (strstr(final_driver, "hcd") != NULL || strcmp(final_driver, "hub") == 0))
                    listed = 0;
            }
            if (listed) {
                v |= 1;
                ++count;
            }
            verdict[i] = v;
        }
        kept_off[h + 1] = kept_off[h] + count;
    }
}
