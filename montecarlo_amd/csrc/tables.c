/*
 * tables.c - the name tables of the library as data.
 *
 * The reference keeps them as literals spread over its sources: the 113 core
 * module names and 56 prefixes of is_safe_module (desktop/ui.py:1402-1470,
 * 1486-1514), the 6 bus tags of a hardware alias (desktop/ui.py:1540), the 8
 * PCI port / hot-plug driver names (montecarlo/libmontecarlo.c:778-780) and
 * the rtw88 / rtl driver-name variants (montecarlo/libmontecarlo.c:552-565).
 * Here they are one table set with compiled-in defaults (the reference's
 * values, verbatim) that a text file can replace at run time:
 *
 *     format 1
 *     [core]              one name per line
 *     [prefix]
 *     [hw_alias_prefix]
 *     [infra_driver]
 *     [name_variant]      "<prefix> <replacement>": the first rule whose prefix
 *                         starts a module name adds <replacement><rest> as a
 *                         second driver name
 *
 * '#' starts a comment.  The active set is process-wide; replace it before
 * other threads use the library.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tables.h"

/* sorted (strcmp order) */
static const char *const CORE_DEFAULT[] = {
    "8021q", "acpi_cpufreq", "acpi_pad", "acpi_thermal", "bonding", "bridge", "btrfs", "cifs", "cpuid",
    "dm_crypt", "dm_mirror", "dm_mod", "dm_snapshot", "dmi_notifier", "dmi_sysfs", "efi_pstore", "efivarfs",
    "efivars", "evdev", "exportfs", "ext2", "ext3", "ext4", "fat", "fuse", "fuseblk", "hid", "hid_generic",
    "hidp", "input_leds", "iso9660", "jbd2", "jfs", "joydev", "kvm", "kvm_amd", "kvm_intel", "led_class",
    "linear", "llc", "lockd", "loop", "lp", "mbcache", "md_mod", "minix", "mousedev", "msdos", "msr",
    "multipath", "nbd", "nfs", "nfsd", "ntfs", "ntfs3", "overlayfs", "parport", "parport_pc", "ppdev",
    "pstore", "pstore_blk", "pstore_ram", "raid0", "raid1", "raid10", "raid456", "raid6_pq", "reiserfs",
    "smb", "smbfs", "snd", "snd_hda_codec", "snd_hda_codec_generic", "snd_hda_core", "snd_hda_intel",
    "snd_hwdep", "snd_pcm", "snd_rawmidi", "snd_seq", "snd_seq_device", "snd_seq_midi",
    "snd_seq_midi_event", "snd_timer", "soundcore", "squashfs", "stp", "tap", "tun", "udf", "uhid",
    "usbhid", "usbkbd", "usbmouse", "vboxdrv", "vboxnetadp", "vboxnetflt", "vboxpci", "veth", "vfat",
    "vhost", "vhost_net", "vhost_vsock", "virtio", "virtio_balloon", "virtio_blk", "virtio_net",
    "virtio_pci", "vmw_balloon", "vmw_vmci", "vmw_vsock_vmci_transport", "xfs", "zram", "zsmalloc",
};

static const char *const PREFIX_DEFAULT[] = {
    "xt_", "nf_", "nft_", "ip_", "ip6_", "ipt_", "ip6t_", "nfnetlink", "netfilter", "conntrack",
    "crypto_", "sha", "aes", "ghash", "crc32", "md5", "des", "ecb", "cbc", "gcm", "ccm", "ctr",
    "k10temp", "coretemp", "ssse3", "aesni", "cpu_", "cpufreq", "intel_", "amd_", "x86_pkg_temp",
    "acpi", "battery", "ac", "button", "fan", "thermal", "i2c_", "spi_", "smbus", "pcieport",
    "pci_bridge", "shpchp", "scsi_mod", "sd_mod", "sr_mod", "sg", "st", "nvme_core", "videodev",
    "videobuf", "v4l2_common", "snd_", "hid_", "vbox", "vmw_",
};

static const char *const HW_PREFIX_DEFAULT[] = {"pci:", "usb:", "platform:", "hid:", "serio:", "of:"};

static const char *const INFRA_DEFAULT[] = {"pcieport", "pci_bridge", "pciehp",  "pcie_aspm",
                                            "pcie_pme", "pcie_edr",   "shpchp", "piix4_smbus"};

/* pairs: prefix, replacement */
static const char *const VARIANT_DEFAULT[] = {"rtw88_", "rtw_", "rtl", "rtw_"};

#define COUNT(a) (uint32_t)(sizeof(a) / sizeof((a)[0]))

static const mcx_tables_view DEFAULTS = {
    MCX_TABLES_FORMAT,
    {CORE_DEFAULT, PREFIX_DEFAULT, HW_PREFIX_DEFAULT, INFRA_DEFAULT, VARIANT_DEFAULT},
    {COUNT(CORE_DEFAULT), COUNT(PREFIX_DEFAULT), COUNT(HW_PREFIX_DEFAULT), COUNT(INFRA_DEFAULT),
     COUNT(VARIANT_DEFAULT)},
};

static const char *const SECTION_NAMES[MCX_TS_COUNT] = {"core", "prefix", "hw_alias_prefix", "infra_driver",
                                                        "name_variant"};

/* a loaded set owns one block of text and the pointer arrays into it */
typedef struct {
    mcx_tables_view view;
    char *text;
    const char **items[MCX_TS_COUNT];
} loaded_tables;

static loaded_tables *g_loaded; /* NULL: the defaults are active */
static uint64_t g_generation = 1;

const mcx_tables_view *mcx_tables_active(void)
{
    return g_loaded ? &g_loaded->view : &DEFAULTS;
}

uint64_t mcx_tables_generation(void)
{
    return g_generation;
}

static void free_loaded(loaded_tables *t)
{
    if (!t)
        return;
    for (int s = 0; s < MCX_TS_COUNT; ++s)
        free((void *)t->items[s]);
    free(t->text);
    free(t);
}

void mcx_tables_reset(void)
{
    loaded_tables *old = g_loaded;
    g_loaded = NULL;
    ++g_generation;
    free_loaded(old);
}

static int cmp_names(const void *a, const void *b)
{
    return strcmp(*(const char *const *)a, *(const char *const *)b);
}

int mcx_tables_load_text(const char *text, size_t len)
{
    if (!text)
        return MCX_E_BAD_ARG;
    loaded_tables *t = calloc(1, sizeof *t);
    if (!t)
        return MCX_E_NOMEM;
    t->text = malloc(len + 1);
    if (!t->text) {
        free(t);
        return MCX_E_NOMEM;
    }
    memcpy(t->text, text, len);
    t->text[len] = '\0';
    uint32_t cap[MCX_TS_COUNT] = {0};
    int section = -1, format = 0, rc = MCX_OK;
    char *save = NULL;
    for (char *line = strtok_r(t->text, "\n", &save); line && rc == MCX_OK; line = strtok_r(NULL, "\n", &save)) {
        char *hash = strchr(line, '#');
        if (hash)
            *hash = '\0';
        char *w1 = line + strspn(line, " \t\r");
        size_t n1 = strcspn(w1, " \t\r");
        if (n1 == 0)
            continue;
        char *w2 = w1 + n1 + strspn(w1 + n1, " \t\r");
        size_t n2 = strcspn(w2, " \t\r");
        w1[n1] = '\0';
        w2[n2] = '\0';
        if (w1[0] == '[') {
            section = -1;
            for (int s = 0; s < MCX_TS_COUNT; ++s)
                if (strlen(SECTION_NAMES[s]) + 2 == n1 && w1[n1 - 1] == ']' &&
                    strncmp(w1 + 1, SECTION_NAMES[s], n1 - 2) == 0)
                    section = s;
            if (section < 0)
                rc = MCX_E_BAD_ARG; /* unknown section */
            continue;
        }
        if (section < 0) {
            if (strcmp(w1, "format") == 0 && atoi(w2) == MCX_TABLES_FORMAT)
                format = MCX_TABLES_FORMAT;
            else
                rc = MCX_E_BAD_ARG;
            continue;
        }
        const int words = section == MCX_TS_VARIANT ? 2 : 1;
        if ((words == 2) != (n2 != 0)) {
            rc = MCX_E_BAD_ARG; /* a rule needs two words, a name is one */
            break;
        }
        uint32_t *n = &t->view.count[section];
        if (*n + 2 > cap[section]) {
            cap[section] = cap[section] ? 2 * cap[section] : 64;
            const char **grown = realloc((void *)t->items[section], cap[section] * sizeof(char *));
            if (!grown) {
                rc = MCX_E_NOMEM;
                break;
            }
            t->items[section] = grown;
        }
        t->items[section][(*n)++] = w1;
        if (words == 2)
            t->items[section][(*n)++] = w2;
    }
    if (rc == MCX_OK && format != MCX_TABLES_FORMAT)
        rc = MCX_E_BAD_ARG;
    if (rc != MCX_OK) {
        free_loaded(t);
        return rc;
    }
    if (t->view.count[MCX_TS_CORE])
        qsort((void *)t->items[MCX_TS_CORE], t->view.count[MCX_TS_CORE], sizeof(char *), cmp_names);
    t->view.format = format;
    for (int s = 0; s < MCX_TS_COUNT; ++s)
        t->view.item[s] = t->items[s];
    loaded_tables *old = g_loaded;
    g_loaded = t;
    ++g_generation;
    free_loaded(old);
    return MCX_OK;
}

int mcx_tables_load_file(const char *path)
{
    if (!path)
        return MCX_E_BAD_ARG;
    FILE *f = fopen(path, "rb");
    if (!f)
        return MCX_E_BAD_ARG;
    size_t cap = 1 << 16, n = 0;
    char *buf = malloc(cap);
    while (buf) {
        n += fread(buf + n, 1, cap - n, f);
        if (n < cap)
            break;
        cap *= 2;
        char *grown = realloc(buf, cap);
        if (!grown)
            free(buf);
        buf = grown;
    }
    fclose(f);
    if (!buf)
        return MCX_E_NOMEM;
    int rc = mcx_tables_load_text(buf, n);
    free(buf);
    return rc;
}

int mcx_tables_dump(char *buf, size_t cap, size_t *need)
{
    const mcx_tables_view *v = mcx_tables_active();
    size_t at = 0;
#define EMIT(...)                                                                  \
    do {                                                                           \
        int w_ = snprintf(buf && at < cap ? buf + at : NULL, buf && at < cap ? cap - at : 0, __VA_ARGS__); \
        if (w_ > 0)                                                                \
            at += (size_t)w_;                                                      \
    } while (0)
    EMIT("# libmontecarlo name tables\nformat %d\n", v->format);
    for (int s = 0; s < MCX_TS_COUNT; ++s) {
        EMIT("[%s]\n", SECTION_NAMES[s]);
        if (s == MCX_TS_VARIANT)
            for (uint32_t k = 0; k + 1 < v->count[s]; k += 2)
                EMIT("%s %s\n", v->item[s][k], v->item[s][k + 1]);
        else
            for (uint32_t k = 0; k < v->count[s]; ++k)
                EMIT("%s\n", v->item[s][k]);
    }
#undef EMIT
    if (need)
        *need = at + 1;
    return (buf && at < cap) ? MCX_OK : MCX_E_TOO_LARGE;
}

int mcx_tables_info(uint32_t *format, uint32_t counts[5], int *is_default)
{
    const mcx_tables_view *v = mcx_tables_active();
    if (format)
        *format = (uint32_t)v->format;
    if (counts)
        for (int s = 0; s < MCX_TS_COUNT; ++s)
            counts[s] = s == MCX_TS_VARIANT ? v->count[s] / 2 : v->count[s];
    if (is_default)
        *is_default = g_loaded == NULL;
    return MCX_OK;
}
