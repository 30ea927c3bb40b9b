/*
 * name_filter.c - the name stages of the reference's safe-module filter, as a
 * host function over a whole list of module names.
 *
 * Reference desktop/ui.py:1391-1518 (is_safe_module): a module is dropped when
 * its name is one of 113 core names (nine set displays, ui.py:1402-1470) or
 * starts with one of 56 prefixes (dangerous_patterns, ui.py:1486-1514, bare
 * str.startswith).  The third stage, "has a hardware alias", is
 * mcx_alias_hw_flags.  The tables are data of the reference and are kept
 * verbatim; the lookup is a binary search over the sorted core names and a
 * first-byte bucket over the prefixes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mcx.h"

/* sorted (strcmp order) */
static const char *const CORE_SORTED[] = {
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
#define N_CORE (sizeof CORE_SORTED / sizeof CORE_SORTED[0])

static const char *const PREFIXES[] = {
    "xt_", "nf_", "nft_", "ip_", "ip6_", "ipt_", "ip6t_", "nfnetlink", "netfilter", "conntrack",
    "crypto_", "sha", "aes", "ghash", "crc32", "md5", "des", "ecb", "cbc", "gcm", "ccm", "ctr",
    "k10temp", "coretemp", "ssse3", "aesni", "cpu_", "cpufreq", "intel_", "amd_", "x86_pkg_temp",
    "acpi", "battery", "ac", "button", "fan", "thermal", "i2c_", "spi_", "smbus", "pcieport",
    "pci_bridge", "shpchp", "scsi_mod", "sd_mod", "sr_mod", "sg", "st", "nvme_core", "videodev",
    "videobuf", "v4l2_common", "snd_", "hid_", "vbox", "vmw_",
};
#define N_PREFIX (sizeof PREFIXES / sizeof PREFIXES[0])

int mcx_safe_table_sizes(uint32_t *n_core, uint32_t *n_prefix)
{
    if (n_core)
        *n_core = (uint32_t)N_CORE;
    if (n_prefix)
        *n_prefix = (uint32_t)N_PREFIX;
    return MCX_OK;
}

static int is_core(const char *name, size_t len)
{
    size_t lo = 0, hi = N_CORE;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        const char *c = CORE_SORTED[mid];
        size_t cl = strlen(c);
        int cmp = memcmp(name, c, len < cl ? len : cl);
        if (cmp == 0)
            cmp = (len > cl) - (len < cl);
        if (cmp == 0)
            return 1;
        if (cmp < 0)
            hi = mid;
        else
            lo = mid + 1;
    }
    return 0;
}

int mcx_safe_module_names(const char *names, const uint32_t *name_off, uint32_t n, uint8_t *safe_out)
{
    if (!name_off || (n && (!safe_out || (!names && name_off[n] != 0))))
        return MCX_E_BAD_ARG;
    /* prefixes bucketed by first byte */
    uint8_t first[256];
    memset(first, 0, sizeof first);
    for (size_t k = 0; k < N_PREFIX; ++k)
        first[(unsigned char)PREFIXES[k][0]] = 1;
    for (uint32_t i = 0; i < n; ++i) {
        if (name_off[i + 1] < name_off[i])
            return MCX_E_BAD_ARG;
        const char *s = names + name_off[i];
        size_t len = name_off[i + 1] - name_off[i];
        int safe = 1;
        if (is_core(s, len)) {
            safe = 0;
        } else if (len && first[(unsigned char)s[0]]) {
            for (size_t k = 0; k < N_PREFIX && safe; ++k) {
                size_t pl = strlen(PREFIXES[k]);
                if (len >= pl && memcmp(s, PREFIXES[k], pl) == 0)
                    safe = 0;
            }
        }
        safe_out[i] = (uint8_t)safe;
    }
    return MCX_OK;
}
