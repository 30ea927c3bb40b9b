/*
 * libmontecarlo.h - the scalar C ABI of the reference library, kept so that its
 * callers (the GTK UI through ctypes, the daemon and the CLI through -lmontecarlo)
 * can load this build unchanged.  Prototypes, the device record and the return
 * conventions are those of reference heads/libmontecarlo.h:10-36; each entry
 * names the reference function it replaces.  These calls walk sysfs / procfs on
 * the host (one object per call; nothing here runs on the GPU).  The batch
 * counterparts are in mcx.h.
 *
 * Conventions inherited from the reference: predicates return 1 / 0 with 0 also
 * meaning "could not tell"; list functions return a count and stop silently when
 * the caller's buffer is full; all buffers are caller allocated.
 *
 * The shipped library reads no environment variable (it does not import getenv; what
 * can be tuned is an argument of mcx_ctx_tune, mcx.h).  (A separate test build,
 * compiled with -DMCX_TEST_SEAMS and never installed, looks the absolute paths
 * the library builds itself - /sys/bus/..., /sys/module/..., /proc/modules - up
 * below $MCX_SYSFS_ROOT so that tests can walk a fake tree.)
 */
#ifndef LIBMONTECARLO_H
#define LIBMONTECARLO_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference heads/libmontecarlo.h:14-20; 496 bytes, no padding */
typedef struct {
    char syspath[256];
    char vidpid[32];
    char product[128];
    char driver[64];
    char subsystem[16];
} mc_device_info_t;

/* montecarlo/libmontecarlo.c:15-32: first line of a file, newline removed; 1 / 0 */
int mc_read_sysattr(const char *path, char *buf, size_t buflen);
/* :35-51: <syspath>/idVendor and idProduct, "0000" when unreadable (32-byte buffers) */
void mc_get_ids(const char *syspath, char *vendor, char *product);
/* :63-112: sub-directories of the 8 /sys/bus/<bus>/drivers directories, 128-byte slots */
int mc_list_candidate_drivers(char out[][128], int max);
/* :233-440: usb interfaces, pci, hid, scsi and pcmcia devices that are not
 * infrastructure, with their bound driver ("None" when unbound), in syspath order */
int mc_list_all_devices(mc_device_info_t *out, int max);
/* :169-203: kept with its arithmetic, which yields NULL for practically every input */
const char *mc_get_device_subsystem(const char *syspath);
/* :115-127: runs `modprobe <driver>`; 1 on exit status 0, else 0 */
int mc_try_load_driver(const char *driver);
/* :130-142: runs `modprobe -r <driver>`; 1 on success, -1 on failure */
int mc_unload_driver(const char *driver);
/* :145-166: does one of the last 30 dmesg lines contain the string */
int mc_dmesg_has_activity(const char *driver);
/* :446-473: /sys/module/<module>/holders has an entry */
int mc_module_has_holders(const char *module);
/* :476-495: integer in /sys/module/<module>/refcnt, -1 when absent */
int mc_get_module_refcount(const char *module);
/* :499-532: first token of every /proc/modules line, NUL separated; returns the count */
int mc_list_loaded_modules(char *out_buf, int max_size);
/* :573-725: a device symlink under /sys/bus/{pci,usb,pcmcia}/drivers/<name or its
 * rtw88_/rtl variant>, or holders */
int mc_driver_is_in_use(const char *driver);

/* :206-230 */
int mc_dev_has_driver(const char *syspath);
/* :826-874: USB mass storage (class 08) on the device, the interface or the parent device */
int mc_is_excluded_device(const char *syspath);
/* :729-823: PCI bridges / SMBus / system peripherals and 8 port drivers; SCSI hosts,
 * targets, generic nodes */
int mc_is_infrastructure_device(const char *syspath, const char *subsystem);

#ifdef __cplusplus
}
#endif
#endif /* LIBMONTECARLO_H */
