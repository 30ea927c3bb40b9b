/*
 * legacy_sysfs.c - the scalar mc_* ABI of include/libmontecarlo.h: one sysfs /
 * procfs question per call, answered on the host.  Behaviour follows reference
 * montecarlo/libmontecarlo.c function by function (line ranges at each one).
 *
 * The reference goes through libudev; its development header is not part of
 * this image, and everything libudev does for these calls is a handful of
 * sysfs reads (attribute files, the `subsystem` / `driver` links, DEVTYPE from
 * `uevent`, parents by walking up the device path), which are done here
 * directly with the rules libudev applies (sd-device: attribute values lose
 * trailing newlines, only the driver / subsystem / module links read as values,
 * directories and unreadable files read as absent, enumeration sorted by
 * syspath).
 *
 * Deliberate differences from the reference, all on the safe side:
 *   - mc_try_load_driver / mc_unload_driver hand the name to a shell in the
 *     reference (libmontecarlo.c:123,133); here a name with anything but
 *     [A-Za-z0-9_-] is refused (the rule of the reference's own privileged
 *     helper, montecarlo/helper.c:18-43, widened to names that start with a
 *     digit such as 8250_pci);
 *   - mc_list_candidate_drivers with max <= 0 writes nothing (the reference
 *     writes one entry before it looks at max);
 *   - mc_get_device_subsystem returns a per-thread buffer, not a process-wide one.
 * Commands are the reference's own ("modprobe", "dmesg | tail -n 30", found
 * through PATH by the shell as there); no environment variable names a command
 * or a path in the shipped build.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <limits.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/libmontecarlo.h"
#include "tables.h"

/* ---- helpers ---------------------------------------------------------------- */

/* The absolute /sys and /proc paths the library builds itself.  The shipped
 * library uses them as they are, like the reference; only the test build
 * (-DMCX_TEST_SEAMS, a separate file that nothing installs) looks them up below
 * $MCX_SYSFS_ROOT, so that the tests can walk a fake tree.  This library is
 * loaded by root processes: no environment variable reaches a path or a command
 * in the shipped build. */
static void sys_path(char *out, size_t n, const char *abs)
{
#ifdef MCX_TEST_SEAMS
    const char *root = getenv("MCX_SYSFS_ROOT");
    snprintf(out, n, "%s%s", root ? root : "", abs);
#else
    snprintf(out, n, "%s", abs);
#endif
}

/* what udev_device_get_sysattr_value returns: file content without trailing
 * newlines; the target's basename for the driver / subsystem / module links;
 * NULL (0) for anything else */
static int sysattr(const char *syspath, const char *name, char *val, size_t n)
{
    char p[PATH_MAX];
    struct stat st;
    snprintf(p, sizeof p, "%s/%s", syspath, name);
    if (lstat(p, &st) != 0)
        return 0;
    if (S_ISLNK(st.st_mode)) {
        if (strcmp(name, "driver") != 0 && strcmp(name, "subsystem") != 0 && strcmp(name, "module") != 0)
            return 0;
        char t[PATH_MAX];
        ssize_t l = readlink(p, t, sizeof t - 1);
        if (l < 0)
            return 0;
        t[l] = 0;
        const char *b = strrchr(t, '/');
        snprintf(val, n, "%s", b ? b + 1 : t);
        return 1;
    }
    if (S_ISDIR(st.st_mode) || !(st.st_mode & S_IRUSR))
        return 0;
    FILE *f = fopen(p, "r");
    if (!f)
        return 0;
    size_t got = fread(val, 1, n - 1, f);
    fclose(f);
    val[got] = 0;
    while (got && val[got - 1] == '\n')
        val[--got] = 0;
    return 1;
}

/* udev_device_new_from_syspath succeeds for a directory that has a uevent file */
static int is_device(const char *syspath)
{
    char p[PATH_MAX];
    struct stat st;
    snprintf(p, sizeof p, "%s/uevent", syspath);
    return stat(p, &st) == 0;
}

static int dev_subsystem(const char *syspath, char *out, size_t n)
{
    return sysattr(syspath, "subsystem", out, n);
}

/* DEVTYPE=... line of uevent */
static int dev_devtype(const char *syspath, char *out, size_t n)
{
    char p[PATH_MAX], line[512];
    snprintf(p, sizeof p, "%s/uevent", syspath);
    FILE *f = fopen(p, "r");
    if (!f)
        return 0;
    int ok = 0;
    while (fgets(line, sizeof line, f)) {
        if (strncmp(line, "DEVTYPE=", 8) == 0) {
            line[strcspn(line, "\n")] = 0;
            snprintf(out, n, "%s", line + 8);
            ok = 1;
            break;
        }
    }
    fclose(f);
    return ok;
}

/* udev_device_get_parent_with_subsystem_devtype */
static int dev_parent(const char *syspath, const char *subsystem, const char *devtype, char *out, size_t n)
{
    char p[PATH_MAX], v[256];
    snprintf(p, sizeof p, "%s", syspath);
    for (;;) {
        char *slash = strrchr(p, '/');
        if (!slash || slash == p)
            return 0;
        *slash = 0;
        if (!is_device(p))
            continue;
        if (!dev_subsystem(p, v, sizeof v) || strcmp(v, subsystem) != 0)
            continue;
        if (!dev_devtype(p, v, sizeof v) || strcmp(v, devtype) != 0)
            continue;
        snprintf(out, n, "%s", p);
        return 1;
    }
}

static int name_is_safe(const char *s, size_t maxlen)
{
    size_t n = strlen(s);
    if (n == 0 || n > maxlen)
        return 0;
    for (size_t i = 0; i < n; ++i) {
        char c = s[i];
        if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'))
            return 0;
    }
    return 1;
}

/* ---- reference libmontecarlo.c:15-32 ---------------------------------------- */
int mc_read_sysattr(const char *path, char *buf, size_t buflen)
{
    FILE *f = fopen(path, "r");
    if (!f)
        return 0;
    if (!fgets(buf, (int)buflen, f)) {
        fclose(f);
        return 0;
    }
    buf[strcspn(buf, "\n")] = 0;
    fclose(f);
    return 1;
}

/* ---- :35-51 ------------------------------------------------------------------ */
void mc_get_ids(const char *syspath, char *vendor, char *product)
{
    char pv[1024], pp[1024];
    snprintf(pv, sizeof pv, "%s/idVendor", syspath);
    snprintf(pp, sizeof pp, "%s/idProduct", syspath);
    if (!mc_read_sysattr(pv, vendor, 32))
        strcpy(vendor, "0000");
    if (!mc_read_sysattr(pp, product, 32))
        strcpy(product, "0000");
}

/* ---- :63-112 ----------------------------------------------------------------- */
int mc_list_candidate_drivers(char out[][128], int max)
{
    static const char *const buses[] = {"usb", "usb-serial", "hid", "pci", "i2c", "sdio", "scsi", "pcmcia", NULL};
    int count = 0;
    if (max <= 0)
        return 0;
    for (int b = 0; buses[b]; ++b) {
        char abs[128], dirp[PATH_MAX];
        snprintf(abs, sizeof abs, "/sys/bus/%s/drivers", buses[b]);
        sys_path(dirp, sizeof dirp, abs);
        DIR *dir = opendir(dirp);
        if (!dir)
            continue;
        struct dirent *ent;
        while ((ent = readdir(dir)) != NULL) {
            if (ent->d_name[0] == '.')
                continue;
            char full[PATH_MAX];
            struct stat st;
            snprintf(full, sizeof full, "%s/%s", dirp, ent->d_name);
            if (lstat(full, &st) == 0 && S_ISDIR(st.st_mode)) {
                strncpy(out[count], ent->d_name, 127);
                out[count][127] = '\0';
                if (++count >= max) {
                    closedir(dir);
                    return count;
                }
            }
        }
        closedir(dir);
    }
    return count;
}

/* ---- :115-127 ---------------------------------------------------------------- */
int mc_try_load_driver(const char *driver)
{
    char shortname[64], cmd[512];
    if (!driver)
        return 0;
    snprintf(shortname, sizeof shortname, "%s", driver); /* the reference truncates to 63 bytes */
    if (!name_is_safe(shortname, 63))
        return 0;
    snprintf(cmd, sizeof cmd, "modprobe %s 2>/dev/null", shortname);
    return system(cmd) == 0;
}

/* ---- :130-142 ---------------------------------------------------------------- */
int mc_unload_driver(const char *driver)
{
    char cmd[512];
    int status = -1;
    if (driver && name_is_safe(driver, 200)) {
        snprintf(cmd, sizeof cmd, "modprobe -r %s 2>/dev/null", driver);
        status = system(cmd);
    }
    if (status == -1 || WEXITSTATUS(status) != 0) {
        printf("Err en la descarga de drivers.");
        return -1;
    }
    return 1;
}

/* ---- :145-166 ---------------------------------------------------------------- */
int mc_dmesg_has_activity(const char *driver)
{
    if (!driver)
        return 0;
    FILE *p = popen("dmesg | tail -n 30", "r");
    if (!p)
        return 0;
    char line[512];
    bool found = false;
    while (fgets(line, sizeof line, p)) {
        if (strstr(line, driver)) {
            found = true;
            break;
        }
    }
    pclose(p);
    return found;
}

/* ---- :169-203 ---------------------------------------------------------------- */
const char *mc_get_device_subsystem(const char *syspath)
{
    static _Thread_local char subsystem[16];
    char link_path[1024], target[1024];
    snprintf(link_path, sizeof link_path, "%s/subsystem", syspath);
    ssize_t len = readlink(link_path, target, sizeof target - 1);
    if (len == -1) {
        strcpy(subsystem, "unknown");
        return subsystem;
    }
    target[len] = '\0';
    const char *bus_name = strrchr(target, '/');
    if (!bus_name) {
        strcpy(subsystem, "unknown");
        return NULL;
    }
    /* the reference's length: strlen("/pci") - strlen(syspath) - 1 wraps around
     * for every real syspath, and the function then returns NULL (:194-197) */
    size_t subsystem_len = strlen(bus_name) - strlen(syspath) - 1;
    if (subsystem_len > sizeof(subsystem) - 1)
        return NULL;
    strncpy(subsystem, bus_name + 1, subsystem_len);
    subsystem[subsystem_len] = '\0';
    return subsystem;
}

/* ---- :206-230 ---------------------------------------------------------------- */
int mc_dev_has_driver(const char *syspath)
{
    if (!syspath || !is_device(syspath))
        return 0;
    char link[1024];
    snprintf(link, sizeof link, "%s/driver", syspath);
    return access(link, F_OK) == 0;
}

/* ---- :729-823 ---------------------------------------------------------------- */
int mc_is_infrastructure_device(const char *syspath, const char *subsystem)
{
    if (!syspath || !subsystem)
        return 0;
    if (!is_device(syspath))
        return 0;
    char v[4096], w[4096];
    if (strcmp(subsystem, "pci") == 0) {
        if (sysattr(syspath, "class", v, sizeof v)) {
            unsigned int class_code = 0;
            sscanf(v, "0x%x", &class_code);
            unsigned int base_class = (class_code >> 16) & 0xFF, sub_class = (class_code >> 8) & 0xFF;
            if (base_class == 0x06)
                return 1; /* bridges */
            if (base_class == 0x0c && sub_class == 0x05)
                return 1; /* SMBus */
            if (base_class == 0x08)
                return 1; /* system peripherals */
        }
        char link[1024], target[1024];
        snprintf(link, sizeof link, "%s/driver", syspath);
        ssize_t len = readlink(link, target, sizeof target - 1);
        if (len != -1) {
            target[len] = '\0';
            const char *name = strrchr(target, '/');
            if (!name)
                return 1;
            ++name;
            /* the 8 port / hot-plug drivers unless a table file replaced them */
            const mcx_tables_view *tv = mcx_tables_active();
            for (uint32_t i = 0; i < tv->count[MCX_TS_INFRA]; ++i)
                if (strcmp(name, tv->item[MCX_TS_INFRA][i]) == 0)
                    return 1;
        }
    }
    if (strcmp(subsystem, "scsi") == 0) {
        if (!dev_devtype(syspath, v, sizeof v))
            return 1;
        if (strcmp(v, "scsi_host") == 0 || strcmp(v, "scsi_target") == 0 || strcmp(v, "scsi_generic") == 0)
            return 1;
        if (!sysattr(syspath, "model", v, sizeof v) && !sysattr(syspath, "vendor", w, sizeof w))
            return 1;
    }
    return 0;
}

/* ---- :826-874 ---------------------------------------------------------------- */
int mc_is_excluded_device(const char *syspath)
{
    if (!syspath || !is_device(syspath))
        return 0;
    char v[4096], parent[PATH_MAX];
    if (sysattr(syspath, "bDeviceClass", v, sizeof v) && strcmp(v, "08") == 0)
        return 1;
    if (sysattr(syspath, "bInterfaceClass", v, sizeof v) && strcmp(v, "08") == 0)
        return 1;
    if (dev_parent(syspath, "usb", "usb_device", parent, sizeof parent) &&
        sysattr(parent, "bDeviceClass", v, sizeof v) && strcmp(v, "08") == 0)
        return 1;
    return 0;
}

/* ---- :233-440 ---------------------------------------------------------------- */
static int cmp_str(const void *a, const void *b)
{
    return strcmp(*(char *const *)a, *(char *const *)b);
}

int mc_list_all_devices(mc_device_info_t *out, int max)
{
    static const char *const subsystems[] = {"usb", "pci", "hid", "scsi", "pcmcia", NULL};
    char **paths = NULL;
    size_t npaths = 0, cap = 0;
    int oom = 0;
    /* what udev_enumerate_scan_devices collects: every entry of
     * /sys/bus/<subsystem>/devices, by its real path, sorted */
    for (int s = 0; subsystems[s] && !oom; ++s) {
        char abs[128], dirp[PATH_MAX];
        snprintf(abs, sizeof abs, "/sys/bus/%s/devices", subsystems[s]);
        sys_path(dirp, sizeof dirp, abs);
        DIR *dir = opendir(dirp);
        if (!dir)
            continue;
        struct dirent *ent;
        while ((ent = readdir(dir)) != NULL) {
            if (ent->d_name[0] == '.')
                continue;
            char link[PATH_MAX], real[PATH_MAX];
            snprintf(link, sizeof link, "%s/%s", dirp, ent->d_name);
            if (!realpath(link, real) || !is_device(real))
                continue;
            if (npaths == cap) {
                const size_t ncap = cap ? 2 * cap : 64;
                char **np = (char **)realloc(paths, ncap * sizeof *paths);
                if (!np) {
                    oom = 1;
                    break;
                }
                paths = np;
                cap = ncap;
            }
            char *copy = strdup(real);
            if (!copy) {
                oom = 1;
                break;
            }
            paths[npaths++] = copy;
        }
        closedir(dir);
    }
    if (oom) {
        /* out of memory while collecting: an empty list, like any other failure */
        for (size_t i = 0; i < npaths; ++i)
            free(paths[i]);
        free(paths);
        return 0;
    }
    if (npaths == 0) {
        free(paths);
        return 0;
    }
    qsort(paths, npaths, sizeof *paths, cmp_str);

    int count = 0;
    for (size_t i = 0; i < npaths; ++i) {
        if (count >= max)
            break;
        const char *path = paths[i];
        if (i > 0 && strcmp(path, paths[i - 1]) == 0)
            continue;
        char subsystem[256];
        if (!dev_subsystem(path, subsystem, sizeof subsystem))
            continue;
        if (mc_is_infrastructure_device(path, subsystem))
            continue;

        char a[4096], b[4096], c[4096], d[4096], e[4096];
        char combined_name[128];
        char vidpid[32] = "????:????";

        if (strcmp(subsystem, "usb") == 0) {
            if (!dev_devtype(path, a, sizeof a) || strcmp(a, "usb_interface") != 0)
                continue;
            if (sysattr(path, "bInterfaceClass", a, sizeof a) && strcmp(a, "09") == 0)
                continue; /* hubs */
            char parent[PATH_MAX];
            if (!dev_parent(path, "usb", "usb_device", parent, sizeof parent))
                continue;
            if (sysattr(parent, "bDeviceClass", a, sizeof a) && strcmp(a, "09") == 0)
                continue;
            int hv = sysattr(parent, "idVendor", a, sizeof a);
            int hp = sysattr(parent, "idProduct", b, sizeof b);
            int hn = sysattr(parent, "product", c, sizeof c);
            int hm = sysattr(parent, "manufacturer", d, sizeof d);
            int hi = sysattr(path, "bInterfaceNumber", e, sizeof e);
            snprintf(vidpid, sizeof vidpid, "%s:%s", hv ? a : "????", hp ? b : "????");
            const char *prod_name = hn ? c : "Unknown Device";
            const char *man_name = hm ? d : "";
            if (hi)
                snprintf(combined_name, sizeof combined_name, "%s %s (If: %s)", man_name, prod_name, e);
            else
                snprintf(combined_name, sizeof combined_name, "%s %s", man_name, prod_name);
            strncpy(out[count].product, combined_name, 127);
        } else if (strcmp(subsystem, "pci") == 0) {
            int hv = sysattr(path, "vendor", a, sizeof a);
            int hp = sysattr(path, "device", b, sizeof b);
            int hl = sysattr(path, "label", c, sizeof c);
            const char *sysname = strrchr(path, '/');
            sysname = sysname ? sysname + 1 : "Unknown";
            if (hv && hp)
                snprintf(vidpid, sizeof vidpid, "%s:%s", a, b);
            if (hl)
                strncpy(out[count].product, c, 127);
            else
                snprintf(out[count].product, 127, "PCI Device %s", sysname);
        } else if (strcmp(subsystem, "hid") == 0) {
            strncpy(vidpid, "HID", sizeof vidpid);
            int hn = sysattr(path, "name", a, sizeof a);
            snprintf(out[count].product, 127, "HID: %s", hn ? a : "HID Device");
        } else if (strcmp(subsystem, "scsi") == 0) {
            strncpy(vidpid, "SCSI", sizeof vidpid);
            int hm = sysattr(path, "model", a, sizeof a);
            int hv = sysattr(path, "vendor", b, sizeof b);
            if (!hv || !hm)
                strncpy(out[count].product, "SCSI Device", 127);
            else
                snprintf(out[count].product, 127, "%s %s", b, a);
        } else if (strcmp(subsystem, "pcmcia") == 0) {
            int hp = sysattr(path, "prod_id", a, sizeof a);
            int hm = sysattr(path, "manf_id", b, sizeof b);
            const char *sysname = strrchr(path, '/');
            sysname = sysname ? sysname + 1 : "Unknown";
            strncpy(vidpid, "PCMCIA", sizeof vidpid);
            if (!hp)
                snprintf(out[count].product, 127, "PCMCIA Device %s", sysname);
            else if (!hm)
                snprintf(out[count].product, 127, "PCMCIA: %s", a);
            else
                snprintf(out[count].product, 127, "PCMCIA: %s %s", b, a);
        } else {
            continue;
        }

        char link[PATH_MAX], target[1024];
        snprintf(link, sizeof link, "%s/driver", path);
        ssize_t len = readlink(link, target, sizeof target - 1);
        if (len != -1) {
            target[len] = '\0';
            const char *dname = strrchr(target, '/');
            const char *final_driver = dname ? dname + 1 : "unknown";
            if (strcmp(subsystem, "usb") == 0 &&
                (strstr(final_driver, "hcd") != NULL || strcmp(final_driver, "hub") == 0))
                continue; /* host controllers */
            strncpy(out[count].driver, final_driver, 63);
        } else {
            strncpy(out[count].driver, "None", 63);
        }
        strncpy(out[count].syspath, path, 255);
        strncpy(out[count].vidpid, vidpid, 31);
        strncpy(out[count].subsystem, subsystem, 15);
        out[count].subsystem[15] = '\0';
        ++count;
    }
    for (size_t i = 0; i < npaths; ++i)
        free(paths[i]);
    free(paths);
    return count;
}

/* ---- :446-473 ---------------------------------------------------------------- */
static int dir_has_entry(const char *path)
{
    DIR *dir = opendir(path);
    if (!dir)
        return 0;
    struct dirent *ent;
    int has = 0;
    while ((ent = readdir(dir)) != NULL) {
        if (ent->d_name[0] == '.')
            continue;
        has = 1;
        break;
    }
    closedir(dir);
    return has;
}

int mc_module_has_holders(const char *module)
{
    char abs[256], path[PATH_MAX];
    snprintf(abs, sizeof abs, "/sys/module/%s/holders", module);
    sys_path(path, sizeof path, abs);
    return dir_has_entry(path);
}

/* ---- :476-495 ---------------------------------------------------------------- */
int mc_get_module_refcount(const char *module)
{
    char abs[256], path[PATH_MAX];
    snprintf(abs, sizeof abs, "/sys/module/%s/refcnt", module);
    sys_path(path, sizeof path, abs);
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    int ref = 0;
    if (fscanf(f, "%d", &ref) != 1)
        ref = -1;
    fclose(f);
    return ref;
}

/* ---- :499-532 ---------------------------------------------------------------- */
int mc_list_loaded_modules(char *out_buf, int max_size)
{
    char path[PATH_MAX];
    sys_path(path, sizeof path, "/proc/modules");
    FILE *f = fopen(path, "r");
    if (!f)
        return 0;
    int count = 0, written = 0;
    char line[256];
    while (fgets(line, sizeof line, f)) {
        char name[64];
        if (sscanf(line, "%63s", name) == 1) {
            int len = (int)strlen(name);
            if (written + len + 1 < max_size) {
                strcpy(out_buf + written, name);
                written += len + 1;
                count++;
            } else {
                break;
            }
        }
    }
    fclose(f);
    return count;
}

/* ---- :538-566 ---------------------------------------------------------------- */
static void driver_names(const char *module_name, char names[][128], int *count, int max_names)
{
    *count = 0;
    if (!module_name || *count >= max_names)
        return;
    strncpy(names[*count], module_name, 127);
    names[*count][127] = '\0';
    (*count)++;
    /* the first rule whose prefix starts the name gives a second name; the
     * default rules: rtw88_8821cu -> rtw_8821cu, else rtlX -> rtw_X */
    const mcx_tables_view *tv = mcx_tables_active();
    for (uint32_t k = 0; k + 1 < tv->count[MCX_TS_VARIANT] && *count < max_names; k += 2) {
        const char *prefix = tv->item[MCX_TS_VARIANT][k];
        const size_t pl = strlen(prefix);
        if (strncmp(module_name, prefix, pl) == 0) {
            snprintf(names[*count], 128, "%s%s", tv->item[MCX_TS_VARIANT][k + 1], module_name + pl);
            (*count)++;
            break;
        }
    }
}

/* a device binding = a symlink entry that is not one of the driver's own files */
static int driver_dir_has_binding(const char *dirp)
{
    DIR *dir = opendir(dirp);
    if (!dir)
        return 0;
    struct dirent *entry;
    int bound = 0;
    while ((entry = readdir(dir)) != NULL) {
        const char *n = entry->d_name;
        if (n[0] == '.')
            continue;
        if (!strcmp(n, "bind") || !strcmp(n, "unbind") || !strcmp(n, "uevent") || !strcmp(n, "module") ||
            !strcmp(n, "new_id") || !strcmp(n, "remove_id"))
            continue;
        char full[PATH_MAX];
        struct stat sb;
        snprintf(full, sizeof full, "%s/%s", dirp, n);
        if (lstat(full, &sb) == 0 && S_ISLNK(sb.st_mode)) {
            bound = 1;
            break;
        }
    }
    closedir(dir);
    return bound;
}

/* ---- :573-725 ---------------------------------------------------------------- */
int mc_driver_is_in_use(const char *driver_name)
{
    static const char *const buses[] = {"pci", "usb", "pcmcia", NULL};
    if (!driver_name || driver_name[0] == '\0')
        return 0;
    char names[4][128];
    int name_count = 0;
    driver_names(driver_name, names, &name_count, 4);
    for (int n = 0; n < name_count; ++n) {
        for (int b = 0; buses[b]; ++b) {
            char abs[512], path[PATH_MAX];
            snprintf(abs, sizeof abs, "/sys/bus/%s/drivers/%s", buses[b], names[n]);
            sys_path(path, sizeof path, abs);
            if (driver_dir_has_binding(path))
                return 1;
        }
    }
    /* holders, under the module's own name */
    char abs[512], path[PATH_MAX];
    snprintf(abs, sizeof abs, "/sys/module/%s/holders", driver_name);
    sys_path(path, sizeof path, abs);
    return dir_has_entry(path);
}
