/*
 * mcx_synth.c - deterministic synthetic inputs for the batch path: a
 * modules.alias-style pattern table, device modalias strings, a direct
 * module-dependency DAG with per-module state bits, and module names.
 *
 * The reference has no data sets; its real inputs are a host's
 * /lib/modules/<ver>/modules.alias (a few 10^4 lines) and /proc/modules.  The
 * shapes below follow what the kernel's file2alias writes for each bus
 * (pci:v%08Xd%08Xsv%08Xsd%08Xbc%02Xsc%02Xi%02X and so on), scaled up.
 *
 * Every item is a pure function of (seed, kind, index): a splitmix64 stream
 * keyed by those three, so any range of items can be generated on its own, in
 * any order, on any rank.  Integer arithmetic only.
 *
 * Built as its own small library (libmcxsynth.so); it is data tooling, not
 * part of the matching path.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t s;
} rng_t;

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static uint64_t rng_next(rng_t *r)
{
    r->s += 0x9E3779B97F4A7C15ULL;
    return mix64(r->s);
}

static rng_t rng_stream(uint64_t seed, uint64_t kind, uint64_t index)
{
    rng_t r;
    r.s = mix64(seed ^ mix64(kind * 0xD1B54A32D192ED03ULL ^ mix64(index + 0x632BE59BD9B4E019ULL)));
    return r;
}

static uint32_t rng_below(rng_t *r, uint32_t n)
{
    return (uint32_t)(rng_next(r) % n);
}

enum { K_MODULE = 1, K_PATTERN = 2, K_STRING = 3, K_VERTEX = 4, K_FLAGS = 5, K_SEEDSET = 6, K_NAME = 7, K_HOST = 8 };

/* ---- vendors: 512 ids, Zipf(1) popularity -------------------------------- */

#define N_VENDORS 512
static uint64_t g_zipf_cdf[N_VENDORS];
static uint32_t g_vendor_id[N_VENDORS];
static int g_vendor_init;

static void vendors_init(void)
{
    if (g_vendor_init)
        return;
    uint64_t acc = 0;
    for (uint32_t k = 0; k < N_VENDORS; ++k) {
        acc += (1ULL << 32) / (k + 1);
        g_zipf_cdf[k] = acc;
        uint32_t id = (uint32_t)(mix64(0x56454E444F52ULL + k) & 0xFFFF);
        g_vendor_id[k] = id ? id : 0x8086;
    }
    g_vendor_init = 1;
}

static uint32_t pick_vendor(rng_t *r)
{
    uint64_t x = rng_next(r) % g_zipf_cdf[N_VENDORS - 1];
    uint32_t lo = 0, hi = N_VENDORS - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (x < g_zipf_cdf[mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    return g_vendor_id[lo];
}

/* ---- pattern text with holes --------------------------------------------- */

enum { BUS_PCI, BUS_USB, BUS_HID, BUS_PLATFORM, BUS_OF, BUS_SERIO, BUS_ACPI };

static int pick_bus(rng_t *r)
{
    uint32_t x = rng_below(r, 1000);
    if (x < 450) return BUS_PCI;
    if (x < 800) return BUS_USB;
    if (x < 850) return BUS_HID;
    if (x < 910) return BUS_PLATFORM;
    if (x < 950) return BUS_OF;
    if (x < 980) return BUS_SERIO;
    return BUS_ACPI;
}

enum { H_HEX, H_NAME, H_QMARK, H_SET, H_TAIL };
typedef struct {
    uint16_t pos;   /* offset of the hole's text in the pattern */
    uint8_t plen;   /* length of that text ("*" = 1, "[0-9A-F]" = 8 ...) */
    uint8_t kind;
    uint8_t arg;    /* H_HEX: canonical width; H_SET: index into g_sets */
} hole_t;

typedef struct {
    char text[160];
    uint32_t len;
    hole_t holes[16];
    uint32_t nholes;
} pat_t;

static const struct {
    const char *text;
    const char *members;
} g_sets[] = {
    {"[0-9A-F]", "0123456789ABCDEF"}, {"[0-9]", "0123456789"}, {"[A-F]", "ABCDEF"},
    {"[0-3]", "0123"}, {"[!0]", "123456789ABCDEF"}, {"[89AB]", "89AB"},
};
#define N_SETS 6

static const char *const g_words[] = {
    "gpio", "uart", "spi", "i2c", "mmc", "eth", "wdt", "rtc", "pwm", "adc", "dma", "usbphy",
    "lcd", "codec", "clk", "reg", "pmic", "nand", "sata", "pcie", "hdmi", "cam", "touch", "leds",
    "keys", "fan", "temp", "accel", "gyro", "nfc", "gnss", "modem"};
#define N_WORDS 32
static const char *const g_vendors_of[] = {"acme", "vnd", "foo", "brcm", "ti", "nxp", "st", "qcom"};
#define N_OFV 8

static void pat_puts(pat_t *p, const char *s)
{
    size_t n = strlen(s);
    memcpy(p->text + p->len, s, n);
    p->len += (uint32_t)n;
}

static void pat_hole(pat_t *p, const char *text, int kind, int arg)
{
    hole_t *h = &p->holes[p->nholes++];
    h->pos = (uint16_t)p->len;
    h->plen = (uint8_t)strlen(text);
    h->kind = (uint8_t)kind;
    h->arg = (uint8_t)arg;
    pat_puts(p, text);
}

/* one hex field: "*" with probability wild_pm/1000, else the value; when
 * special != 0 one digit of a literal field becomes '?' (1) or a bracket (2) */
static void pat_hex(pat_t *p, rng_t *r, const char *tag, uint32_t value, int width, uint32_t wild_pm,
                    int special, uint32_t digit, uint32_t setno)
{
    pat_puts(p, tag);
    if (rng_below(r, 1000) < wild_pm) {
        pat_hole(p, "*", H_HEX, width);
        return;
    }
    char buf[16];
    snprintf(buf, sizeof buf, "%0*X", width, value);
    if (!special) {
        pat_puts(p, buf);
        return;
    }
    digit %= (uint32_t)width;
    for (int k = 0; k < width; ++k) {
        if ((uint32_t)k == digit) {
            if (special == 1)
                pat_hole(p, "?", H_QMARK, 0);
            else
                pat_hole(p, g_sets[setno % N_SETS].text, H_SET, (int)(setno % N_SETS));
        } else {
            char c[2] = {buf[k], 0};
            pat_puts(p, c);
        }
    }
}

static void module_home(uint64_t seed, uint32_t m, int *bus, uint32_t *vendor)
{
    rng_t r = rng_stream(seed, K_MODULE, m);
    *bus = pick_bus(&r);
    *vendor = pick_vendor(&r);
}

static uint32_t module_of_pattern(uint32_t j)
{
    return j / 5;
}

static void gen_pattern(uint64_t seed, uint32_t j, pat_t *p)
{
    vendors_init();
    int bus;
    uint32_t vendor;
    module_home(seed, module_of_pattern(j), &bus, &vendor);
    rng_t r = rng_stream(seed, K_PATTERN, j);
    if (rng_below(&r, 10) == 0)
        vendor = pick_vendor(&r);
    /* 1 % of patterns carry a '?', 2 % a bracket, inside the device field */
    uint32_t sp = rng_below(&r, 100);
    int special = sp == 0 ? 1 : (sp <= 2 ? 2 : 0);
    uint32_t digit = rng_below(&r, 8), setno = rng_below(&r, N_SETS);
    uint32_t device = (uint32_t)rng_next(&r) & 0xFFFF;
    p->len = 0;
    p->nholes = 0;
    switch (bus) {
    case BUS_PCI:
        pat_puts(p, "pci:");
        if (rng_below(&r, 100) < 5) {
            /* class alias: every id wild, class triple literal */
            pat_hex(p, &r, "v", 0, 8, 1000, 0, 0, 0);
            pat_hex(p, &r, "d", 0, 8, 1000, 0, 0, 0);
            pat_hex(p, &r, "sv", 0, 8, 1000, 0, 0, 0);
            pat_hex(p, &r, "sd", 0, 8, 1000, 0, 0, 0);
            pat_hex(p, &r, "bc", rng_below(&r, 0x14), 2, 0, 0, 0, 0);
            pat_hex(p, &r, "sc", rng_below(&r, 8), 2, 0, 0, 0, 0);
            pat_hex(p, &r, "i", rng_below(&r, 4), 2, 500, 0, 0, 0);
        } else {
            pat_hex(p, &r, "v", vendor, 8, 0, 0, 0, 0);
            pat_hex(p, &r, "d", device, 8, 100, special, 4 + digit % 4, setno);
            pat_hex(p, &r, "sv", pick_vendor(&r), 8, 600, 0, 0, 0);
            pat_hex(p, &r, "sd", (uint32_t)rng_next(&r) & 0xFFFF, 8, 600, 0, 0, 0);
            pat_hex(p, &r, "bc", rng_below(&r, 0x14), 2, 850, 0, 0, 0);
            pat_hex(p, &r, "sc", rng_below(&r, 8), 2, 850, 0, 0, 0);
            pat_hex(p, &r, "i", rng_below(&r, 4), 2, 850, 0, 0, 0);
        }
        pat_hole(p, "*", H_TAIL, 0);
        break;
    case BUS_USB:
        pat_puts(p, "usb:");
        if (rng_below(&r, 100) < 5) {
            /* interface-class alias */
            pat_hex(p, &r, "v", 0, 4, 1000, 0, 0, 0);
            pat_hex(p, &r, "p", 0, 4, 1000, 0, 0, 0);
            pat_hex(p, &r, "d", 0, 4, 1000, 0, 0, 0);
            pat_hex(p, &r, "dc", 0, 2, 1000, 0, 0, 0);
            pat_hex(p, &r, "dsc", 0, 2, 1000, 0, 0, 0);
            pat_hex(p, &r, "dp", 0, 2, 1000, 0, 0, 0);
            pat_hex(p, &r, "ic", rng_below(&r, 0x10), 2, 0, 0, 0, 0);
            pat_hex(p, &r, "isc", rng_below(&r, 8), 2, 0, 0, 0, 0);
            pat_hex(p, &r, "ip", rng_below(&r, 4), 2, 300, 0, 0, 0);
            pat_hex(p, &r, "in", 0, 2, 1000, 0, 0, 0);
        } else {
            pat_hex(p, &r, "v", vendor, 4, 0, 0, 0, 0);
            pat_hex(p, &r, "p", device, 4, 100, special, digit % 4, setno);
            pat_hex(p, &r, "d", (uint32_t)rng_next(&r) & 0xFFFF, 4, 800, 0, 0, 0);
            pat_hex(p, &r, "dc", rng_below(&r, 0x10), 2, 850, 0, 0, 0);
            pat_hex(p, &r, "dsc", rng_below(&r, 8), 2, 850, 0, 0, 0);
            pat_hex(p, &r, "dp", rng_below(&r, 4), 2, 850, 0, 0, 0);
            pat_hex(p, &r, "ic", rng_below(&r, 0x10), 2, 700, 0, 0, 0);
            pat_hex(p, &r, "isc", rng_below(&r, 8), 2, 700, 0, 0, 0);
            pat_hex(p, &r, "ip", rng_below(&r, 4), 2, 700, 0, 0, 0);
            pat_hex(p, &r, "in", rng_below(&r, 4), 2, 800, 0, 0, 0);
        }
        pat_hole(p, "*", H_TAIL, 0);
        break;
    case BUS_HID:
        pat_puts(p, "hid:");
        pat_hex(p, &r, "b", 3 + rng_below(&r, 3), 4, 0, 0, 0, 0);
        pat_hex(p, &r, "g", 1 + rng_below(&r, 4), 4, 600, 0, 0, 0);
        pat_hex(p, &r, "v", vendor, 8, 50, 0, 0, 0);
        pat_hex(p, &r, "p", device, 8, 100, special, 4 + digit % 4, setno);
        break;
    case BUS_PLATFORM: {
        char name[48];
        snprintf(name, sizeof name, "platform:%s_%s%u", g_vendors_of[vendor % N_OFV],
                 g_words[device % N_WORDS], (device >> 8) % 16);
        pat_puts(p, name);
        if (rng_below(&r, 10) == 0)
            pat_hole(p, "*", H_TAIL, 0);
        break;
    }
    case BUS_OF: {
        char compat[48];
        snprintf(compat, sizeof compat, "%s,%s%u", g_vendors_of[vendor % N_OFV],
                 g_words[device % N_WORDS], (device >> 8) % 64);
        pat_puts(p, "of:N");
        pat_hole(p, "*", H_NAME, 0);
        pat_puts(p, "T");
        pat_hole(p, "*", H_NAME, 0);
        pat_puts(p, "C");
        pat_puts(p, compat);
        if (rng_below(&r, 2) == 0) {
            pat_puts(p, "C");
            pat_hole(p, "*", H_NAME, 0);
        }
        break;
    }
    case BUS_SERIO:
        pat_puts(p, "serio:");
        pat_hex(p, &r, "ty", rng_below(&r, 8), 2, 300, 0, 0, 0);
        pat_hex(p, &r, "pr", rng_below(&r, 0x40), 2, 500, 0, 0, 0);
        pat_hex(p, &r, "id", rng_below(&r, 0x100), 2, 500, 0, 0, 0);
        pat_hex(p, &r, "ex", rng_below(&r, 0x100), 2, 500, 0, 0, 0);
        break;
    default: { /* BUS_ACPI: acpi*:PNP0C09:* */
        char id[24];
        snprintf(id, sizeof id, ":PNP%04X:", device & 0x0FFF);
        pat_puts(p, "acpi");
        pat_hole(p, "*", H_TAIL, 0);
        pat_puts(p, id);
        pat_hole(p, "*", H_TAIL, 0);
        break;
    }
    }
}

static uint32_t fill_hole(const hole_t *h, rng_t *r, char *out)
{
    static const char hex[] = "0123456789ABCDEF";
    uint32_t n = 0;
    switch (h->kind) {
    case H_HEX:
        for (int k = 0; k < h->arg; ++k)
            out[n++] = hex[rng_below(r, 16)];
        break;
    case H_QMARK:
        out[n++] = hex[rng_below(r, 16)];
        break;
    case H_SET: {
        const char *m = g_sets[h->arg].members;
        out[n++] = m[rng_below(r, (uint32_t)strlen(m))];
        break;
    }
    case H_NAME: {
        const char *w = g_words[rng_below(r, N_WORDS)];
        size_t wl = strlen(w);
        memcpy(out, w, wl);
        n = (uint32_t)wl;
        break;
    }
    default: { /* H_TAIL: 0..3 hex digits, often nothing */
        uint32_t k = rng_below(r, 8);
        k = k < 5 ? 0 : k - 4;
        for (uint32_t q = 0; q < k; ++q)
            out[n++] = hex[rng_below(r, 16)];
        break;
    }
    }
    return n;
}

static uint32_t gen_string(uint64_t seed, uint32_t P_total, uint64_t i, char *out)
{
    vendors_init();
    rng_t r = rng_stream(seed, K_STRING, i);
    uint32_t n = 0;
    if (P_total && rng_below(&r, 100) < 70) {
        /* an instance of one table pattern: at least one match is guaranteed */
        pat_t p;
        gen_pattern(seed, rng_below(&r, P_total), &p);
        uint32_t at = 0;
        for (uint32_t h = 0; h < p.nholes; ++h) {
            memcpy(out + n, p.text + at, p.holes[h].pos - at);
            n += p.holes[h].pos - at;
            n += fill_hole(&p.holes[h], &r, out + n);
            at = p.holes[h].pos + p.holes[h].plen;
        }
        memcpy(out + n, p.text + at, p.len - at);
        n += p.len - at;
        return n;
    }
    /* a device nobody has an alias for (most of these miss) */
    uint32_t v = pick_vendor(&r), d = (uint32_t)rng_next(&r) & 0xFFFF;
    switch (pick_bus(&r)) {
    case BUS_PCI:
        n = (uint32_t)sprintf(out, "pci:v%08Xd%08Xsv%08Xsd%08Xbc%02Xsc%02Xi%02X", v, d, pick_vendor(&r),
                              (uint32_t)rng_next(&r) & 0xFFFF, rng_below(&r, 0x14), rng_below(&r, 8),
                              rng_below(&r, 4));
        break;
    case BUS_USB:
        n = (uint32_t)sprintf(out, "usb:v%04Xp%04Xd%04Xdc%02Xdsc%02Xdp%02Xic%02Xisc%02Xip%02Xin%02X", v, d,
                              (uint32_t)rng_next(&r) & 0xFFFF, rng_below(&r, 0x10), rng_below(&r, 8),
                              rng_below(&r, 4), rng_below(&r, 0x10), rng_below(&r, 8), rng_below(&r, 4),
                              rng_below(&r, 4));
        break;
    case BUS_HID:
        n = (uint32_t)sprintf(out, "hid:b%04Xg%04Xv%08Xp%08X", 3 + rng_below(&r, 3), 1 + rng_below(&r, 4), v, d);
        break;
    case BUS_PLATFORM:
        n = (uint32_t)sprintf(out, "platform:%s_%s%u", g_vendors_of[v % N_OFV], g_words[d % N_WORDS], (d >> 8) % 16);
        break;
    case BUS_OF:
        n = (uint32_t)sprintf(out, "of:N%sT%sC%s,%s%u", g_words[rng_below(&r, N_WORDS)],
                              g_words[rng_below(&r, N_WORDS)], g_vendors_of[v % N_OFV], g_words[d % N_WORDS],
                              (d >> 8) % 64);
        break;
    case BUS_SERIO:
        n = (uint32_t)sprintf(out, "serio:ty%02Xpr%02Xid%02Xex%02X", rng_below(&r, 8), rng_below(&r, 0x40),
                              rng_below(&r, 0x100), rng_below(&r, 0x100));
        break;
    default:
        n = (uint32_t)sprintf(out, "acpi:PNP%04X:", d & 0x0FFF);
        break;
    }
    return n;
}

/* ---- exported generators --------------------------------------------------
 * Two-call protocol: with arena == NULL only sizes are computed and the byte
 * count is returned; with a buffer of at least that capacity the data is
 * written.  Offsets are relative to the first item of the requested range. */

uint64_t mcxs_alias_table(uint64_t seed, uint32_t j0, uint32_t j1, char *arena, uint64_t cap,
                          uint32_t *a_off, uint32_t *a_mod)
{
    uint64_t w = 0;
    pat_t p;
    for (uint32_t j = j0; j < j1; ++j) {
        gen_pattern(seed, j, &p);
        if (arena && w + p.len <= cap)
            memcpy(arena + w, p.text, p.len);
        if (a_off)
            a_off[j - j0] = (uint32_t)w;
        if (a_mod)
            a_mod[j - j0] = module_of_pattern(j);
        w += p.len;
    }
    if (a_off)
        a_off[j1 - j0] = (uint32_t)w;
    return w;
}

uint32_t mcxs_alias_modules(uint32_t P_total)
{
    return P_total ? module_of_pattern(P_total - 1) + 1 : 0;
}

uint64_t mcxs_modalias_strings(uint64_t seed, uint32_t P_total, uint64_t i0, uint64_t i1, char *S,
                               uint64_t cap, uint32_t *s_off)
{
    uint64_t w = 0;
    char buf[256];
    for (uint64_t i = i0; i < i1; ++i) {
        uint32_t n = gen_string(seed, P_total, i, buf);
        if (S && w + n <= cap)
            memcpy(S + w, buf, n);
        if (s_off)
            s_off[i - i0] = (uint32_t)w;
        w += n;
    }
    if (s_off)
        s_off[i1 - i0] = (uint32_t)w;
    return w;
}

/*
 * Direct-dependency DAG: vertex u needs a geometric number (mean mean_deg) of
 * lower-numbered vertices, drawn with a strong bias towards small ids (a few
 * "core" modules are held by very many).  Rows are sorted and free of
 * duplicates.  Call with col == NULL to count edges.
 */
static uint32_t gen_row(uint64_t seed, uint32_t u, uint32_t mean_deg, uint32_t *out, uint32_t cap)
{
    if (u == 0)
        return 0;
    rng_t r = rng_stream(seed, K_VERTEX, u);
    uint32_t d = 0;
    while (d < cap && rng_below(&r, mean_deg + 1) != 0)
        ++d;
    uint32_t n = 0;
    for (uint32_t k = 0; k < d; ++k) {
        /* product of two uniforms: density -ln(x), mass near 0; one edge in
         * eight goes to a small set of "core" modules instead (the first few
         * hundred ids, again biased towards 0), which gives hubs with 10^4
         * and more holders at V = 10^6 */
        uint64_t a = rng_next(&r) >> 32, b = rng_next(&r) >> 32;
        uint64_t f = (a * b) >> 32;
        uint32_t span = u;
        if ((rng_next(&r) & 7u) == 0u && u > 512u)
            span = 512u;
        uint32_t v = (uint32_t)(((uint64_t)span * f) >> 32);
        uint32_t q = 0;
        while (q < n && out[q] < v)
            ++q;
        if (q < n && out[q] == v)
            continue;
        memmove(out + q + 1, out + q, sizeof(uint32_t) * (n - q));
        out[q] = v;
        ++n;
    }
    return n;
}

uint64_t mcxs_dep_graph(uint64_t seed, uint32_t V, uint32_t mean_deg, uint32_t *rowptr, uint32_t *col,
                        uint64_t cap_edges)
{
    uint32_t row[256];
    uint64_t e = 0;
    for (uint32_t u = 0; u < V; ++u) {
        uint32_t n = gen_row(seed, u, mean_deg, row, 255);
        if (rowptr)
            rowptr[u] = (uint32_t)e;
        if (col && e + n <= cap_edges)
            memcpy(col + e, row, sizeof(uint32_t) * n);
        e += n;
    }
    if (rowptr)
        rowptr[V] = (uint32_t)e;
    return e;
}

/*
 * Per-module state: LOADED = a Bernoulli(load_pm/1000) draw, closed under
 * dependencies (a loaded module's dependencies are loaded); BOUND = 10 % of
 * the loaded modules nobody holds; SAFE = 70 %.
 * Bits: LOADED 1, BOUND 2, SAFE 4.
 */
void mcxs_dep_flags(uint64_t seed, const uint32_t *rowptr, const uint32_t *col, uint32_t V,
                    uint32_t load_pm, uint8_t *in_flags)
{
    uint8_t *held = (uint8_t *)calloc(V ? V : 1, 1);
    for (uint32_t u = 0; u < V; ++u) {
        rng_t r = rng_stream(seed, K_FLAGS, u);
        uint8_t f = 0;
        if (rng_below(&r, 1000) < load_pm)
            f |= 1;
        if (rng_below(&r, 10) < 7)
            f |= 4;
        in_flags[u] = f;
    }
    /* edges point to lower ids, so one descending sweep closes the set */
    for (uint32_t u = V; u-- > 0;) {
        if (!(in_flags[u] & 1))
            continue;
        for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
            in_flags[col[k]] |= 1;
            held[col[k]] = 1;
        }
    }
    for (uint32_t u = 0; u < V; ++u) {
        rng_t r = rng_stream(seed, K_FLAGS, (uint64_t)u + ((uint64_t)1 << 40));
        if ((in_flags[u] & 1) && !held[u] && rng_below(&r, 10) == 0)
            in_flags[u] |= 2;
    }
    free(held);
}

/* n_seeds distinct loaded modules (fewer if not that many are loaded) */
uint32_t mcxs_seed_bitmap(uint64_t seed, uint32_t V, uint32_t n_seeds, const uint8_t *in_flags,
                          uint32_t *bitmap)
{
    uint32_t W = (V + 31) / 32, got = 0;
    memset(bitmap, 0, sizeof(uint32_t) * W);
    if (!V)
        return 0;
    rng_t r = rng_stream(seed, K_SEEDSET, 0);
    for (uint64_t tries = 0; got < n_seeds && tries < (uint64_t)n_seeds * 64 + 1024; ++tries) {
        uint32_t v = rng_below(&r, V);
        if (!(in_flags[v] & 1) || (bitmap[v >> 5] >> (v & 31) & 1u))
            continue;
        bitmap[v >> 5] |= 1u << (v & 31);
        ++got;
    }
    return got;
}

/*
 * Module names: "<stem>_<hex>" where a quarter of the stems are names or
 * prefixes the reference's safe-module filter refuses (snd_, nf_, ext4 ...),
 * so that the filter has something to do.  Two-call protocol as above.
 */
uint64_t mcxs_module_names(uint64_t seed, uint32_t m0, uint32_t m1, char *buf, uint64_t cap, uint32_t *off)
{
    static const char *const risky[] = {"snd_", "nf_", "xt_", "hid_", "intel_", "amd_", "i2c_", "sha", "ac",
                                        "st", "sg", "fan", "vbox", "crc32", "acpi", "thermal"};
    static const char *const exact[] = {"ext4", "kvm", "loop", "bridge", "usbhid", "snd", "fuse", "tun"};
    static const char *const fine[] = {"e1000", "r8169", "iwlwifi", "ath9k", "nouveau", "radeon", "xhci_pci",
                                       "ahci", "igb", "tg3", "bnx2", "uvcvideo", "btusb", "rtw88_8821cu",
                                       "rtl8xxxu", "mt76", "nvme", "sdhci", "ixgbe", "mlx5_core"};
    uint64_t w = 0;
    char name[64];
    for (uint32_t m = m0; m < m1; ++m) {
        rng_t r = rng_stream(seed, K_NAME, m);
        uint32_t x = rng_below(&r, 100);
        int n;
        if (x < 5)
            n = snprintf(name, sizeof name, "%s", exact[rng_below(&r, 8)]);
        else if (x < 25)
            n = snprintf(name, sizeof name, "%s%x", risky[rng_below(&r, 16)], m);
        else
            n = snprintf(name, sizeof name, "%s_%x", fine[rng_below(&r, 20)], m);
        if (buf && w + (uint64_t)n <= cap)
            memcpy(buf + w, name, (size_t)n);
        if (off)
            off[m - m0] = (uint32_t)w;
        w += (uint64_t)n;
    }
    if (off)
        off[m1 - m0] = (uint32_t)w;
    return w;
}

/*
 * A fleet of hosts for the inventory pipeline.  Host h has
 * mean_dev/2 + (draw below mean_dev + 1) devices (the strings of the batch are
 * dealt out to the hosts in order) and up to max_extra extra loaded modules
 * drawn uniformly from [0, V) (duplicates possible, as in a list collected
 * from several sources).  dev_cnt and extra_cnt receive h1 - h0 counts;
 * extra_ids (may be NULL to count) the lists back to back.  Returns the
 * number of extra entries.
 */
uint64_t mcxs_fleet_hosts(uint64_t seed, uint32_t V, uint64_t h0, uint64_t h1, uint32_t mean_dev,
                          uint32_t max_extra, uint32_t *dev_cnt, uint32_t *extra_cnt, uint32_t *extra_ids,
                          uint64_t cap)
{
    uint64_t w = 0;
    for (uint64_t h = h0; h < h1; ++h) {
        rng_t r = rng_stream(seed, K_HOST, h);
        uint32_t nd = mean_dev / 2 + rng_below(&r, mean_dev + 1);
        uint32_t nx = V ? rng_below(&r, max_extra + 1) : 0;
        if (dev_cnt)
            dev_cnt[h - h0] = nd;
        if (extra_cnt)
            extra_cnt[h - h0] = nx;
        for (uint32_t k = 0; k < nx; ++k) {
            uint32_t m = rng_below(&r, V);
            if (extra_ids && w < cap)
                extra_ids[w] = m;
            ++w;
        }
    }
    return w;
}
