/*
 * montecarlo_cli - command line front end of libmontecarlo.so.
 *
 * The three subcommands of the reference's tool keep their output, messages
 * and exit codes (reference montecarlo/montecarlo.c:7-55):
 *   list              JSON array of the candidate driver names, one per line
 *   load <driver>     exit 0 when modprobe succeeded, else 1
 *   unload <driver>   always exit 0
 * and two more ask the batch path:
 *   match [-d <modules dir>] <modalias>...
 *       which module kmod would load for each modalias: first matching line
 *       of <dir>/modules.alias (default /lib/modules/<running kernel>), and
 *       how many lines match.  Runs the alias match on GPU 0.
 *   roots [-d <modules dir>] [-p <proc modules file>]
 *       the dashboard's module list for this host (reference
 *       desktop/ui.py:1565-1599) without its "used driver" column: loaded
 *       modules with a safe name and a hardware alias that nothing holds, each
 *       with the dashboard's "(In Use)" / "(Idle)" tag (ui.py:1579-1590).
 *   bench [-d <modules dir>] [-n <strings>] [-r <repetitions>]
 *       times the batch path on this host's tables: <strings> modalias strings
 *       made from the lines of modules.alias (every wildcard filled in) against
 *       the whole table, and the dependency flags over modules.dep with every
 *       module taken as loaded; device time per call on GPU 0, one JSON line.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/utsname.h>

#include "../../include/libmontecarlo.h"
#include "../../include/mcx.h"

static int usage(const char *prog, const char *sub)
{
    if (sub)
        fprintf(stderr, "Uso: %s %s <driver>\n", prog, sub);
    else
        fprintf(stderr, "Uso: %s [list|load <driver>|unload <driver>]\n", prog);
    return 1;
}

static int cmd_list(void)
{
    enum { SLOTS = 256 };
    static char names[SLOTS][128];
    const int n = mc_list_candidate_drivers(names, SLOTS);
    puts("[");
    for (int i = 0; i < n; ++i)
        printf("  \"%s\"%s\n", names[i], i + 1 < n ? "," : "");
    puts("]");
    return 0;
}

static int default_dir(char *out, size_t cap)
{
    struct utsname u;
    if (uname(&u) != 0)
        return -1;
    snprintf(out, cap, "/lib/modules/%s", u.release);
    return 0;
}

static char *read_file(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return NULL;
    size_t cap = 1 << 16, n = 0;
    char *buf = malloc(cap);
    while (buf) {
        n += fread(buf + n, 1, cap - n, f);
        if (n < cap)
            break;
        cap *= 2;
        char *bigger = realloc(buf, cap);
        if (!bigger) {
            free(buf);
            buf = NULL;
        } else {
            buf = bigger;
        }
    }
    fclose(f);
    *len = n;
    return buf;
}

/* options shared by match and roots; returns the index of the first operand */
static int parse_opts(int argc, char **argv, int at, const char **dir, const char **proc)
{
    while (at + 1 < argc && argv[at][0] == '-' && argv[at][1] && !argv[at][2]) {
        if (argv[at][1] == 'd')
            *dir = argv[at + 1];
        else if (argv[at][1] == 'p' && proc)
            *proc = argv[at + 1];
        else
            break;
        at += 2;
    }
    return at;
}

static int fail_tab(mcx_modtab *t, const char *what)
{
    fprintf(stderr, "%s: %s\n", what, mcx_modtab_last_error(t));
    mcx_modtab_destroy(t);
    return 2;
}

static int fail_ctx(mcx_ctx *ctx, mcx_modtab *t, const char *what, int rc)
{
    fprintf(stderr, "%s: %s (%s)\n", what, mcx_strerror(rc), ctx ? mcx_last_error(ctx) : "no context");
    if (ctx)
        mcx_ctx_destroy(ctx);
    mcx_modtab_destroy(t);
    return 2;
}

static int cmd_match(int argc, char **argv)
{
    char dflt[300];
    const char *dir = NULL;
    int at = parse_opts(argc, argv, 2, &dir, NULL);
    if (at >= argc) {
        fprintf(stderr, "Uso: %s match [-d <dir>] <modalias>...\n", argv[0]);
        return 1;
    }
    if (!dir) {
        if (default_dir(dflt, sizeof dflt) != 0)
            return 2;
        dir = dflt;
    }
    mcx_modtab *t = mcx_modtab_create();
    if (!t)
        return 2;
    if (mcx_modtab_load_dir(t, dir) != MCX_OK)
        return fail_tab(t, dir);
    mcx_ctx *ctx = NULL;
    int rc = mcx_ctx_create(0, NULL, &ctx);
    if (rc != MCX_OK)
        return fail_ctx(NULL, t, "mcx_ctx_create", rc);
    /* kmod compares with '-' read as '_' on both sides */
    if ((rc = mcx_alias_table_load_modtab(ctx, t, MCX_TABLE_NORMALIZE_DASH)) != MCX_OK)
        return fail_ctx(ctx, t, "mcx_alias_table_load_modtab", rc);
    const uint32_t n = (uint32_t)(argc - at);
    uint32_t *off = calloc((size_t)n + 1, 4), *first = calloc(n, 4), *count = calloc(n, 4);
    size_t bytes = 0;
    for (uint32_t i = 0; i < n; ++i)
        bytes += strlen(argv[at + i]);
    char *arena = malloc(bytes ? bytes : 1);
    if (!off || !first || !count || !arena)
        return 2;
    for (uint32_t i = 0; i < n; ++i) {
        const size_t len = strlen(argv[at + i]);
        memcpy(arena + off[i], argv[at + i], len);
        off[i + 1] = off[i] + (uint32_t)len;
    }
    if ((rc = mcx_alias_match(ctx, arena, off, n, first, count)) != MCX_OK)
        return fail_ctx(ctx, t, "mcx_alias_match", rc);
    const uint32_t *a_mod;
    const uint32_t *a_off;
    const char *a_arena;
    mcx_modtab_alias_arrays(t, &a_arena, &a_off, &a_mod);
    int missed = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (first[i] == MCX_NO_MATCH) {
            printf("%s: no module\n", argv[at + i]);
            missed = 1;
        } else {
            printf("%s: %s (%u matching alias%s, first at line entry %u)\n", argv[at + i],
                   mcx_modtab_module_name(t, a_mod[first[i]]), count[i], count[i] == 1 ? "" : "es", first[i]);
        }
    }
    free(arena);
    free(off);
    free(first);
    free(count);
    mcx_ctx_destroy(ctx);
    mcx_modtab_destroy(t);
    return missed;
}

static int cmd_roots(int argc, char **argv)
{
    char dflt[300];
    const char *dir = NULL, *proc = "/proc/modules";
    (void)parse_opts(argc, argv, 2, &dir, &proc);
    if (!dir) {
        if (default_dir(dflt, sizeof dflt) != 0)
            return 2;
        dir = dflt;
    }
    mcx_modtab *t = mcx_modtab_create();
    if (!t)
        return 2;
    if (mcx_modtab_load_dir(t, dir) != MCX_OK)
        return fail_tab(t, dir);
    size_t plen = 0;
    char *ptext = read_file(proc, &plen);
    if (!ptext) {
        fprintf(stderr, "%s: cannot read\n", proc);
        mcx_modtab_destroy(t);
        return 2;
    }
    uint32_t P = 0, V = 0;
    mcx_modtab_counts(t, &P, &V, NULL);
    uint8_t *in = calloc(V ? V : 1, 1), *out = calloc(V ? V : 1, 1), *safe = calloc(V ? V : 1, 1),
            *hw = calloc(V ? V : 1, 1);
    uint32_t *name_off = calloc((size_t)V + 1, 4), unknown = 0;
    uint64_t nbytes = 0;
    mcx_modtab_module_names(t, NULL, 0, NULL, &nbytes);
    char *names = malloc(nbytes ? nbytes : 1);
    if (!in || !out || !safe || !hw || !name_off || !names)
        return 2;
    mcx_modtab_module_names(t, names, nbytes, name_off, &nbytes);
    mcx_safe_module_names(names, name_off, V, safe);
    if (mcx_modtab_mark_loaded(t, ptext, plen, in, &unknown) != MCX_OK)
        return fail_tab(t, proc);
    mcx_ctx *ctx = NULL;
    int rc = mcx_ctx_create(0, NULL, &ctx);
    if (rc != MCX_OK)
        return fail_ctx(NULL, t, "mcx_ctx_create", rc);
    if ((rc = mcx_alias_table_load_modtab(ctx, t, MCX_TABLE_NORMALIZE_DASH)) != MCX_OK ||
        (rc = mcx_dep_graph_load_modtab(ctx, t)) != MCX_OK || (rc = mcx_alias_hw_flags(ctx, V, hw)) != MCX_OK)
        return fail_ctx(ctx, t, "loading the tables", rc);
    for (uint32_t v = 0; v < V; ++v)
        if (safe[v] && hw[v])
            in[v] |= MCX_F_SAFE;
    if ((rc = mcx_dep_flags(ctx, in, out, NULL)) != MCX_OK)
        return fail_ctx(ctx, t, "mcx_dep_flags", rc);
    uint32_t shown = 0;
    for (uint32_t v = 0; v < V; ++v)
        if (out[v] & MCX_O_ROOT) {
            /* the dashboard's status tag (reference desktop/ui.py:1579-1590): device links under the
             * module's name or its name variant on the pci, usb or pcmcia bus of THIS host */
            const char *name = mcx_modtab_module_name(t, v);
            printf("%s%s\n", name, mc_driver_is_in_use(name) ? " (In Use)" : " (Idle)");
            ++shown;
        }
    fprintf(stderr, "%u of %u modules listed (%u loaded modules are not in the tables)\n", shown, V, unknown);
    free(in);
    free(out);
    free(safe);
    free(hw);
    free(name_off);
    free(names);
    free(ptext);
    mcx_ctx_destroy(ctx);
    mcx_modtab_destroy(t);
    return 0;
}

/* a modalias that pattern `pat` matches: `*` becomes "0", `?` "0", a bracket
 * expression its first member, an escaped byte itself */
static size_t string_from_pattern(const char *pat, size_t len, char *out)
{
    size_t n = 0;
    for (size_t k = 0; k < len; ++k) {
        const char c = pat[k];
        if (c == '*' || c == '?') {
            out[n++] = '0';
        } else if (c == '\\' && k + 1 < len) {
            out[n++] = pat[++k];
        } else if (c == '[') {
            size_t e = k + 1;
            const int neg = e < len && (pat[e] == '!' || pat[e] == '^');
            if (neg)
                ++e;
            const size_t first = e;
            if (e < len)
                ++e; /* a leading ']' is a member */
            while (e < len && pat[e] != ']')
                ++e;
            if (e >= len) {
                out[n++] = c; /* unterminated: a literal '[' */
            } else {
                out[n++] = neg ? '~' : pat[first]; /* '~' is in no alias set worth the name */
                k = e;
            }
        } else {
            out[n++] = c;
        }
    }
    return n;
}

static int cmd_bench(int argc, char **argv)
{
    char dflt[300];
    const char *dir = NULL;
    uint32_t n = 100000, reps = 20;
    int at = 2;
    while (at + 1 < argc && argv[at][0] == '-' && argv[at][1] && !argv[at][2]) {
        if (argv[at][1] == 'd')
            dir = argv[at + 1];
        else if (argv[at][1] == 'n')
            n = (uint32_t)strtoul(argv[at + 1], NULL, 10);
        else if (argv[at][1] == 'r')
            reps = (uint32_t)strtoul(argv[at + 1], NULL, 10);
        else
            break;
        at += 2;
    }
    if (at < argc || n == 0 || reps == 0) {
        fprintf(stderr, "Uso: %s bench [-d <dir>] [-n <strings>] [-r <repetitions>]\n", argv[0]);
        return 1;
    }
    if (!dir) {
        if (default_dir(dflt, sizeof dflt) != 0)
            return 2;
        dir = dflt;
    }
    mcx_modtab *t = mcx_modtab_create();
    if (!t)
        return 2;
    if (mcx_modtab_load_dir(t, dir) != MCX_OK)
        return fail_tab(t, dir);
    uint32_t P = 0, V = 0;
    uint64_t alias_bytes = 0;
    mcx_modtab_counts(t, &P, &V, &alias_bytes);
    if (P == 0) {
        fprintf(stderr, "%s: no alias lines\n", dir);
        mcx_modtab_destroy(t);
        return 2;
    }
    const uint32_t *a_mod, *a_off;
    const char *a_arena;
    mcx_modtab_alias_arrays(t, &a_arena, &a_off, &a_mod);
    /* string i comes from line (i * 2654435761) mod P: spread over the table, repeatable */
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t j = (uint32_t)(((uint64_t)i * 2654435761u) % P);
        bytes += a_off[j + 1] - a_off[j];
    }
    char *arena = malloc(bytes ? bytes : 1);
    uint32_t *off = calloc((size_t)n + 1, 4), *first = calloc(n, 4);
    if (!arena || !off || !first || bytes > 0xFFFFFFFFu)
        return 2;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t j = (uint32_t)(((uint64_t)i * 2654435761u) % P);
        off[i + 1] = off[i] + (uint32_t)string_from_pattern(a_arena + a_off[j], a_off[j + 1] - a_off[j], arena + off[i]);
    }
    mcx_ctx *ctx = NULL;
    int rc = mcx_ctx_create(0, NULL, &ctx);
    if (rc != MCX_OK)
        return fail_ctx(NULL, t, "mcx_ctx_create", rc);
    if ((rc = mcx_alias_table_load_modtab(ctx, t, MCX_TABLE_NORMALIZE_DASH)) != MCX_OK)
        return fail_ctx(ctx, t, "mcx_alias_table_load_modtab", rc);
    void *d_S = NULL, *d_off = NULL, *d_first = NULL, *d_count = NULL;
    if ((rc = mcx_dev_alloc(ctx, off[n], &d_S)) != MCX_OK || (rc = mcx_dev_alloc(ctx, 4 * ((size_t)n + 1), &d_off)) != MCX_OK ||
        (rc = mcx_dev_alloc(ctx, 4 * (size_t)n, &d_first)) != MCX_OK ||
        (rc = mcx_dev_alloc(ctx, 4 * (size_t)n, &d_count)) != MCX_OK ||
        (rc = mcx_h2d(ctx, d_S, arena, off[n])) != MCX_OK || (rc = mcx_h2d(ctx, d_off, off, 4 * ((size_t)n + 1))) != MCX_OK)
        return fail_ctx(ctx, t, "device buffers", rc);
    float ms = 0.f;
    for (uint32_t r = 0; r < 3 && rc == MCX_OK; ++r)
        rc = mcx_alias_match_dev(ctx, d_S, d_off, n, d_first, d_count);
    if (rc == MCX_OK)
        rc = mcx_timer_begin(ctx);
    for (uint32_t r = 0; r < reps && rc == MCX_OK; ++r)
        rc = mcx_alias_match_dev(ctx, d_S, d_off, n, d_first, d_count);
    if (rc == MCX_OK)
        rc = mcx_timer_end(ctx, &ms);
    if (rc == MCX_OK)
        rc = mcx_d2h(ctx, first, d_first, 4 * (size_t)n);
    if (rc != MCX_OK)
        return fail_ctx(ctx, t, "mcx_alias_match_dev", rc);
    uint32_t matched = 0;
    for (uint32_t i = 0; i < n; ++i)
        matched += first[i] != MCX_NO_MATCH;
    const double match_ms = ms / reps;
    /* dependency flags with every module loaded */
    double flags_ms = -1.0;
    uint32_t E = 0;
    if (V) {
        const uint32_t *rowptr, *col;
        void *d_in = NULL, *d_out = NULL;
        uint8_t *in = malloc(V);
        if (!in)
            return 2;
        memset(in, MCX_F_LOADED, V);
        if (mcx_modtab_dep_arrays(t, 1, &rowptr, &col, &E) == MCX_OK && (rc = mcx_dep_graph_load_modtab(ctx, t)) == MCX_OK &&
            (rc = mcx_dev_alloc(ctx, V, &d_in)) == MCX_OK && (rc = mcx_dev_alloc(ctx, V, &d_out)) == MCX_OK &&
            (rc = mcx_h2d(ctx, d_in, in, V)) == MCX_OK) {
            for (uint32_t r = 0; r < 3 && rc == MCX_OK; ++r)
                rc = mcx_dep_flags_dev(ctx, d_in, d_out, NULL);
            if (rc == MCX_OK)
                rc = mcx_timer_begin(ctx);
            for (uint32_t r = 0; r < reps && rc == MCX_OK; ++r)
                rc = mcx_dep_flags_dev(ctx, d_in, d_out, NULL);
            if (rc == MCX_OK && (rc = mcx_timer_end(ctx, &ms)) == MCX_OK)
                flags_ms = ms / reps;
        }
        if (rc != MCX_OK)
            return fail_ctx(ctx, t, "dependency flags", rc);
        mcx_dev_free(ctx, d_in);
        mcx_dev_free(ctx, d_out);
        free(in);
    }
    printf("{\"dir\": \"%s\", \"patterns\": %u, \"modules\": %u, \"dep_edges\": %u, \"strings\": %u, "
           "\"matched\": %u, \"repetitions\": %u, \"alias_match_ms\": %.6f, \"strings_per_s\": %.6g, "
           "\"pairs_per_s\": %.6g, \"dep_flags_ms\": %.6f}\n",
           dir, P, V, E, n, matched, reps, match_ms, n / (match_ms * 1e-3), (double)n * P / (match_ms * 1e-3), flags_ms);
    mcx_dev_free(ctx, d_S);
    mcx_dev_free(ctx, d_off);
    mcx_dev_free(ctx, d_first);
    mcx_dev_free(ctx, d_count);
    free(arena);
    free(off);
    free(first);
    mcx_ctx_destroy(ctx);
    mcx_modtab_destroy(t);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2)
        return usage(argv[0], NULL);
    const char *cmd = argv[1];
    if (!strcmp(cmd, "list"))
        return cmd_list();
    if (!strcmp(cmd, "load") || !strcmp(cmd, "unload")) {
        if (argc < 3)
            return usage(argv[0], cmd);
        if (cmd[0] == 'l')
            return mc_try_load_driver(argv[2]) ? 0 : 1;
        mc_unload_driver(argv[2]);
        return 0;
    }
    if (!strcmp(cmd, "match"))
        return cmd_match(argc, argv);
    if (!strcmp(cmd, "roots"))
        return cmd_roots(argc, argv);
    if (!strcmp(cmd, "bench"))
        return cmd_bench(argc, argv);
    fprintf(stderr, "Comando desconocido: %s\n", cmd);
    return usage(argv[0], NULL);
}
