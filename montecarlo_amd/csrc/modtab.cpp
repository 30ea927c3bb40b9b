/*
 * modtab.cpp - text loaders: the kernel's modules.alias / modules.dep files and
 * /proc/modules, parsed into the arenas and CSR arrays the batch entry points
 * take (include/mcx.h, "text tables").  Host only.
 *
 * The reference never reads these files itself: it asks kmod, one module at a
 * time (modprobe at montecarlo/libmontecarlo.c:123, `modinfo -F alias` at
 * desktop/ui.py:1527), and reads /proc/modules one token per line
 * (montecarlo/libmontecarlo.c:499-532).  File formats (kernel depmod output):
 *   modules.alias   "alias <glob> <module>" per line, '#' comment lines
 *   modules.dep     "<path>/<file>.ko[.xz|.zst|.gz]: <path>/<dep>.ko ..." per
 *                   line: ALL modules the first one needs, directly or not, in
 *                   load order
 *   /proc/modules   "<name> <size> <refcnt> <holders> <state> <address>"
 * A module's name is its file's base name up to the first '.', with '-'
 * written as '_' (the rule the reference applies at desktop/ui.py:481-482).
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mcx.h"

struct mcx_modtab {
    std::vector<std::string> names;                  /* module id -> name */
    std::unordered_map<std::string, uint32_t> ids;
    /* alias table, file order */
    std::string arena;
    std::vector<uint32_t> a_off{0};
    std::vector<uint32_t> a_mod;
    /* per module: everything modules.dep lists for it (ids, file order) */
    std::vector<std::vector<uint32_t>> deps;
    std::vector<uint8_t> has_dep_line;
    /* CSR built on demand */
    bool csr_direct = false, csr_valid = false;
    std::vector<uint32_t> rowptr, col;
    std::string err;
};

namespace {

std::string module_name_of(const char *p, size_t n)
{
    /* base name, cut at the first '.', '-' -> '_' */
    size_t b = n;
    while (b > 0 && p[b - 1] != '/')
        --b;
    std::string s;
    for (size_t i = b; i < n && p[i] != '.'; ++i)
        s.push_back(p[i] == '-' ? '_' : p[i]);
    return s;
}

uint32_t intern(mcx_modtab *t, const std::string &name)
{
    auto it = t->ids.find(name);
    if (it != t->ids.end())
        return it->second;
    uint32_t id = (uint32_t)t->names.size();
    t->names.push_back(name);
    t->ids.emplace(name, id);
    t->deps.emplace_back();
    t->has_dep_line.push_back(0);
    t->csr_valid = false;
    return id;
}

bool is_blank(char c)
{
    return c == ' ' || c == '\t' || c == '\r';
}

/* next whitespace-separated token of [p, e) */
bool next_token(const char *&p, const char *e, const char *&tok, size_t &len)
{
    while (p < e && is_blank(*p))
        ++p;
    if (p >= e)
        return false;
    tok = p;
    while (p < e && !is_blank(*p))
        ++p;
    len = (size_t)(p - tok);
    return true;
}

template <class F>
void for_each_line(const char *text, size_t len, F f)
{
    const char *p = text, *end = text + len;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl : end;
        f(p, e);
        p = nl ? nl + 1 : end;
    }
}

void build_csr(mcx_modtab *t, bool direct)
{
    const uint32_t V = (uint32_t)t->names.size();
    t->rowptr.assign((size_t)V + 1, 0);
    t->col.clear();
    std::vector<uint32_t> mark(V, 0xFFFFFFFFu), via(V, 0xFFFFFFFFu);
    for (uint32_t u = 0; u < V; ++u) {
        t->rowptr[u] = (uint32_t)t->col.size();
        std::vector<uint32_t> row;
        /* the file's list without repeats and without u itself */
        for (uint32_t d : t->deps[u])
            if (d != u && mark[d] != u) {
                mark[d] = u;
                row.push_back(d);
            }
        if (direct) {
            /* modules.dep lists the closure: d is kept unless another listed module
             * needs it too (it is then in that module's own list).  This is the
             * transitive reduction of the file's lists: a module that uses symbols of
             * both B and C, where B uses C as well, is a holder of C in sysfs but has
             * no edge to C here.  A holder being loaded implies that everything between
             * it and C is loaded, so has-holders, in-use and root come out the same;
             * holder counts can be smaller than /sys/module/<m>/refcnt. */
            for (uint32_t o : row)
                for (uint32_t d : t->deps[o])
                    if (d != o)
                        via[d] = u;
            std::vector<uint32_t> keep;
            for (uint32_t d : row)
                if (via[d] != u)
                    keep.push_back(d);
            row.swap(keep);
        }
        std::sort(row.begin(), row.end());
        t->col.insert(t->col.end(), row.begin(), row.end());
    }
    t->rowptr[V] = (uint32_t)t->col.size();
    t->csr_direct = direct;
    t->csr_valid = true;
}

int read_file(const std::string &path, std::string &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return -1;
    char buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0)
        out.append(buf, n);
    fclose(f);
    return 0;
}

} // namespace

extern "C" {

mcx_modtab *mcx_modtab_create(void)
{
    return new (std::nothrow) mcx_modtab();
}

void mcx_modtab_destroy(mcx_modtab *t)
{
    delete t;
}

const char *mcx_modtab_last_error(mcx_modtab *t)
{
    return t ? t->err.c_str() : "null table";
}

int mcx_modtab_add_alias_text(mcx_modtab *t, const char *text, size_t len)
{
    if (!t || (len && !text))
        return MCX_E_BAD_ARG;
    int rc = MCX_OK;
    size_t lineno = 0;
    for_each_line(text, len, [&](const char *p, const char *e) {
        ++lineno;
        const char *tok;
        size_t n;
        if (rc != MCX_OK || !next_token(p, e, tok, n) || tok[0] == '#')
            return;
        if (n != 5 || memcmp(tok, "alias", 5) != 0)
            return; /* other keywords of a modprobe.d-style file are not aliases */
        const char *pat, *mod;
        size_t pn, mn;
        if (!next_token(p, e, pat, pn) || !next_token(p, e, mod, mn)) {
            t->err = "line " + std::to_string(lineno) + ": alias needs a pattern and a module";
            rc = MCX_E_BAD_ARG;
            return;
        }
        if (t->arena.size() + pn > 0xFFFFFFFFull) {
            t->err = "alias text beyond 4 GiB";
            rc = MCX_E_TOO_LARGE;
            return;
        }
        t->arena.append(pat, pn);
        t->a_off.push_back((uint32_t)t->arena.size());
        t->a_mod.push_back(intern(t, module_name_of(mod, mn)));
    });
    return rc;
}

int mcx_modtab_add_dep_text(mcx_modtab *t, const char *text, size_t len)
{
    if (!t || (len && !text))
        return MCX_E_BAD_ARG;
    int rc = MCX_OK;
    size_t lineno = 0;
    for_each_line(text, len, [&](const char *p, const char *e) {
        ++lineno;
        const char *tok;
        size_t n;
        if (rc != MCX_OK || !next_token(p, e, tok, n) || tok[0] == '#')
            return;
        if (tok[n - 1] != ':') {
            t->err = "line " + std::to_string(lineno) + ": no ':' after the module path";
            rc = MCX_E_BAD_ARG;
            return;
        }
        const uint32_t u = intern(t, module_name_of(tok, n - 1));
        t->has_dep_line[u] = 1;
        t->deps[u].clear();
        while (next_token(p, e, tok, n)) {
            const uint32_t d = intern(t, module_name_of(tok, n)); /* may grow t->deps */
            t->deps[u].push_back(d);
        }
        t->csr_valid = false;
    });
    return rc;
}

int mcx_modtab_load_dir(mcx_modtab *t, const char *dir)
{
    if (!t || !dir)
        return MCX_E_BAD_ARG;
    std::string text;
    const std::string base(dir);
    if (read_file(base + "/modules.alias", text) != 0) {
        t->err = "cannot read " + base + "/modules.alias";
        return MCX_E_BAD_ARG;
    }
    int rc = mcx_modtab_add_alias_text(t, text.data(), text.size());
    if (rc != MCX_OK)
        return rc;
    if (read_file(base + "/modules.dep", text) != 0) {
        t->err = "cannot read " + base + "/modules.dep";
        return MCX_E_BAD_ARG;
    }
    return mcx_modtab_add_dep_text(t, text.data(), text.size());
}

int mcx_modtab_counts(mcx_modtab *t, uint32_t *patterns, uint32_t *modules, uint64_t *alias_bytes)
{
    if (!t)
        return MCX_E_BAD_ARG;
    if (patterns)
        *patterns = (uint32_t)t->a_mod.size();
    if (modules)
        *modules = (uint32_t)t->names.size();
    if (alias_bytes)
        *alias_bytes = t->arena.size();
    return MCX_OK;
}

int mcx_modtab_alias_arrays(mcx_modtab *t, const char **arena, const uint32_t **a_off, const uint32_t **a_mod)
{
    if (!t || !arena || !a_off || !a_mod)
        return MCX_E_BAD_ARG;
    *arena = t->arena.data();
    *a_off = t->a_off.data();
    *a_mod = t->a_mod.data();
    return MCX_OK;
}

int mcx_modtab_dep_arrays(mcx_modtab *t, int direct, const uint32_t **rowptr, const uint32_t **col,
                          uint32_t *edges)
{
    if (!t || !rowptr || !col)
        return MCX_E_BAD_ARG;
    if (!t->csr_valid || t->csr_direct != (direct != 0))
        build_csr(t, direct != 0);
    *rowptr = t->rowptr.data();
    *col = t->col.data();
    if (edges)
        *edges = (uint32_t)t->col.size();
    return MCX_OK;
}

const char *mcx_modtab_module_name(mcx_modtab *t, uint32_t id)
{
    return (t && id < t->names.size()) ? t->names[id].c_str() : nullptr;
}

uint32_t mcx_modtab_module_id(mcx_modtab *t, const char *name)
{
    if (!t || !name)
        return MCX_NO_MATCH;
    auto it = t->ids.find(module_name_of(name, strlen(name)));
    return it == t->ids.end() ? MCX_NO_MATCH : it->second;
}

int mcx_modtab_module_names(mcx_modtab *t, char *names, uint64_t cap, uint32_t *name_off, uint64_t *bytes_out)
{
    if (!t || !bytes_out)
        return MCX_E_BAD_ARG;
    uint64_t w = 0;
    for (size_t m = 0; m < t->names.size(); ++m) {
        if (name_off)
            name_off[m] = (uint32_t)w;
        if (names && w + t->names[m].size() <= cap)
            memcpy(names + w, t->names[m].data(), t->names[m].size());
        w += t->names[m].size();
    }
    if (name_off)
        name_off[t->names.size()] = (uint32_t)w;
    *bytes_out = w;
    return (names && w > cap) ? MCX_E_TOO_LARGE : MCX_OK;
}

int mcx_modtab_mark_loaded(mcx_modtab *t, const char *proc_modules, size_t len, uint8_t *in_flags,
                           uint32_t *unknown)
{
    if (!t || (len && !proc_modules) || (!in_flags && !t->names.empty()))
        return MCX_E_BAD_ARG;
    uint32_t missing = 0;
    for_each_line(proc_modules, len, [&](const char *p, const char *e) {
        const char *tok;
        size_t n;
        if (!next_token(p, e, tok, n))
            return;
        auto it = t->ids.find(module_name_of(tok, n));
        if (it == t->ids.end())
            ++missing; /* out-of-tree module: not in this kernel's tables */
        else
            in_flags[it->second] |= MCX_F_LOADED;
    });
    if (unknown)
        *unknown = missing;
    return MCX_OK;
}

int mcx_alias_table_load_modtab(mcx_ctx *ctx, mcx_modtab *t, uint32_t flags)
{
    if (!ctx || !t)
        return MCX_E_BAD_ARG;
    return mcx_alias_table_load(ctx, t->arena.data(), t->a_off.data(), t->a_mod.data(), (uint32_t)t->a_mod.size(),
                                flags);
}

int mcx_dep_graph_load_modtab(mcx_ctx *ctx, mcx_modtab *t)
{
    if (!ctx || !t)
        return MCX_E_BAD_ARG;
    if (!t->csr_valid || !t->csr_direct)
        build_csr(t, true);
    return mcx_dep_graph_load(ctx, t->rowptr.data(), t->col.data(), (uint32_t)t->names.size(),
                              (uint32_t)t->col.size());
}

} /* extern "C" */
