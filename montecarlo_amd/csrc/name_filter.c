/*
 * name_filter.c - the name stages of the reference's safe-module filter, as a
 * host function over a whole list of module names.
 *
 * Reference desktop/ui.py:1391-1518 (is_safe_module): a module is dropped when
 * its name is one of 113 core names (nine set displays, ui.py:1402-1470) or
 * starts with one of 56 prefixes (dangerous_patterns, ui.py:1486-1514, bare
 * str.startswith).  The third stage, "has a hardware alias", is
 * mcx_alias_hw_flags.  The tables are data (tables.c: the reference's values
 * unless a file replaced them); the lookup is a binary search over the sorted
 * core names and a first-byte bucket over the prefixes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mcx.h"
#include "tables.h"

int mcx_safe_table_sizes(uint32_t *n_core, uint32_t *n_prefix)
{
    const mcx_tables_view *t = mcx_tables_active();
    if (n_core)
        *n_core = t->count[MCX_TS_CORE];
    if (n_prefix)
        *n_prefix = t->count[MCX_TS_PREFIX];
    return MCX_OK;
}

static int is_core(const mcx_tables_view *t, const char *name, size_t len)
{
    const char *const *CORE_SORTED = t->item[MCX_TS_CORE];
    size_t lo = 0, hi = t->count[MCX_TS_CORE];
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
    const mcx_tables_view *t = mcx_tables_active();
    const char *const *PREFIXES = t->item[MCX_TS_PREFIX];
    const size_t N_PREFIX = t->count[MCX_TS_PREFIX];
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
        if (is_core(t, s, len)) {
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
