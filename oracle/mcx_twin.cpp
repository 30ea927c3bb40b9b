/*
 * mcx_twin.cpp - CPU twin of the device matcher.  TEST INFRASTRUCTURE ONLY.
 *
 * Builds the product's own host pattern compiler (glob_compile.cpp) and runs
 * the product's matcher routine (mcx_prog.h) on the host, so that the pattern
 * compiler and the program semantics can be checked against glibc fnmatch
 * (mcx_oracle.c) on a machine without a GPU.  The product library never calls
 * this; it has no CPU path.
 */
#include <stdint.h>
#include <string.h>
#include <string>

#include "../include/mcx.h"
#include "../montecarlo_amd/csrc/glob_compile.h"

namespace {
struct HostStr {
    const uint8_t *p;
    uint32_t operator[](uint32_t i) const { return p[i]; }
};

static void normalize(std::string &s, bool *bad)
{
    *bad = false;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '-') {
            s[i] = '_';
        } else if (s[i] == ']') {
            *bad = true;
            return;
        } else if (s[i] == '[') {
            while (i < s.size() && s[i] != ']')
                ++i;
            if (i >= s.size()) {
                *bad = true;
                return;
            }
        }
    }
}
} // namespace

extern "C" {

/* 0 = compiles, MCX_E_PATTERN = rejected as outside the supported grammar */
int twin_compile_check(const char *pattern, uint32_t n, int normalize_dash)
{
    mcx_compiled_table t;
    std::string err;
    return mcx_compile_pattern((const uint8_t *)pattern, n, normalize_dash != 0, t, err);
}

/* 1 = match, 0 = no match, <0 = pattern rejected */
int twin_fnmatch(const char *pattern, uint32_t pn, const char *string, uint32_t sn)
{
    mcx_compiled_table t;
    std::string err;
    int rc = mcx_compile_pattern((const uint8_t *)pattern, pn, false, t, err);
    if (rc != 0)
        return rc;
    HostStr s{(const uint8_t *)string};
    return mcx_prog_match(t.prog.data() + t.prog_off[0], s, sn,
                          t.sets.empty() ? nullptr : t.sets.data())
               ? 1
               : 0;
}

/* Same contract as orc_alias_match; returns 0 or a negative MCX_E_* code.
 * The prefilter record is applied first, as the device does, so that it is
 * covered too. */
int twin_alias_match(const char *arena, const uint32_t *a_off, uint32_t P, const char *S,
                     const uint32_t *s_off, uint32_t i0, uint32_t i1, int normalize_dash,
                     uint32_t *first_idx, uint32_t *count)
{
    mcx_compiled_table t;
    std::string err;
    int rc = mcx_compile_table(arena, a_off, P, normalize_dash != 0, t, err);
    if (rc != 0)
        return rc;
    const uint32_t *sets = t.sets.empty() ? nullptr : t.sets.data();
    for (uint32_t i = i0; i < i1; ++i) {
        std::string str(S + s_off[i], S + s_off[i + 1]);
        bool bad = false;
        if (normalize_dash)
            normalize(str, &bad);
        uint32_t first = MCX_NO_MATCH, cnt = 0;
        if (!bad) {
            uint32_t len = (uint32_t)str.size();
            uint8_t head[16] = {0};
            memcpy(head, str.data(), len < 16 ? len : 16);
            uint32_t h[4];
            memcpy(h, head, 16);
            HostStr s{(const uint8_t *)str.data()};
            for (uint32_t j = 0; j < P; ++j) {
                const mcx_prefix_t &pf = t.prefix[j];
                uint32_t d = ((h[0] ^ pf.pfx[0]) & pf.mask[0]) | ((h[1] ^ pf.pfx[1]) & pf.mask[1]) |
                             ((h[2] ^ pf.pfx[2]) & pf.mask[2]) | ((h[3] ^ pf.pfx[3]) & pf.mask[3]);
                if (d != 0 || len < MCX_META_MINLEN(t.meta[j]))
                    continue;
                if (mcx_prog_match(t.prog.data() + t.prog_off[j], s, len, sets)) {
                    if (first == MCX_NO_MATCH)
                        first = j;
                    ++cnt;
                }
            }
        }
        first_idx[i] = first;
        count[i] = cnt;
    }
    return 0;
}

} /* extern "C" */
