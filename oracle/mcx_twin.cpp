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
#include <vector>

#include "../include/mcx.h"
#include "../montecarlo_amd/csrc/glob_compile.h"

namespace {
/* bounded reads: bytes beyond the string read as 0 (they are always masked) */
struct HostStr {
    const uint8_t *p;
    uint32_t len;
    uint32_t byte(uint32_t q) const { return q < len ? p[q] : 0u; }
    uint32_t word(uint32_t q) const
    {
        return byte(q) | (byte(q + 1) << 8) | (byte(q + 2) << 16) | (byte(q + 3) << 24);
    }
};

struct HostProg {
    const uint32_t *p;
    uint32_t operator[](uint32_t w) const { return p[w]; }
};

struct HostOcc {
    uint64_t t[256];
    uint64_t operator[](uint32_t b) const { return t[b]; }
};

/* the device's choice: segment masks from the occurrence table for strings of
 * up to 64 bytes whose pattern sits in its slot, position-loop matcher for
 * longer strings and for bodies in the overflow array */
static bool match_like_device(const uint32_t *slot, const uint32_t *ext, const HostStr &s, uint32_t len,
                              const uint32_t *sets)
{
    const uint32_t hdr = slot[0];
    if (MCX_HDR_FLAGS(hdr) & MCX_PF_EXT)
        return mcx_prog_match(HostProg{ext}, slot[MCX_SLOT_DIR], hdr, s, len, sets);
    if (len > MCX_OCC_MAX_LEN)
        return mcx_prog_match(HostProg{slot}, MCX_SLOT_BODY, hdr, s, len, sets);
    HostOcc occ;
    memset(&occ, 0, sizeof occ);
    for (uint32_t q = 0; q < len; ++q)
        occ.t[s.byte(q)] |= (uint64_t)1 << q;
    return mcx_slot_match_occ(HostProg{slot}, occ, len, sets);
}

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
    mcx_pattern_ir ir;
    std::string err;
    return mcx_parse_pattern((const uint8_t *)pattern, n, normalize_dash != 0, t, ir, err);
}

/* 1 = match, 0 = no match, <0 = pattern rejected */
int twin_fnmatch(const char *pattern, uint32_t pn, const char *string, uint32_t sn)
{
    mcx_compiled_table t;
    mcx_pattern_ir ir;
    std::string err;
    int rc = mcx_parse_pattern((const uint8_t *)pattern, pn, false, t, ir, err);
    if (rc != 0)
        return rc;
    uint32_t slot[MCX_SLOT_WORDS + 4] = {0}; /* read-ahead room */
    std::vector<uint32_t> ext;
    mcx_fill_slot(ir, 0u, 0u, 1u, t.sets, slot, ext);
    ext.insert(ext.end(), 4, 0u);
    HostStr s{(const uint8_t *)string, sn};
    const uint32_t *sets = t.sets.empty() ? nullptr : t.sets.data();
    const bool is_ext = (MCX_HDR_FLAGS(slot[0]) & MCX_PF_EXT) != 0;
    int r = (is_ext ? mcx_prog_match(HostProg{ext.data()}, slot[MCX_SLOT_DIR], slot[0], s, sn, sets)
                    : mcx_prog_match(HostProg{slot}, MCX_SLOT_BODY, slot[0], s, sn, sets))
                ? 1
                : 0;
    if ((match_like_device(slot, ext.data(), s, sn, sets) ? 1 : 0) != r)
        return -100; /* the two matchers disagree */
    if (sn <= 64u) {
        /* the tile kernel's way: start positions of tag segments from the set of lower-case positions */
        const uint64_t lower = mcx_lower_set(s, sn);
        const bool r2 = is_ext ? mcx_prog_match(HostProg{ext.data()}, slot[MCX_SLOT_DIR], slot[0], s, sn, sets, lower, true)
                               : mcx_prog_match(HostProg{slot}, MCX_SLOT_BODY, slot[0], s, sn, sets, lower, true);
        if ((r2 ? 1 : 0) != r)
            return -101;
    }
    return r;
}

/* A compiled table kept between calls, so that the indexed CPU path can be
 * timed without the pattern compiler (the CPU baseline that is not the naive
 * double loop). */
struct twin_table {
    mcx_compiled_table t;
    bool normalize_dash;
};

/* threads / sub_min: the compiler's options (0 = its defaults) */
void *twin_table_create(const char *arena, const uint32_t *a_off, uint32_t P, int normalize_dash, uint32_t threads,
                        uint32_t sub_min)
{
    twin_table *tt = new twin_table();
    std::string err;
    tt->normalize_dash = normalize_dash != 0;
    mcx_compile_options opt;
    opt.threads = threads;
    opt.sub_min = sub_min;
    if (mcx_compile_table(arena, a_off, P, tt->normalize_dash, tt->t, err, &opt) != 0) {
        delete tt;
        return nullptr;
    }
    return tt;
}

/* compile only: the return code and the error text (for tests of the error paths) */
int twin_compile_error(const char *arena, const uint32_t *a_off, uint32_t P, uint32_t threads, char *msg, uint32_t cap)
{
    mcx_compiled_table t;
    std::string err;
    mcx_compile_options opt;
    opt.threads = threads;
    int rc = mcx_compile_table(arena, a_off, P, false, t, err, &opt);
    if (msg && cap) {
        strncpy(msg, err.c_str(), cap - 1);
        msg[cap - 1] = 0;
    }
    return rc;
}

void twin_table_destroy(void *h)
{
    delete (twin_table *)h;
}

/* FNV-1a digests of the arrays of a kept table, in the order slot, ext, sig, slots, tags, sub,
 * sub_tags, lens, sets, dup_start, dup_idx, hw_tag, then {n_distinct, n_sub_groups, max_group, P}:
 * what a change of the compiler (or of its number of threads) must leave alone */
static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i)
        h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

void twin_table_digest(void *h, uint64_t out[13])
{
    const mcx_compiled_table &t = ((const twin_table *)h)->t;
    out[0] = fnv(t.slot.data(), 4 * t.slot.size());
    out[1] = fnv(t.ext.data(), 4 * t.ext.size());
    out[2] = fnv(t.sig.data(), 4 * t.sig.size());
    out[3] = fnv(t.slots.data(), sizeof(mcx_slot_t) * t.slots.size());
    out[4] = fnv(t.tags.data(), t.tags.size());
    out[5] = fnv(t.sub.data(), sizeof(mcx_sub_t) * t.sub.size());
    out[6] = fnv(t.sub_tags.data(), t.sub_tags.size());
    out[7] = fnv(t.lens.data(), 4 * t.lens.size());
    out[8] = fnv(t.sets.data(), 4 * t.sets.size());
    out[9] = fnv(t.dup_start.data(), 4 * t.dup_start.size());
    out[10] = fnv(t.dup_idx.data(), 4 * t.dup_idx.size());
    out[11] = fnv(t.hw_tag.data(), t.hw_tag.size());
    const uint32_t n[4] = {t.n_distinct, t.n_sub_groups, t.max_group, t.P};
    out[12] = fnv(n, sizeof n);
}

/* strings i0..i1 against a kept table; candidates come from the literal-prefix
 * index, probed the way the device probes it, so the index is covered too */
int twin_table_match(void *h, const char *S, const uint32_t *s_off, uint32_t i0, uint32_t i1,
                     uint32_t *first_idx, uint32_t *count)
{
    const twin_table *tt = (const twin_table *)h;
    const mcx_compiled_table &t = tt->t;
    const uint32_t *sets = t.sets.empty() ? nullptr : t.sets.data();
    const size_t mask = t.slots.size() - 1;
    for (uint32_t i = i0; i < i1; ++i) {
        std::string str(S + s_off[i], S + s_off[i + 1]);
        bool bad = false;
        if (tt->normalize_dash)
            normalize(str, &bad);
        uint32_t first = MCX_NO_MATCH, cnt = 0;
        if (!bad) {
            uint32_t len = (uint32_t)str.size();
            HostStr s{(const uint8_t *)str.data(), len};
            /* the string's gram set, as the device builds it */
            uint32_t bloom[MCX_BLOOM_WORDS] = {0};
            bloom[MCX_BLOOM_FREE >> 5] |= 1u << (MCX_BLOOM_FREE & 31u);
            for (uint32_t q = 0; q + 4 <= len; ++q) {
                uint32_t bit = mcx_gram_bit(s.word(q));
                bloom[bit >> 5] |= 1u << (bit & 31u);
            }
            for (uint32_t L : t.lens) {
                if (L > len)
                    continue;
                uint32_t kw[MCX_KEY_MAX / 4] = {0};
                for (uint32_t k = 0; k < L; ++k)
                    kw[k >> 2] |= s.byte(k) << (8u * (k & 3u));
                const uint32_t a = mcx_key_hash(kw, L);
                for (size_t at = a & mask; t.slots[at].count != 0; at = (at + 1) & mask) {
                    const mcx_slot_t &sl = t.slots[at];
                    if (sl.hash != a || sl.len != L || memcmp(sl.key, kw, sizeof kw) != 0)
                        continue;
                    for (uint32_t k = sl.start; k < sl.start + sl.count; ++k) {
                        if (!mcx_sig_pass(t.sig[k], bloom))
                            continue;
                        const uint32_t *slot = &t.slot[(size_t)k * MCX_SLOT_WORDS];
                        /* the key's bytes were compared by the probe; the body
                         * does not hold them */
                        if (match_like_device(slot, t.ext.data(), s, len, sets)) {
                            if (slot[1] < first)
                                first = slot[1];
                            cnt += slot[MCX_SLOT_MULT];
                        }
                    }
                    break;
                }
            }
        }
        first_idx[i] = first;
        count[i] = cnt;
    }
    return 0;
}

/* Same contract as orc_alias_match; returns 0 or a negative MCX_E_* code. */
int twin_alias_match(const char *arena, const uint32_t *a_off, uint32_t P, const char *S,
                     const uint32_t *s_off, uint32_t i0, uint32_t i1, int normalize_dash,
                     uint32_t *first_idx, uint32_t *count)
{
    twin_table tt;
    std::string err;
    tt.normalize_dash = normalize_dash != 0;
    int rc = mcx_compile_table(arena, a_off, P, tt.normalize_dash, tt.t, err);
    if (rc != 0)
        return rc;
    return twin_table_match(&tt, S, s_off, i0, i1, first_idx, count);
}

} /* extern "C" */
