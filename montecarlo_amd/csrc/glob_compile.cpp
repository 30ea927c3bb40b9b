/*
 * glob_compile.cpp - fnmatch(3) pattern text -> segment program.
 *
 * Grammar and corner cases follow glibc 2.35 fnmatch with flags = 0 in the C
 * locale, the function kmod calls on behalf of the reference
 * (reference montecarlo/libmontecarlo.c:123 -> modprobe -> fnmatch):
 *   - '\' quotes the next byte; a trailing '\' can never match;
 *   - '[' opens a bracket expression: optional '!' or '^', then elements; a
 *     ']' in first position is a literal; 'a-b' is a range unless the '-' is
 *     last; '\' quotes inside brackets too; a range whose end is the end of
 *     the pattern can never match; a bracket that is never closed makes the
 *     '[' an ordinary character and the rest is parsed as pattern text.
 * What a bracket expression does depends only on the one string byte it is
 * applied to, so it is evaluated here once for every byte value and stored as
 * a 256-bit set.  The evaluation walks the expression the way glibc does for
 * that byte (stop at the first element that matches, then skip to the closing
 * bracket), because for ill-formed expressions the outcome depends on where
 * the walk stops.
 */
#include "glob_compile.h"
#include "tables.h"

#include <algorithm>
#include <array>
#include <map>
#include <string.h>
#include <unordered_map>

#include "../../include/mcx.h"

namespace {

enum { BR_FAIL = 0, BR_MATCH = 1, BR_LITERAL = 2, BR_UNSUP = 3 };

struct Pat {
    const uint8_t *s;
    size_t n;
    unsigned at(size_t i) const { return i < n ? s[i] : 0u; }
};

static bool posix_subexpr(const Pat &P, unsigned c, size_t p)
{
    if (c != '[')
        return false;
    unsigned d = P.at(p);
    return d == ':' || d == '=' || d == '.';
}

/* Outcome of the bracket expression that starts right after a '[' at index p
 * when applied to string byte fn.  BR_MATCH sets *pnext to the index after
 * the closing bracket. */
static int bracket_eval(const Pat &P, size_t p, unsigned fn, size_t *pnext)
{
    bool neg = (P.at(p) == '!' || P.at(p) == '^');
    if (neg)
        ++p;
    unsigned c = P.at(p);
    ++p;
    for (;;) {
        bool element = false;
        if (c == '\\') {
            if (P.at(p) == 0)
                return BR_FAIL;
            c = P.at(p);
            ++p;
            element = true;
        } else if (posix_subexpr(P, c, p)) {
            return BR_UNSUP;
        } else if (c == 0) {
            return BR_LITERAL;
        } else {
            element = true;
        }
        if (element) {
            bool is_range = (P.at(p) == '-' && P.at(p + 1) != 0 && P.at(p + 1) != ']');
            if (!is_range && c == fn)
                goto matched;
            unsigned cold = c;
            c = P.at(p);
            ++p;
            if (c == '-' && P.at(p) != ']') {
                unsigned cend = P.at(p);
                ++p;
                if (cend == '[' && P.at(p) == '.')
                    return BR_UNSUP;
                if (cend == '\\') {
                    cend = P.at(p);
                    ++p;
                }
                if (cend == 0)
                    return BR_FAIL;
                if (cold <= fn && fn <= cend)
                    goto matched;
                c = P.at(p);
                ++p;
            }
        }
        if (c == ']')
            break;
    }
    if (!neg)
        return BR_FAIL;
    *pnext = p;
    return BR_MATCH;

matched:
    for (;;) {
        c = P.at(p);
        ++p;
        if (c == ']')
            break;
        if (c == 0)
            return BR_LITERAL;
        if (c == '\\') {
            if (P.at(p) == 0)
                return BR_FAIL;
            ++p;
        } else if (posix_subexpr(P, c, p)) {
            return BR_UNSUP;
        }
    }
    if (neg)
        return BR_FAIL;
    *pnext = p;
    return BR_MATCH;
}

enum { M_LIT = MCX_M_LIT, M_ANY = MCX_M_ANY, M_SET = MCX_M_SET };
typedef mcx_matcher Matcher;

static int intern_set(mcx_compiled_table &t, const std::array<uint32_t, 8> &bits, uint32_t *id)
{
    auto it = t.set_ids.find(bits);
    if (it != t.set_ids.end()) {
        *id = it->second;
        return 0;
    }
    uint32_t nid = (uint32_t)(t.sets.size() / 8);
    if (nid > 0xFFFFu)
        return -1;
    t.sets.insert(t.sets.end(), bits.begin(), bits.end());
    t.set_ids.emplace(bits, nid);
    *id = nid;
    return 0;
}

static bool is_space(unsigned c)
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c;
}

/* reference desktop/ui.py:1540-1549: alias_line.strip().startswith(prefix) */
static uint8_t hw_tag_of(const uint8_t *pat, size_t n)
{
    const mcx_tables_view *tv = mcx_tables_active(); /* the six bus tags unless a table file replaced them */
    size_t b = 0;
    while (b < n && is_space(pat[b]))
        ++b;
    for (uint32_t k = 0; k < tv->count[MCX_TS_HW_PREFIX]; ++k) {
        const char *t = tv->item[MCX_TS_HW_PREFIX][k];
        size_t tl = strlen(t);
        if (n - b >= tl && memcmp(pat + b, t, tl) == 0)
            return 1;
    }
    return 0;
}

static bool nseg_too_long(const mcx_pattern_ir &ir)
{
    for (const auto &sg : ir.segs)
        if (sg.size() > 0xFFFFu)
            return true;
    return false;
}

} // namespace

int mcx_parse_pattern(const uint8_t *pat_in, size_t n, bool normalize_dash, mcx_compiled_table &out,
                      mcx_pattern_ir &ir, std::string &err)
{
    if (n > MCX_MAX_PATTERN_BYTES) {
        err = "pattern longer than 4096 bytes";
        return MCX_E_PATTERN;
    }
    std::vector<uint8_t> norm;
    const uint8_t *pat = pat_in;
    for (size_t i = 0; i < n; ++i)
        if (pat_in[i] == 0) {
            err = "NUL byte inside pattern";
            return MCX_E_PATTERN;
        }
    if (normalize_dash) {
        /* kmod: '-' becomes '_' outside [...]; a bracket must be closed */
        norm.assign(pat_in, pat_in + n);
        for (size_t i = 0; i < n; ++i) {
            if (norm[i] == '-') {
                norm[i] = '_';
            } else if (norm[i] == ']') {
                err = "unmatched ']' under MCX_TABLE_NORMALIZE_DASH";
                return MCX_E_PATTERN;
            } else if (norm[i] == '[') {
                while (i < n && norm[i] != ']')
                    ++i;
                if (i >= n) {
                    err = "unmatched '[' under MCX_TABLE_NORMALIZE_DASH";
                    return MCX_E_PATTERN;
                }
            }
        }
        pat = norm.data();
    }

    Pat P{pat, n};
    std::vector<std::vector<Matcher>> segs(1);
    bool has_star = false, pending_seg = false, never = false;
    size_t i = 0;
    auto push = [&](Matcher m) {
        if (pending_seg) {
            segs.emplace_back();
            pending_seg = false;
        }
        segs.back().push_back(m);
    };
    while (i < n && !never) {
        unsigned c = pat[i++];
        switch (c) {
        case '?':
            push(Matcher{M_ANY, 0, 0});
            break;
        case '*':
            has_star = true;
            pending_seg = true;
            break;
        case '\\':
            if (i >= n) {
                never = true; /* trailing backslash loses */
                break;
            }
            push(Matcher{M_LIT, pat[i++], 0});
            break;
        case '[': {
            std::array<uint32_t, 8> bits{};
            size_t pnext = 0;
            bool have_next = false, literal = false, literal_ok = false;
            for (unsigned fn = 1; fn < 256; ++fn) {
                size_t pn = 0;
                int r = bracket_eval(P, i, fn, &pn);
                if (r == BR_UNSUP) {
                    err = "POSIX [:class:], [=x=] or [.x.] inside a bracket expression";
                    return MCX_E_PATTERN;
                }
                if (r == BR_MATCH) {
                    if (have_next && pn != pnext) {
                        err = "bracket expression whose end depends on the string byte";
                        return MCX_E_PATTERN;
                    }
                    have_next = true;
                    pnext = pn;
                    bits[fn >> 5] |= 1u << (fn & 31);
                } else if (r == BR_LITERAL) {
                    literal = true;
                    if (fn == '[')
                        literal_ok = true;
                }
            }
            if (literal) {
                if (have_next) {
                    err = "bracket expression that is closed for some bytes only";
                    return MCX_E_PATTERN;
                }
                /* never closed: '[' is an ordinary byte, unless even that is
                 * refused, in which case nothing can pass this point */
                if (literal_ok)
                    push(Matcher{M_LIT, (uint8_t)'[', 0});
                else
                    never = true;
                /* continue with the text right after the '[' */
            } else if (!have_next) {
                /* closed, but no byte is accepted */
                never = true;
            } else {
                uint32_t cnt = 0, last = 0;
                for (unsigned b = 1; b < 256; ++b)
                    if (bits[b >> 5] >> (b & 31) & 1u) {
                        ++cnt;
                        last = b;
                    }
                if (cnt == 255) {
                    push(Matcher{M_ANY, 0, 0});
                } else if (cnt == 1) {
                    push(Matcher{M_LIT, (uint8_t)last, 0});
                } else {
                    uint32_t id = 0;
                    if (intern_set(out, bits, &id) != 0) {
                        err = "more than 65536 distinct bracket sets";
                        return MCX_E_PATTERN;
                    }
                    push(Matcher{M_SET, 0, (uint16_t)id});
                }
                i = pnext;
            }
            break;
        }
        default:
            push(Matcher{M_LIT, (uint8_t)c, 0});
            break;
        }
    }

    uint32_t flags = 0;
    if (never) {
        flags = MCX_PF_NEVER;
        segs.assign(1, {});
    } else {
        if (has_star)
            flags |= MCX_PF_HAS_STAR;
        if (pending_seg)
            flags |= MCX_PF_TRAIL_STAR;
    }
    if (segs.size() > MCX_MAX_SEGMENTS) {
        err = "more than 255 '*'-separated segments";
        return MCX_E_PATTERN;
    }
    uint32_t minlen = 0;
    for (auto &sg : segs)
        minlen += (uint32_t)sg.size();
    ir.segs.swap(segs);
    ir.flags = flags;
    ir.minlen = minlen;
    return 0;
}

namespace {
/* the MCX_SET_WORDS words of one set entry */
static void emit_set_entry(uint32_t pos, uint32_t id, const uint32_t *bits, std::vector<uint32_t> &prog)
{
    std::vector<uint8_t> in, notin;
    for (unsigned b = 1; b < 256; ++b)
        ((bits[b >> 5] >> (b & 31)) & 1u ? in : notin).push_back((uint8_t)b);
    uint32_t w[MCX_SET_WORDS] = {pos | (id << 16), MCX_SET_GENERAL, 0u, 0u};
    const std::vector<uint8_t> *list = nullptr;
    if (in.size() <= MCX_SET_MEMBERS)
        list = &in;
    else if (notin.size() <= MCX_SET_MEMBERS)
        list = &notin;
    if (list) {
        w[1] = (uint32_t)list->size() | (list == &notin ? 0x100u : 0u);
        for (size_t i = 0; i < list->size(); ++i) {
            if (i < 2)
                w[1] |= (uint32_t)(*list)[i] << (16 + 8 * i);
            else if (i < 6)
                w[2] |= (uint32_t)(*list)[i] << (8 * (i - 2));
            else
                w[3] |= (uint32_t)(*list)[i] << (8 * (i - 6));
        }
    }
    prog.insert(prog.end(), w, w + MCX_SET_WORDS);
}
} // namespace

void mcx_emit_program(const mcx_pattern_ir &ir, uint32_t keylen, const std::vector<uint32_t> &sets,
                      std::vector<uint32_t> &prog, std::vector<uint32_t> *seg_words)
{
    prog.push_back(MCX_HDR(ir.minlen, ir.flags, keylen, ir.segs.size()));
    for (size_t si = 0; si < ir.segs.size(); ++si) {
        const auto &sg = ir.segs[si];
        if (seg_words)
            seg_words->push_back((uint32_t)prog.size());
        const uint32_t nchars = (uint32_t)sg.size(), nchunks = (nchars + 3u) / 4u;
        const uint32_t c0 = si == 0 ? keylen / 4u : 0u; /* chunks the index key covers */
        std::vector<uint32_t> setlist;
        std::vector<uint32_t> value(nchunks, 0u), mask(nchunks, 0u);
        bool masked = false;
        for (uint32_t k = 0; k < nchars; ++k) {
            const Matcher &m = sg[k];
            uint32_t sh = (k & 3u) * 8u;
            if (m.kind == M_LIT) {
                value[k / 4] |= (uint32_t)m.byte << sh;
                mask[k / 4] |= 0xFFu << sh;
            } else {
                masked = true;
                if (m.kind == M_SET)
                    setlist.push_back(k | ((uint32_t)m.set << 16));
            }
        }
        if (si > 0 && setlist.empty() && nchars >= 1 && nchars <= 3) {
            /* header and chunk in one word; three bits say which bytes are literals */
            uint32_t lit = 0;
            for (uint32_t k = 0; k < nchars; ++k)
                lit |= (sg[k].kind == M_LIT ? 1u : 0u) << k;
            prog.push_back((value[0] & 0xFFFFFFu) | (nchars << 24) | (lit << 26) | MCX_SEG_INLINE);
            continue;
        }
        prog.push_back(nchars | ((uint32_t)setlist.size() << 16) | (masked ? MCX_SEG_MASKED : 0u));
        for (uint32_t c = c0; c < nchunks; ++c)
            prog.push_back(value[c]);
        if (masked) {
            /* four bits per chunk: which of its bytes are literals */
            std::vector<uint32_t> mw((nchunks - c0 + 7u) / 8u, 0u);
            for (uint32_t c = c0; c < nchunks; ++c) {
                uint32_t nib = 0;
                for (uint32_t b = 0; b < 4; ++b)
                    nib |= ((mask[c] >> (8u * b)) & 1u) << b;
                mw[(c - c0) / 8u] |= nib << (4u * ((c - c0) % 8u));
            }
            prog.insert(prog.end(), mw.begin(), mw.end());
        }
        for (uint32_t e : setlist)
            emit_set_entry(e & 0xFFFFu, e >> 16, &sets[(size_t)(e >> 16) * 8], prog);
    }
}

void mcx_fill_slot(const mcx_pattern_ir &ir, uint32_t keylen, uint32_t orig, uint32_t mult,
                   const std::vector<uint32_t> &sets, uint32_t *slot, std::vector<uint32_t> &ext)
{
    std::vector<uint32_t> one, seg_words;
    mcx_emit_program(ir, keylen, sets, one, &seg_words);
    std::fill(slot, slot + MCX_SLOT_WORDS, 0u);
    slot[0] = one[0];
    slot[1] = orig;
    slot[MCX_SLOT_MULT] = mult;
    const size_t body = one.size() - 1;
    if (ir.segs.size() <= MCX_DIR_SEGS && body <= MCX_SLOT_WORDS - MCX_SLOT_BODY) {
        for (size_t k = 0; k < seg_words.size(); ++k) /* word of `one` -> word of the slot */
            slot[MCX_SLOT_DIR + k / 4] |= (seg_words[k] - 1u + MCX_SLOT_BODY) << (8u * (k % 4));
        std::copy(one.begin() + 1, one.end(), slot + MCX_SLOT_BODY);
    } else {
        slot[0] = MCX_HDR(ir.minlen, ir.flags | MCX_PF_EXT, keylen, ir.segs.size());
        slot[MCX_SLOT_DIR] = (uint32_t)ext.size();
        ext.insert(ext.end(), one.begin() + 1, one.end());
    }
}

namespace {
/* literal prefix of the anchored segment, at most MCX_KEY_MAX bytes */
static std::string key_of(const mcx_pattern_ir &ir)
{
    std::string k;
    for (const Matcher &m : ir.segs[0]) {
        if (m.kind != M_LIT || k.size() >= MCX_KEY_MAX)
            break;
        k.push_back((char)m.byte);
    }
    return k;
}

/* up to four Bloom bits of the pattern's grams, preferring grams that are not
 * wholly inside the index key (every string that reaches the pattern through
 * the index already starts with the key) */
static uint32_t sig_of(const mcx_pattern_ir &ir, size_t keylen)
{
    std::vector<uint32_t> bits;
    for (size_t si = 0; si < ir.segs.size(); ++si) {
        const auto &sg = ir.segs[si];
        for (size_t k = 0; k + 4 <= sg.size(); ++k) {
            if (sg[k].kind != M_LIT || sg[k + 1].kind != M_LIT || sg[k + 2].kind != M_LIT ||
                sg[k + 3].kind != M_LIT)
                continue;
            if (si == 0 && k + 4 <= keylen)
                continue;
            uint32_t g = (uint32_t)sg[k].byte | ((uint32_t)sg[k + 1].byte << 8) |
                         ((uint32_t)sg[k + 2].byte << 16) | ((uint32_t)sg[k + 3].byte << 24);
            uint32_t b = mcx_gram_bit(g);
            if (std::find(bits.begin(), bits.end(), b) == bits.end())
                bits.push_back(b);
        }
    }
    uint32_t sig = 0;
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t b = MCX_BLOOM_FREE;
        if (!bits.empty()) {
            /* spread the picks over the list */
            size_t n = bits.size();
            b = n <= 4 ? (k < n ? bits[k] : bits[0]) : bits[(k * (n - 1)) / 3];
        }
        sig |= b << (8u * k);
    }
    return sig;
}

/* the distinct four-literal windows of the pattern that are not wholly inside
 * the index key: every string the pattern matches holds each of them */
/* signature from up to four grams picked by the caller */
static uint32_t sig_from(const std::vector<uint32_t> &grams)
{
    uint32_t bits[4], nb = 0;
    for (uint32_t g : grams) {
        const uint32_t b = mcx_gram_bit(g);
        bool have = false;
        for (uint32_t k = 0; k < nb; ++k)
            have = have || bits[k] == b;
        if (!have && nb < 4)
            bits[nb++] = b;
    }
    uint32_t sig = 0;
    for (uint32_t k = 0; k < 4; ++k)
        sig |= (k < nb ? bits[k] : (nb ? bits[0] : MCX_BLOOM_FREE)) << (8u * k);
    return sig;
}

/*
 * The patterns that are screened side by side - a group, or inside a large group the run of
 * one anchor gram - mostly differ in a few bytes ("bc0Csc03i30" / "bc0Csc03i00"), and a string
 * that reaches them holds what they share.  Only the grams that tell them apart reject
 * anything, so each pattern's signature takes the grams fewest of its neighbours hold
 * (ties: the rarer one in the whole group, then the later one in the pattern), spread over
 * the pattern.
 * cand[i]: the grams of neighbour i; sig_choice[ids[i]] gets the signature.
 */
static void choose_sigs(const std::vector<std::vector<uint32_t>> &cand, const uint32_t *ids, size_t first, size_t n,
                        const std::unordered_map<uint32_t, uint32_t> *group_freq, std::vector<uint32_t> &sig_choice,
                        std::vector<uint8_t> &sig_chosen)
{
    if (n < 2)
        return;
    std::unordered_map<uint32_t, uint32_t> here;
    for (size_t i = first; i < first + n; ++i)
        for (uint32_t g : cand[i])
            here[g]++;
    struct Pick {
        uint32_t near, wide, at, gram;
    };
    std::vector<Pick> picks;
    std::vector<uint32_t> chosen;
    for (size_t i = first; i < first + n; ++i) {
        if (cand[i].size() <= 4)
            continue; /* all of them fit the signature anyway */
        picks.clear();
        for (size_t a = 0; a < cand[i].size(); ++a) {
            const uint32_t g = cand[i][a];
            uint32_t wide = 0;
            if (group_freq) {
                auto it = group_freq->find(g);
                wide = it == group_freq->end() ? 0u : it->second;
            }
            picks.push_back(Pick{here[g], wide, (uint32_t)a, g});
        }
        std::sort(picks.begin(), picks.end(), [](const Pick &x, const Pick &y) {
            if (x.near != y.near)
                return x.near < y.near;
            if (x.wide != y.wide)
                return x.wide < y.wide;
            return x.at > y.at;
        });
        /* rarest first, but spread over the pattern: grams three or more places apart share at
         * most one byte, so four of them watch (nearly) sixteen bytes instead of seven */
        chosen.clear();
        uint32_t taken[4];
        for (int pass = 0; pass < 2 && chosen.size() < 4; ++pass)
            for (size_t a = 0; a < picks.size() && chosen.size() < 4; ++a) {
                bool ok = true;
                for (size_t t = 0; t < chosen.size(); ++t) {
                    const uint32_t gap = picks[a].at > taken[t] ? picks[a].at - taken[t] : taken[t] - picks[a].at;
                    ok = ok && (pass == 0 ? gap >= 3u : gap != 0u);
                }
                if (ok) {
                    taken[chosen.size()] = picks[a].at;
                    chosen.push_back(picks[a].gram);
                }
            }
        sig_choice[ids[i]] = sig_from(chosen);
        sig_chosen[ids[i]] = 1;
    }
}

static std::vector<uint32_t> grams_of(const mcx_pattern_ir &ir, size_t keylen)
{
    std::vector<uint32_t> out;
    for (size_t si = 0; si < ir.segs.size(); ++si) {
        const auto &sg = ir.segs[si];
        for (size_t k = 0; k + 4 <= sg.size(); ++k) {
            if (sg[k].kind != M_LIT || sg[k + 1].kind != M_LIT || sg[k + 2].kind != M_LIT ||
                sg[k + 3].kind != M_LIT)
                continue;
            if (si == 0 && k + 4 <= keylen)
                continue;
            uint32_t g = (uint32_t)sg[k].byte | ((uint32_t)sg[k + 1].byte << 8) |
                         ((uint32_t)sg[k + 2].byte << 16) | ((uint32_t)sg[k + 3].byte << 24);
            if (std::find(out.begin(), out.end(), g) == out.end())
                out.push_back(g);
        }
    }
    return out;
}
} // namespace

int mcx_compile_table(const char *arena, const uint32_t *a_off, uint32_t P,
                      bool normalize_dash, mcx_compiled_table &out, std::string &err)
{
    out = mcx_compiled_table();
    out.P = P;
    out.hw_tag.resize(P);
    std::vector<mcx_pattern_ir> irs(P);
    std::vector<std::string> keys(P);
    std::vector<uint32_t> order, mult(P, 0u);
    order.reserve(P);
    /* patterns with the same text are indexed once, under their first position */
    std::unordered_map<std::string, uint32_t> seen;
    seen.reserve(P);
    std::unordered_map<uint32_t, std::vector<uint32_t>> repeats; /* first position -> later positions */
    for (uint32_t j = 0; j < P; ++j) {
        if (a_off[j + 1] < a_off[j]) {
            err = "a_off not monotonic at pattern " + std::to_string(j);
            return MCX_E_BAD_ARG;
        }
        const uint8_t *pat = (const uint8_t *)arena + a_off[j];
        size_t n = (size_t)(a_off[j + 1] - a_off[j]);
        out.hw_tag[j] = hw_tag_of(pat, n);
        auto ins = seen.emplace(std::string((const char *)pat, n), j);
        if (!ins.second) {
            /* a repeat: parsed (and, if need be, refused) when first seen */
            if (mult[ins.first->second]) {
                mult[ins.first->second]++;
                repeats[ins.first->second].push_back(j);
            }
            continue;
        }
        std::string e;
        int rc = mcx_parse_pattern(pat, n, normalize_dash, out, irs[j], e);
        if (rc != 0) {
            err = "pattern " + std::to_string(j) + ": " + e;
            return rc;
        }
        if (nseg_too_long(irs[j])) {
            err = "pattern " + std::to_string(j) + ": segment longer than 65535 bytes";
            return MCX_E_PATTERN;
        }
        if (irs[j].flags & MCX_PF_NEVER)
            continue; /* matches nothing: never a candidate */
        keys[j] = key_of(irs[j]);
        mult[j] = 1;
        order.push_back(j);
    }
    /* index order: by key length, then key bytes, then table position */
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (keys[x].size() != keys[y].size())
            return keys[x].size() < keys[y].size();
        int c = keys[x].compare(keys[y]);
        if (c != 0)
            return c < 0;
        return x < y;
    });
    size_t n_idx = order.size();
    out.n_distinct = (uint32_t)n_idx;
    /* large groups: every pattern picks the rarest of its grams within the group as
     * its anchor; the group is reordered (no anchor first, then by anchor gram, then
     * by table position) and the runs are remembered for the second index */
    struct SubRun {
        uint32_t gram, gid, start, count;
    };
    struct SubGroup {
        uint32_t gid, rest;
    };
    /* test seam: MCX_SUB_MIN in the environment moves the size from which a group gets a
     * second index (tests set it low so that small tables exercise that path) */
    const char *sub_env = getenv("MCX_SUB_MIN");
    const size_t sub_min = sub_env && *sub_env ? (size_t)strtoul(sub_env, nullptr, 10) : (size_t)MCX_SUB_MIN;
    std::vector<SubRun> sub_runs;
    std::unordered_map<uint32_t, SubGroup> sub_groups; /* by group start */
    std::vector<uint32_t> sig_choice(P, 0u); /* by pattern: the signature picked among its neighbours */
    std::vector<uint8_t> sig_chosen(P, 0u);
    for (size_t g0 = 0; g0 < n_idx;) {
        size_t g1 = g0 + 1;
        while (g1 < n_idx && keys[order[g1]] == keys[order[g0]])
            ++g1;
        if (g1 - g0 > out.max_group)
            out.max_group = (uint32_t)(g1 - g0);
        if (g1 - g0 > sub_min) {
            const size_t keylen = keys[order[g0]].size();
            std::vector<std::vector<uint32_t>> cand(g1 - g0);
            std::unordered_map<uint32_t, uint32_t> freq;
            for (size_t k = g0; k < g1; ++k) {
                cand[k - g0] = grams_of(irs[order[k]], keylen);
                for (uint32_t g : cand[k - g0])
                    freq[g]++;
            }
            struct Item {
                uint32_t j, has, gram, ci;
            };
            std::vector<Item> items(g1 - g0);
            for (size_t k = g0; k < g1; ++k) {
                Item it{order[k], 0u, 0u, (uint32_t)(k - g0)};
                uint32_t best = 0xFFFFFFFFu;
                for (uint32_t g : cand[k - g0]) {
                    const uint32_t f = freq[g];
                    if (f < best || (f == best && g < it.gram)) {
                        best = f;
                        it.gram = g;
                        it.has = 1u;
                    }
                }
                items[k - g0] = it;
            }
            std::sort(items.begin(), items.end(), [](const Item &x, const Item &y) {
                if (x.has != y.has)
                    return x.has < y.has;
                if (x.gram != y.gram)
                    return x.gram < y.gram;
                return x.j < y.j;
            });
            const uint32_t gid = out.n_sub_groups++;
            uint32_t rest = 0;
            for (size_t k = g0; k < g1; ++k) {
                const Item &it = items[k - g0];
                order[k] = it.j;
                if (!it.has) {
                    ++rest;
                    continue;
                }
                if (sub_runs.empty() || sub_runs.back().gid != gid || sub_runs.back().gram != it.gram)
                    sub_runs.push_back(SubRun{it.gram, gid, (uint32_t)k, 0u});
                sub_runs.back().count++;
            }
            sub_groups[(uint32_t)g0] = SubGroup{gid, rest};
            /* signatures: what tells a pattern from the others of its run */
            std::vector<std::vector<uint32_t>> by_place(g1 - g0);
            for (size_t k = g0; k < g1; ++k)
                by_place[k - g0] = std::move(cand[items[k - g0].ci]);
            for (size_t a = rest; a < g1 - g0;) {
                size_t b = a + 1;
                while (b < g1 - g0 && items[b].gram == items[a].gram)
                    ++b;
                choose_sigs(by_place, &order[g0], a, b - a, &freq, sig_choice, sig_chosen);
                a = b;
            }
        } else if (g1 - g0 > 1) {
            const size_t keylen = keys[order[g0]].size();
            std::vector<std::vector<uint32_t>> cand(g1 - g0);
            for (size_t k = g0; k < g1; ++k)
                cand[k - g0] = grams_of(irs[order[k]], keylen);
            choose_sigs(cand, &order[g0], 0, g1 - g0, nullptr, sig_choice, sig_chosen);
        }
        g0 = g1;
    }
    out.slot.assign(n_idx * (size_t)MCX_SLOT_WORDS, 0u);
    out.sig.resize(n_idx);
    out.dup_start.assign(n_idx ? n_idx : 1, 0u);
    struct Group {
        uint32_t start, count;
    };
    std::vector<Group> groups;
    for (size_t k = 0; k < n_idx; ++k) {
        uint32_t j = order[k];
        if (k == 0 || keys[j] != keys[order[k - 1]]) {
            groups.push_back(Group{(uint32_t)k, 0});
            if (out.lens.empty() || out.lens.back() != keys[j].size())
                out.lens.push_back((uint32_t)keys[j].size());
        }
        groups.back().count++;
        mcx_fill_slot(irs[j], (uint32_t)keys[j].size(), j, mult[j], out.sets,
                      &out.slot[k * (size_t)MCX_SLOT_WORDS], out.ext);
        out.sig[k] = sig_chosen[j] ? sig_choice[j] : sig_of(irs[j], keys[j].size());
        if (mult[j] > 1u) {
            out.dup_start[k] = (uint32_t)out.dup_idx.size();
            out.dup_idx.push_back(j);
            const std::vector<uint32_t> &later = repeats[j];
            out.dup_idx.insert(out.dup_idx.end(), later.begin(), later.end());
        }
    }
    if (out.dup_idx.empty())
        out.dup_idx.push_back(0u);
    /* the matchers read up to two words past a program, the screen up to three
     * signatures past a group: keep them in bounds */
    out.sig.insert(out.sig.end(), 4, 0u);
    out.slot.insert(out.slot.end(), 4, 0u);
    out.ext.insert(out.ext.end(), 4, 0u);
    /* hash table, never empty.  A probe that meets an occupied slot costs one
     * more dependent load, so the table is kept sparse (one slot in eight
     * used) as long as that is cheap, and half empty beyond */
    size_t cap = 16;
    const size_t spread = groups.size() <= ((size_t)1 << 17) ? 8 : 2;
    while (cap < spread * groups.size())
        cap *= 2;
    out.slots.assign(cap, mcx_slot_t{0, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}});
    for (const Group &g : groups) {
        const std::string &key = keys[order[g.start]];
        uint32_t kw[MCX_KEY_MAX / 4] = {0};
        memcpy(kw, key.data(), key.size());
        const uint32_t a = mcx_key_hash(kw, (uint32_t)key.size());
        size_t at = a & (cap - 1);
        while (out.slots[at].count != 0)
            at = (at + 1) & (cap - 1);
        mcx_slot_t sl = mcx_slot_t{a, (uint32_t)key.size(), g.start, g.count, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}};
        memcpy(sl.key, kw, sizeof kw);
        auto sg = sub_groups.find(g.start);
        if (sg != sub_groups.end()) {
            sl.pad[0] = sg->second.gid + 1u;
            sl.pad[1] = sg->second.rest;
        }
        out.slots[at] = sl;
    }
    /* the tag arrays repeat their first three bytes behind the last one: a reader takes the
     * tags of four consecutive entries as one word from any entry on */
    out.tags.resize(cap + 3);
    for (size_t at = 0; at < cap + 3; ++at) {
        const mcx_slot_t &sl = out.slots[at & (cap - 1)];
        out.tags[at] = sl.count ? (uint8_t)mcx_key_tag(sl.hash) : 0u;
    }
    /* second index, never empty, one entry in four used */
    size_t scap = 16;
    while (scap < 4 * sub_runs.size())
        scap *= 2;
    out.sub.assign(scap, mcx_sub_t{0, 0, 0, 0});
    out.sub_tags.assign(scap + 3, 0u);
    for (const SubRun &r : sub_runs) {
        const uint32_t h = mcx_sub_hash(r.gram, r.gid);
        size_t at = h & (scap - 1);
        while (out.sub[at].count != 0)
            at = (at + 1) & (scap - 1);
        out.sub[at] = mcx_sub_t{r.gram, r.gid, r.start, r.count};
        out.sub_tags[at] = (uint8_t)mcx_key_tag(h);
    }
    for (size_t at = 0; at < 3; ++at)
        out.sub_tags[scap + at] = out.sub_tags[at];
    return 0;
}
