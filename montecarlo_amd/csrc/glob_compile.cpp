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

#include <array>
#include <map>
#include <string.h>

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

enum { M_LIT, M_ANY, M_SET };
struct Matcher {
    uint8_t kind;
    uint8_t byte;
    uint16_t set;
};

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

static void put16(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x & 0xFF));
    v.push_back((uint8_t)((x >> 8) & 0xFF));
}

static bool is_space(unsigned c)
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c;
}

/* reference desktop/ui.py:1540-1549: alias_line.strip().startswith(prefix) */
static uint8_t hw_tag_of(const uint8_t *pat, size_t n)
{
    static const char *tags[] = {"pci:", "usb:", "platform:", "hid:", "serio:", "of:"};
    size_t b = 0;
    while (b < n && is_space(pat[b]))
        ++b;
    for (const char *t : tags) {
        size_t tl = strlen(t);
        if (n - b >= tl && memcmp(pat + b, t, tl) == 0)
            return 1;
    }
    return 0;
}

} // namespace

int mcx_compile_pattern(const uint8_t *pat_in, size_t n, bool normalize_dash,
                        mcx_compiled_table &out, std::string &err)
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

    /* emit */
    if (out.prog_off.empty())
        out.prog_off.push_back(0);
    put16(out.prog, minlen);
    out.prog.push_back((uint8_t)flags);
    out.prog.push_back((uint8_t)segs.size());
    for (auto &sg : segs) {
        std::vector<uint8_t> code;
        for (const Matcher &m : sg) {
            if (m.kind == M_ANY) {
                code.push_back(MCX_OP_ANY);
            } else if (m.kind == M_SET) {
                code.push_back(MCX_OP_SET);
                put16(code, m.set);
            } else if (m.byte <= 3) {
                code.push_back(MCX_OP_ESC);
                code.push_back(m.byte);
            } else {
                code.push_back(m.byte);
            }
        }
        put16(out.prog, (uint32_t)sg.size());
        put16(out.prog, (uint32_t)code.size());
        out.prog.insert(out.prog.end(), code.begin(), code.end());
    }
    while (out.prog.size() & 3u)
        out.prog.push_back(0);
    out.prog_off.push_back((uint32_t)out.prog.size());

    /* prefilter record: leading literal bytes of the anchored segment */
    mcx_prefix_t pf;
    memset(&pf, 0, sizeof pf);
    uint8_t *pb = (uint8_t *)pf.pfx, *mb = (uint8_t *)pf.mask;
    if (never) {
        mb[0] = 0xFF; /* first byte must be NUL: only the empty string gets
                         through, and the program itself refuses it */
    } else {
        const std::vector<Matcher> &s0 = segs[0];
        for (size_t k = 0; k < s0.size() && k < MCX_PREFIX_BYTES; ++k) {
            if (s0[k].kind != M_LIT)
                break;
            pb[k] = s0[k].byte;
            mb[k] = 0xFF;
        }
    }
    out.prefix.push_back(pf);
    out.meta.push_back(MCX_META(minlen, flags));
    out.hw_tag.push_back(hw_tag_of(pat_in, n));
    return 0;
}

int mcx_compile_table(const char *arena, const uint32_t *a_off, uint32_t P,
                      bool normalize_dash, mcx_compiled_table &out, std::string &err)
{
    out = mcx_compiled_table();
    out.prog_off.reserve((size_t)P + 1);
    out.prefix.reserve(P);
    out.meta.reserve(P);
    out.hw_tag.reserve(P);
    out.prog_off.push_back(0);
    for (uint32_t j = 0; j < P; ++j) {
        if (a_off[j + 1] < a_off[j]) {
            err = "a_off not monotonic at pattern " + std::to_string(j);
            return MCX_E_BAD_ARG;
        }
        std::string e;
        int rc = mcx_compile_pattern((const uint8_t *)arena + a_off[j],
                                     (size_t)(a_off[j + 1] - a_off[j]), normalize_dash, out, e);
        if (rc != 0) {
            err = "pattern " + std::to_string(j) + ": " + e;
            return rc;
        }
    }
    return 0;
}
