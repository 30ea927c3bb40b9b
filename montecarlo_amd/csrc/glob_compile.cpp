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
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <thread>
#include <unordered_map>
#include <utility>

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
    for (size_t k = 0; k < ir.nsegs(); ++k)
        if (ir.seg(k).size() > 0xFFFFu)
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
    static thread_local std::vector<Matcher> flat; /* kept between calls */
    static thread_local std::vector<uint32_t> ends;
    flat.clear();
    ends.clear();
    bool has_star = false, pending_seg = false, never = false;
    size_t i = 0;
    auto push = [&](Matcher m) {
        if (pending_seg) {
            ends.push_back((uint32_t)flat.size()); /* the segment before ends here */
            pending_seg = false;
        }
        flat.push_back(m);
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
        flat.clear();
        ends.clear();
    } else {
        if (has_star)
            flags |= MCX_PF_HAS_STAR;
        if (pending_seg)
            flags |= MCX_PF_TRAIL_STAR;
    }
    ends.push_back((uint32_t)flat.size());
    if (ends.size() > MCX_MAX_SEGMENTS) {
        err = "more than 255 '*'-separated segments";
        return MCX_E_PATTERN;
    }
    const uint32_t minlen = (uint32_t)flat.size();
    ir.m.assign(flat.begin(), flat.end());
    ir.seg_end.assign(ends.begin(), ends.end());
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
    prog.push_back(MCX_HDR(ir.minlen, ir.flags, keylen, ir.nsegs()));
    static thread_local std::vector<uint32_t> setlist, value, mask, mw; /* kept between calls: no allocation per segment */
    for (size_t si = 0; si < ir.nsegs(); ++si) {
        const mcx_seg sg = ir.seg(si);
        if (seg_words)
            seg_words->push_back((uint32_t)prog.size());
        const uint32_t nchars = (uint32_t)sg.size(), nchunks = (nchars + 3u) / 4u;
        const uint32_t c0 = si == 0 ? keylen / 4u : 0u; /* chunks the index key covers */
        setlist.clear();
        value.assign(nchunks, 0u);
        mask.assign(nchunks, 0u);
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
            mw.assign((nchunks - c0 + 7u) / 8u, 0u);
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
    static thread_local std::vector<uint32_t> one, seg_words;
    one.clear();
    seg_words.clear();
    mcx_emit_program(ir, keylen, sets, one, &seg_words);
    std::fill(slot, slot + MCX_SLOT_WORDS, 0u);
    slot[0] = one[0];
    slot[1] = orig;
    slot[MCX_SLOT_MULT] = mult;
    const size_t body = one.size() - 1;
    if (ir.nsegs() <= MCX_DIR_SEGS && body <= MCX_SLOT_WORDS - MCX_SLOT_BODY) {
        for (size_t k = 0; k < seg_words.size(); ++k) /* word of `one` -> word of the slot */
            slot[MCX_SLOT_DIR + k / 4] |= (seg_words[k] - 1u + MCX_SLOT_BODY) << (8u * (k % 4));
        std::copy(one.begin() + 1, one.end(), slot + MCX_SLOT_BODY);
    } else {
        slot[0] = MCX_HDR(ir.minlen, ir.flags | MCX_PF_EXT, keylen, ir.nsegs());
        slot[MCX_SLOT_DIR] = (uint32_t)ext.size();
        ext.insert(ext.end(), one.begin() + 1, one.end());
    }
}

namespace {
/* up to four Bloom bits of the pattern's grams, preferring grams that are not
 * wholly inside the index key (every string that reaches the pattern through
 * the index already starts with the key) */
static uint32_t sig_of(const mcx_pattern_ir &ir, size_t keylen)
{
    std::vector<uint32_t> bits;
    for (size_t si = 0; si < ir.nsegs(); ++si) {
        const mcx_seg sg = ir.seg(si);
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
    for (size_t si = 0; si < ir.nsegs(); ++si) {
        const mcx_seg sg = ir.seg(si);
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

namespace {

/* run fn(t) for t = 0 .. n-1, each on a thread of its own (the caller's for t = 0) */
template <class F>
static void on_threads(unsigned n, F fn)
{
    std::vector<std::thread> th;
    th.reserve(n > 0 ? n - 1 : 0);
    for (unsigned t = 1; t < n; ++t)
        th.emplace_back([&fn, t] { fn(t); });
    fn(0u);
    for (auto &x : th)
        x.join();
}

/* n objects built and torn down range by range on the threads that use them (three million
 * parsed patterns cost a quarter of a second to construct and as much to free from one thread) */
template <class T>
struct ThreadedArray {
    T *p;
    size_t n;
    unsigned threads;
    ThreadedArray(size_t count, unsigned nthreads) : p((T *)::operator new(count * sizeof(T) + 1)), n(count), threads(nthreads)
    {
        on_threads(threads, [&](unsigned t) {
            for (size_t k = n * t / threads; k < n * (t + 1) / threads; ++k)
                new (&p[k]) T();
        });
    }
    ~ThreadedArray()
    {
        on_threads(threads, [&](unsigned t) {
            for (size_t k = n * t / threads; k < n * (t + 1) / threads; ++k)
                p[k].~T();
        });
        ::operator delete(p);
    }
    ThreadedArray(const ThreadedArray &) = delete;
    ThreadedArray &operator=(const ThreadedArray &) = delete;
    T &operator[](size_t k) { return p[k]; }
    const T &operator[](size_t k) const { return p[k]; }
};

/* hash of a pattern's text: eight bytes a step */
static uint64_t text_hash(const uint8_t *p, size_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull);
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 29;
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 32);
}

/* a pattern's index key: the literal prefix of its anchored segment, at most MCX_KEY_MAX bytes */
struct KeyOf {
    uint8_t len;
    uint8_t b[MCX_KEY_MAX];
};

static inline int key_cmp(const KeyOf &x, const KeyOf &y)
{
    if (x.len != y.len)
        return x.len < y.len ? -1 : 1;
    return memcmp(x.b, y.b, x.len);
}

struct CompileError {
    uint32_t j = 0xFFFFFFFFu;
    int rc = 0;
    std::string text;
    void take(uint32_t at, int code, const std::string &what)
    {
        if (at < j) {
            j = at;
            rc = code;
            text = what;
        }
    }
};

struct SubRun {
    uint32_t gram, gid, start, count;
};

/* anchors, order and signatures of one group of the index (positions g0 .. g1 of `order`);
 * touches nothing but the group's own part of order / sig_choice / sig_chosen and `runs` */
static void settle_group(const ThreadedArray<mcx_pattern_ir> &irs, size_t keylen, uint32_t *order, size_t g0, size_t g1,
                         bool large, uint32_t gid, std::vector<SubRun> *runs, uint32_t *rest_out,
                         std::vector<uint32_t> &sig_choice, std::vector<uint8_t> &sig_chosen)
{
    std::vector<std::vector<uint32_t>> cand(g1 - g0);
    if (!large) {
        for (size_t k = g0; k < g1; ++k)
            cand[k - g0] = grams_of(irs[order[k]], keylen);
        choose_sigs(cand, &order[g0], 0, g1 - g0, nullptr, sig_choice, sig_chosen);
        return;
    }
    /* every pattern picks the rarest of its grams within the group as its anchor; the group
     * is reordered (no anchor first, then by anchor gram, then by table position) and the
     * runs are remembered for the second index */
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
    uint32_t rest = 0;
    for (size_t k = g0; k < g1; ++k) {
        const Item &it = items[k - g0];
        order[k] = it.j;
        if (!it.has) {
            ++rest;
            continue;
        }
        if (runs->empty() || runs->back().gram != it.gram)
            runs->push_back(SubRun{it.gram, gid, (uint32_t)k, 0u});
        runs->back().count++;
    }
    *rest_out = rest;
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
}

} // namespace

/*
 * The compile is split over threads where its parts are independent - text hashes, the search
 * for repeated texts (by hash partition), parsing (by ranges of the table), the sort into index
 * order (ranges, then merges), anchors and signatures (by group), slots (by ranges of the
 * index) - and every part is put together in table order, so the arrays do not depend on the
 * number of threads.
 */
int mcx_compile_table(const char *arena, const uint32_t *a_off, uint32_t P, bool normalize_dash,
                      mcx_compiled_table &out, std::string &err, const mcx_compile_options *opt)
{
    out = mcx_compiled_table();
    out.P = P;
    out.hw_tag.resize(P);
    const size_t sub_min = opt && opt->sub_min ? (size_t)opt->sub_min : (size_t)MCX_SUB_MIN;
    unsigned T = opt ? opt->threads : 0u;
    if (T == 0u) {
        T = std::thread::hardware_concurrency();
        T = T == 0u ? 1u : (T > 16u ? 16u : T);
        if ((size_t)T * 4096u > (size_t)P) /* small tables: threads cost more than they give */
            T = (unsigned)((size_t)P / 4096u) ? (unsigned)((size_t)P / 4096u) : 1u;
    }
    if (T > 64u)
        T = 64u;

#ifdef MCX_COMPILE_TIMING /* diagnostic build: where the compile's time goes, on stderr */
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "compile %-10s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
#else
    auto lap = [](const char *) {};
#endif
    CompileError bad;
    /* patterns behind a broken offset are never looked at */
    uint32_t limit = P;
    for (uint32_t j = 0; j < P; ++j)
        if (a_off[j + 1] < a_off[j]) {
            limit = j;
            break;
        }
    auto range_of = [&](unsigned t, size_t n) { return std::pair<size_t, size_t>{n * t / T, n * (t + 1) / T}; };

    /* 1. text hashes and the bus-tag byte */
    std::vector<uint32_t> first_of(limit); /* first table position with j's text */
    std::vector<uint64_t> hash(limit);
    std::vector<uint8_t> owner(limit); /* the thread that looks after the text's copies */
    on_threads(T, [&](unsigned t) {
        auto r = range_of(t, limit);
        for (size_t j = r.first; j < r.second; ++j) {
            const uint8_t *pat = (const uint8_t *)arena + a_off[j];
            const size_t n = (size_t)(a_off[j + 1] - a_off[j]);
            out.hw_tag[j] = hw_tag_of(pat, n);
            hash[j] = text_hash(pat, n);
            owner[j] = (uint8_t)((hash[j] >> 40) % T);
            first_of[j] = (uint32_t)j;
        }
    });

    lap("hash");
    /* 2. patterns with the same text are indexed once, under their first position:
     * thread t owns the texts whose hash falls into its share and walks the table in order, so
     * "first" means first; only the copies' entries of first_of are written (neighbouring
     * entries belong to different threads) */
    on_threads(T, [&](unsigned t) {
        size_t mine = 0;
        for (size_t j = 0; j < limit; ++j)
            mine += owner[j] == t;
        size_t cap = 16;
        while (cap < 2 * mine)
            cap *= 2;
        std::vector<uint32_t> cell(cap, 0xFFFFFFFFu);
        for (size_t j = 0; j < limit; ++j) {
            if (owner[j] != t)
                continue;
            const size_t n = (size_t)(a_off[j + 1] - a_off[j]);
            size_t at = (size_t)hash[j] & (cap - 1);
            for (;; at = (at + 1) & (cap - 1)) {
                const uint32_t f = cell[at];
                if (f == 0xFFFFFFFFu) {
                    cell[at] = (uint32_t)j;
                    break;
                }
                if (hash[f] == hash[j] && (size_t)(a_off[f + 1] - a_off[f]) == n &&
                    memcmp(arena + a_off[f], arena + a_off[j], n) == 0) {
                    first_of[j] = f;
                    break;
                }
            }
        }
    });
    std::vector<uint64_t>().swap(hash);
    std::vector<uint8_t>().swap(owner);
    lap("dedup");

    /* 3. parse every distinct text; bracket sets are numbered per thread first */
    ThreadedArray<mcx_pattern_ir> irs(limit, T);
    ThreadedArray<KeyOf> keys(limit, T);
    ThreadedArray<uint8_t> live(limit, T); /* 1: parsed and able to match something */
    struct Local {
        mcx_compiled_table sets;          /* sets + set_ids of this thread's range */
        std::vector<uint32_t> set_first;  /* per local set: the pattern that brought it in */
        std::vector<uint32_t> with_sets;  /* patterns of the range that name a set */
        CompileError bad;
    };
    std::vector<Local> loc(T);
    on_threads(T, [&](unsigned t) {
        Local &L = loc[t];
        auto r = range_of(t, limit);
        for (size_t j = r.first; j < r.second; ++j) {
            if (first_of[j] != j)
                continue;
            const uint8_t *pat = (const uint8_t *)arena + a_off[j];
            const size_t n = (size_t)(a_off[j + 1] - a_off[j]);
            const size_t sets_before = L.sets.sets.size() / 8;
            std::string e;
            int rc = mcx_parse_pattern(pat, n, normalize_dash, L.sets, irs[j], e);
            if (rc != 0) {
                L.bad.take((uint32_t)j, rc, "pattern " + std::to_string(j) + ": " + e);
                return; /* whatever follows in this range comes later in the table */
            }
            if (nseg_too_long(irs[j])) {
                L.bad.take((uint32_t)j, MCX_E_PATTERN, "pattern " + std::to_string(j) + ": segment longer than 65535 bytes");
                return;
            }
            for (size_t k = sets_before; k < L.sets.sets.size() / 8; ++k)
                L.set_first.push_back((uint32_t)j);
            if (irs[j].flags & MCX_PF_NEVER)
                continue; /* matches nothing: never a candidate */
            bool names_set = false;
            for (const Matcher &m : irs[j].m)
                names_set = names_set || m.kind == M_SET;
            if (names_set)
                L.with_sets.push_back((uint32_t)j);
            KeyOf &key = keys[j];
            key.len = 0;
            memset(key.b, 0, sizeof key.b);
            for (const Matcher &m : irs[j].seg(0)) {
                if (m.kind != M_LIT || key.len >= MCX_KEY_MAX)
                    break;
                key.b[key.len++] = m.byte;
            }
            live[j] = 1u;
        }
    });
    lap("parse");
    /* the table's set numbers: order of first appearance in the table = thread after thread */
    std::vector<std::vector<uint16_t>> remap(T);
    for (unsigned t = 0; t < T; ++t) {
        bad.take(loc[t].bad.j, loc[t].bad.rc, loc[t].bad.text);
        const size_t ns = loc[t].sets.sets.size() / 8;
        remap[t].resize(ns);
        for (size_t k = 0; k < ns; ++k) {
            std::array<uint32_t, 8> bits;
            std::copy(loc[t].sets.sets.begin() + 8 * k, loc[t].sets.sets.begin() + 8 * k + 8, bits.begin());
            uint32_t id = 0;
            if (intern_set(out, bits, &id) != 0) {
                const uint32_t j = loc[t].set_first[k];
                bad.take(j, MCX_E_PATTERN, "pattern " + std::to_string(j) + ": more than 65536 distinct bracket sets");
                break;
            }
            remap[t][k] = (uint16_t)id;
        }
    }
    if (bad.j != 0xFFFFFFFFu || limit < P) {
        if (bad.j != 0xFFFFFFFFu) {
            err = bad.text;
            return bad.rc;
        }
        err = "a_off not monotonic at pattern " + std::to_string(limit);
        return MCX_E_BAD_ARG;
    }
    on_threads(T, [&](unsigned t) {
        if (t == 0)
            return; /* the first thread's numbers are the table's */
        for (uint32_t j : loc[t].with_sets)
            for (Matcher &m : irs[j].m)
                if (m.kind == M_SET)
                    m.set = remap[t][m.set];
    });
    loc.clear();

    lap("sets");
    /* 4. copies of a text: how many, and which table positions (ascending) */
    std::vector<uint32_t> mult(P, 0u), rep_start(P + 1, 0u), rep;
    for (uint32_t j = 0; j < P; ++j) {
        const uint32_t f = first_of[j];
        if (!live[f])
            continue;
        mult[f]++;
        if (f != j)
            rep_start[f + 1]++;
    }
    for (uint32_t j = 0; j < P; ++j)
        rep_start[j + 1] += rep_start[j];
    rep.resize(rep_start[P]);
    {
        std::vector<uint32_t> fill(rep_start.begin(), rep_start.end() - 1);
        for (uint32_t j = 0; j < P; ++j) {
            const uint32_t f = first_of[j];
            if (f != j && live[f])
                rep[fill[f]++] = j;
        }
    }

    lap("copies");
    /* 5. index order: by key length, then key bytes, then table position */
    std::vector<uint32_t> order;
    order.reserve(P);
    for (uint32_t j = 0; j < P; ++j)
        if (first_of[j] == j && live[j])
            order.push_back(j);
    std::vector<uint32_t>().swap(first_of);
    const size_t n_idx = order.size();
    out.n_distinct = (uint32_t)n_idx;
    auto before = [&](uint32_t x, uint32_t y) {
        const int c = key_cmp(keys[x], keys[y]);
        return c != 0 ? c < 0 : x < y;
    };
    {
        std::vector<size_t> cut(T + 1);
        for (unsigned t = 0; t <= T; ++t)
            cut[t] = n_idx * t / T;
        on_threads(T, [&](unsigned t) { std::sort(order.begin() + cut[t], order.begin() + cut[t + 1], before); });
        for (unsigned width = 1; width < T; width *= 2) {
            const unsigned pairs = (T + 2 * width - 1) / (2 * width);
            on_threads(pairs, [&](unsigned p) {
                const unsigned a = p * 2 * width, m = a + width, b = std::min(T, a + 2 * width);
                if (m < b)
                    std::inplace_merge(order.begin() + cut[a], order.begin() + cut[m], order.begin() + cut[b], before);
            });
        }
    }

    lap("sort");
    /* 6. groups = runs of one key; the large ones get a number (second index) */
    struct Group {
        uint32_t start, count, gid, rest; /* gid: 0 = no second index, else number + 1 */
    };
    std::vector<Group> groups;
    for (size_t k = 0; k < n_idx; ++k) {
        if (k == 0 || key_cmp(keys[order[k]], keys[order[k - 1]]) != 0) {
            groups.push_back(Group{(uint32_t)k, 0u, 0u, 0u});
            if (out.lens.empty() || out.lens.back() != keys[order[k]].len)
                out.lens.push_back((uint32_t)keys[order[k]].len);
        }
        groups.back().count++;
    }
    std::vector<uint32_t> work; /* groups with something to settle, the large ones first */
    for (int pass = 0; pass < 2; ++pass)
        for (size_t g = 0; g < groups.size(); ++g) {
            Group &G = groups[g];
            if (pass == 0 && G.count > out.max_group)
                out.max_group = G.count;
            const bool large = G.count > sub_min;
            if (pass == 0 && large)
                G.gid = ++out.n_sub_groups;
            if (G.count > 1u && large == (pass == 0))
                work.push_back((uint32_t)g);
        }
    std::vector<std::vector<SubRun>> runs_of(out.n_sub_groups);
    std::vector<uint32_t> sig_choice(P, 0u); /* by pattern: the signature picked among its neighbours */
    std::vector<uint8_t> sig_chosen(P, 0u);
    {
        const size_t n_large = out.n_sub_groups;
        std::atomic<size_t> next{0};
        on_threads(T, [&](unsigned) {
            for (;;) {
                /* a large group a turn, then the small ones by the hundred */
                size_t a = next.load(std::memory_order_relaxed), b;
                do
                    b = a < n_large ? a + 1 : a + 128;
                while (!next.compare_exchange_weak(a, b, std::memory_order_relaxed));
                if (a >= work.size())
                    break;
                for (size_t w = a; w < std::min(b, work.size()); ++w) {
                    Group &G = groups[work[w]];
                    settle_group(irs, keys[order[G.start]].len, order.data(), G.start, (size_t)G.start + G.count,
                                 G.gid != 0u, G.gid - 1u, G.gid ? &runs_of[G.gid - 1u] : nullptr, &G.rest, sig_choice,
                                 sig_chosen);
                }
            }
        });
    }

    lap("groups");
    /* 7. slots, signatures and the copies' positions, in index order */
    out.slot.reserve(n_idx * (size_t)MCX_SLOT_WORDS + 4); /* room for the padding behind: no second copy */
    out.slot.assign(n_idx * (size_t)MCX_SLOT_WORDS, 0u);
    out.sig.reserve(n_idx + 4);
    out.sig.resize(n_idx);
    out.dup_start.assign(n_idx ? n_idx : 1, 0u);
    {
        size_t at = 0;
        for (size_t k = 0; k < n_idx; ++k)
            if (mult[order[k]] > 1u) {
                out.dup_start[k] = (uint32_t)at;
                at += mult[order[k]];
            }
        out.dup_idx.assign(at ? at : 1, 0u);
    }
    std::vector<std::vector<uint32_t>> ext_of(T);
    on_threads(T, [&](unsigned t) {
        auto r = range_of(t, n_idx);
        for (size_t k = r.first; k < r.second; ++k) {
            const uint32_t j = order[k];
            mcx_fill_slot(irs[j], (uint32_t)keys[j].len, j, mult[j], out.sets, &out.slot[k * (size_t)MCX_SLOT_WORDS],
                          ext_of[t]);
            out.sig[k] = sig_chosen[j] ? sig_choice[j] : sig_of(irs[j], keys[j].len);
            if (mult[j] > 1u) {
                uint32_t *d = &out.dup_idx[out.dup_start[k]];
                *d++ = j;
                for (uint32_t q = rep_start[j]; q < rep_start[j + 1]; ++q)
                    *d++ = rep[q];
            }
        }
    });
    for (unsigned t = 0; t < T; ++t) {
        /* bodies too long for a slot: one array, in index order */
        const uint32_t base = (uint32_t)out.ext.size();
        if (t > 0 && base != 0u && !ext_of[t].empty()) {
            auto r = range_of(t, n_idx);
            for (size_t k = r.first; k < r.second; ++k) {
                uint32_t *slot = &out.slot[k * (size_t)MCX_SLOT_WORDS];
                if (MCX_HDR_FLAGS(slot[0]) & MCX_PF_EXT)
                    slot[MCX_SLOT_DIR] += base;
            }
        }
        out.ext.insert(out.ext.end(), ext_of[t].begin(), ext_of[t].end());
    }
    lap("slots");
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
    std::vector<uint32_t> home(groups.size());
    on_threads(T, [&](unsigned t) {
        auto r = range_of(t, groups.size());
        for (size_t g = r.first; g < r.second; ++g) {
            const KeyOf &key = keys[order[groups[g].start]];
            uint32_t kw[MCX_KEY_MAX / 4] = {0};
            memcpy(kw, key.b, key.len);
            home[g] = mcx_key_hash(kw, (uint32_t)key.len);
        }
    });
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group &g = groups[gi];
        if (gi + 16 < groups.size())
            __builtin_prefetch(&out.slots[home[gi + 16] & (cap - 1)], 1);
        const KeyOf &key = keys[order[g.start]];
        uint32_t kw[MCX_KEY_MAX / 4] = {0};
        memcpy(kw, key.b, key.len);
        const uint32_t a = home[gi];
        size_t at = a & (cap - 1);
        while (out.slots[at].count != 0)
            at = (at + 1) & (cap - 1);
        mcx_slot_t sl = mcx_slot_t{a, (uint32_t)key.len, g.start, g.count, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}};
        memcpy(sl.key, kw, sizeof kw);
        if (g.gid) {
            sl.pad[0] = g.gid;
            sl.pad[1] = g.rest;
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
    lap("index");
    /* second index, never empty, one entry in four used */
    size_t n_runs = 0;
    for (const auto &rv : runs_of)
        n_runs += rv.size();
    size_t scap = 16;
    while (scap < 4 * n_runs)
        scap *= 2;
    out.sub.assign(scap, mcx_sub_t{0, 0, 0, 0});
    out.sub_tags.assign(scap + 3, 0u);
    for (const auto &rv : runs_of)
        for (const SubRun &r : rv) {
            const uint32_t h = mcx_sub_hash(r.gram, r.gid);
            size_t at = h & (scap - 1);
            while (out.sub[at].count != 0)
                at = (at + 1) & (scap - 1);
            out.sub[at] = mcx_sub_t{r.gram, r.gid, r.start, r.count};
            out.sub_tags[at] = (uint8_t)mcx_key_tag(h);
        }
    for (size_t at = 0; at < 3; ++at)
        out.sub_tags[scap + at] = out.sub_tags[at];
    lap("sub index");
    return 0;
}
