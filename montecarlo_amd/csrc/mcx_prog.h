/*
 * mcx_prog.h - compiled form of one alias pattern, the routine that runs it
 * against one string, and the key hash of the literal-prefix index.
 *
 * A glob pattern (fnmatch flags = 0, C locale) is a sequence of single-byte
 * matchers (literal byte, '?', bracket set) cut into segments by '*'.  The
 * first segment is anchored at the start of the string, the last one at its
 * end unless the pattern ends in '*', every other segment floats and is placed
 * at its leftmost fit (leftmost placement of a fixed-length segment never
 * loses a match for a plain glob).  This is the decision glibc's fnmatch makes
 * (posix/fnmatch_loop.c, the '*' case), without its recursion.
 *
 * Encoding, 32-bit words:
 *   header        minlen | flags | keylen | nseg (MCX_HDR)
 *   body, per segment:
 *     segment header: nchars (bits 0..15) | nsets (16..29) | masked (31)
 *     chunks: the segment's matchers four at a time.  String bytes q+4c ..
 *       q+4c+3 are read as one little-endian word w; chunk c fits iff
 *       ((w ^ value) & mask) == 0.  A literal has mask byte 0xFF; '?', set
 *       positions and the padding of the last chunk have mask byte 0.  One
 *       value word per chunk; a segment of literals only needs no masks (they
 *       follow from nchars), a "masked" segment appends four mask bits per
 *       chunk (bit b: byte b is a literal), eight chunks to a word.
 *       The first segment omits its first keylen/4 chunks: a string reaches a
 *       pattern through the prefix index, which has compared those bytes.
 *     nsets entries of MCX_SET_WORDS words, one per bracket set in the segment:
 *       {position in segment (bits 0..15) | set number (16..31)},
 *       {n (0..7) | negated (8) | member 0 | member 1}, {members 2..5},
 *       {members 6..9}: the byte at that position must be one of the n member
 *       bytes (negated: none of them, and not NUL).  A set with more than ten
 *       members either way has n = MCX_SET_GENERAL and is looked up in the
 *       table's 256-bit sets by its number.
 *     A later segment of one to three literals or '?' (the "sv", "bc", "ic" of
 *     a modalias) is a single word, header and chunk at once: the bytes (bits
 *     0..23) | nchars (24..25) | which bytes are literals (26..28) | inline (30).
 *
 * Storage: every indexed pattern owns one slot of MCX_SLOT_WORDS words:
 *   {header, position of the pattern in the caller's table, multiplicity,
 *    four words of segment directory, body ...}
 * Patterns with the same text share one slot: its position is the smallest of
 * theirs and its multiplicity their number, which is all that first-match and
 * match-count need.
 * The directory holds, one byte each, the slot word at which the header of
 * segment 0..15 sits, so that the segments of one pattern can be evaluated by
 * different lanes.  A pattern with more than sixteen segments or a body that
 * does not fit lives in an overflow array, its slot being
 *   {header with MCX_PF_EXT, table position, word offset of the body there}.
 *
 * Gram signature: a string that matches a pattern contains every run of
 * literals of the pattern, hence every 4-byte window ("gram") of such a run.
 * Each string gets a 256-bit Bloom set of its grams; each pattern names up to
 * four Bloom bits of its own grams, and is skipped when one of them is missing
 * from the string's set.  Bit 255 is set for every string (patterns with
 * fewer than four grams point their spare entries at it).
 *
 * The same source is compiled for the device (the product path) and for the
 * host, where only the pattern compiler (key hash) and the test suite's CPU
 * twin of the device routine use it.
 */
#ifndef MCX_PROG_H
#define MCX_PROG_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MCX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define MCX_HD static inline
#endif

#if defined(__clang__)
#define MCX_UNROLL _Pragma("unroll")
#else
#define MCX_UNROLL
#endif

#define MCX_PF_TRAIL_STAR 1u
#define MCX_PF_HAS_STAR   2u
#define MCX_PF_NEVER      4u
#define MCX_PF_EXT        8u   /* slot holds {header, overflow offset} only */

#define MCX_SLOT_WORDS    32u
#define MCX_SLOT_MULT     2u   /* number of table patterns with this text */
#define MCX_SLOT_DIR      3u   /* first segment directory word of a slot */
#define MCX_SLOT_BODY     7u   /* first body word of a slot */
#define MCX_DIR_SEGS      16u  /* segments the directory can name */
#define MCX_SEG_MASKED     0x80000000u
#define MCX_SEG_INLINE     0x40000000u
#define MCX_SET_WORDS      4u
#define MCX_SET_MEMBERS    10u
#define MCX_SET_GENERAL    0xFFu
#define MCX_BLOOM_WORDS   8u
#define MCX_BLOOM_FREE    255u /* the bit every string has */

#define MCX_MAX_PATTERN_BYTES 4096u
#define MCX_MAX_SEGMENTS      255u
#define MCX_KEY_MAX           32u   /* longest literal prefix used as index key */

/* header word: minlen 0..12, flags 13..16, keylen 17..22, nseg 24..31.  keylen
 * = number of leading literals of the first segment that the index key of the
 * pattern's group covers (a string reaches the pattern only after those bytes
 * have been compared). */
#define MCX_HDR(minlen, flags, keylen, nseg) \
    ((uint32_t)(minlen) | ((uint32_t)(flags) << 13) | ((uint32_t)(keylen) << 17) | ((uint32_t)(nseg) << 24))
#define MCX_HDR_MINLEN(w) ((w) & 0x1FFFu)
#define MCX_HDR_FLAGS(w) (((w) >> 13) & 0xFu)
#define MCX_HDR_KEYLEN(w) (((w) >> 17) & 0x3Fu)
#define MCX_HDR_NSEG(w) ((w) >> 24)

/* one slot of the prefix index: a group of patterns that share the literal
 * prefix `key`; empty when count == 0.  64 bytes, so that a slot never
 * straddles a cache line. */
typedef struct {
    uint32_t hash;   /* mcx_key_hash of the key: a cheap first compare */
    uint32_t len;    /* key length in bytes */
    uint32_t start;  /* first pattern of the group, in index order */
    uint32_t count;
    uint32_t key[8]; /* the literal prefix itself, zero padded */
    uint32_t pad[4];
} mcx_slot_t;

/* Large groups (more than MCX_SUB_MIN patterns under one key: the prefix says
 * little, e.g. every "of:N*T*C..." line) carry a second index.  pad[0] of
 * the group's slot = group number + 1 (0: none), pad[1] = number of patterns
 * at the head of the group that have no anchor gram.  Every other pattern of
 * the group names one four-byte window of its literals that any string it
 * matches must hold (its anchor gram); the group is ordered by anchor gram, and
 * (group, gram) -> run of patterns is a second open-addressing table with tag
 * bytes of its own.  A caller that ignores it scans the whole group. */
#define MCX_SUB_MIN 128u
typedef struct {
    uint32_t gram;  /* the four bytes, first byte lowest */
    uint32_t gid;   /* group number */
    uint32_t start; /* first pattern of the run, in index order */
    uint32_t count; /* 0: empty entry */
} mcx_sub_t;

MCX_HD uint32_t mcx_sub_hash(uint32_t gram, uint32_t gid)
{
    uint32_t a = (gram ^ (gid * 0x9E3779B1u)) * 0x85EBCA6Bu;
    a ^= a >> 15;
    a *= 0x7FEB352Du;
    a ^= a >> 16;
    return a;
}

/* One byte per index slot, read before the 64-byte entry where misses are the
 * rule: 0 marks an empty slot, anything else is this function of the key's hash. */
MCX_HD uint32_t mcx_key_tag(uint32_t hash)
{
    return 1u + (hash >> 24) % 255u;
}

/* Hash of a key of `len` <= MCX_KEY_MAX bytes held in little-endian words
 * kw[0..7] (bytes past `len` must be zero): over the words the key reaches
 * into, no further - a probe that knows its key is short reads and mixes only
 * those.  It only spreads the keys over the table: a probe compares the key
 * itself. */
MCX_HD uint32_t mcx_key_hash_step(uint32_t a, uint32_t word)
{
    a = (a ^ word) * 0x01000193u;
    return a ^ (a >> 15);
}
MCX_HD uint32_t mcx_key_hash_start(uint32_t len)
{
    return 0x811C9DC5u ^ (len * 0x9E3779B1u);
}
MCX_HD uint32_t mcx_key_hash_end(uint32_t a)
{
    a *= 0x7FEB352Du;
    return a ^ (a >> 16);
}
MCX_HD uint32_t mcx_key_hash(const uint32_t *kw, uint32_t len)
{
    uint32_t a = mcx_key_hash_start(len);
    MCX_UNROLL
    for (uint32_t k = 0; k < MCX_KEY_MAX / 4u; ++k)
        a = 4u * k < len ? mcx_key_hash_step(a, kw[k]) : a;
    return mcx_key_hash_end(a);
}

/* Bloom bit (0..254) of the gram whose four bytes form the word g */
MCX_HD uint32_t mcx_gram_bit(uint32_t g)
{
    uint32_t h = g * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    return (h >> 24) % 255u;
}

/* does the 256-bit set `bloom` hold the four bits named by `sig`? */
template <class Bloom>
MCX_HD bool mcx_sig_pass(uint32_t sig, const Bloom &bloom)
{
    /* four independent reads, no early exit */
    uint32_t all = 1u;
MCX_UNROLL
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t bit = (sig >> (8u * k)) & 0xFFu;
        all &= bloom[bit >> 5] >> (bit & 31u);
    }
    return (all & 1u) != 0u;
}

/*
 * Str is any type with
 *   uint32_t word(uint32_t q)  bytes q..q+3 as a little-endian word; bytes at
 *                              or beyond the string's length may hold anything
 *                              (they are always masked), but reads up to 8
 *                              bytes past the end are safe
 *   uint32_t byte(uint32_t q)  byte q, q < length
 * Prog is any type whose operator[](w) returns word w of the program body;
 * reads of up to two words past the body are safe (their values are masked).
 */

/* four mask bits -> four mask bytes */
MCX_HD uint32_t mcx_expand_mask(uint32_t nib)
{
    return (((nib & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
}

/* layout of segment k whose header sits at word w */
struct mcx_seg_t {
    uint32_t nchars, nsets, nchunks, c0; /* c0: chunks the index key covers      */
    uint32_t values;                     /* word of chunk 0's value (if stored)  */
    uint32_t masks;                      /* first mask word of a masked segment  */
    uint32_t kind;                       /* 0 literals only, 1 masked, 2 inline  */
    uint32_t inline_mask;
    uint32_t setlist, next;
};

/* from the segment's header word sh, read at word w */
MCX_HD mcx_seg_t mcx_seg_from(uint32_t sh, uint32_t w, uint32_t k, uint32_t hdr)
{
    mcx_seg_t g;
    const bool inl = (sh & MCX_SEG_INLINE) != 0u;
    g.nchars = inl ? ((sh >> 24) & 3u) : (sh & 0xFFFFu);
    g.nsets = inl ? 0u : ((sh >> 16) & 0x3FFFu);
    g.kind = inl ? 2u : (sh >> 31);
    g.inline_mask = mcx_expand_mask(sh >> 26) & 0x00FFFFFFu;
    g.nchunks = (g.nchars + 3u) >> 2;
    g.c0 = k == 0u ? (MCX_HDR_KEYLEN(hdr) >> 2) : 0u;
    const uint32_t stored = g.nchunks - g.c0;
    /* an inline segment is its own value word: its mask hides the header bits */
    g.values = inl ? w : w + 1u - g.c0;
    g.masks = w + 1u + stored;
    g.setlist = inl ? w + 1u : g.masks + (g.kind == 1u ? (stored + 7u) / 8u : 0u);
    g.next = g.setlist + MCX_SET_WORDS * g.nsets;
    return g;
}

template <class Prog>
MCX_HD mcx_seg_t mcx_seg_at(const Prog &p, uint32_t w, uint32_t k, uint32_t hdr)
{
    return mcx_seg_from(p[w], w, k, hdr);
}

/* value and mask of chunk c (c0 <= c; a chunk past the last one, read ahead and
 * never used, gets whatever lies there) */
template <class Prog>
MCX_HD uint32_t mcx_chunk_mask(const Prog &p, const mcx_seg_t &g, uint32_t c)
{
    const uint32_t r = c - g.c0;
    const uint32_t mw = p[g.masks + (r >> 3)];
    const uint32_t left = g.nchars > 4u * c ? g.nchars - 4u * c : 0u;
    const uint32_t implied = left >= 4u ? 0xFFFFFFFFu : (left ? 0xFFFFFFFFu >> (8u * (4u - left)) : 0u);
    const uint32_t stored = mcx_expand_mask(mw >> (4u * (r & 7u)));
    return g.kind == 0u ? implied : (g.kind == 1u ? stored : g.inline_mask);
}

template <class Prog>
MCX_HD void mcx_chunk(const Prog &p, const mcx_seg_t &g, uint32_t c, uint32_t &v, uint32_t &m)
{
    v = p[g.values + c];
    const uint32_t r = c - g.c0;
    const uint32_t mw = p[g.masks + (r >> 3)];
    const uint32_t left = g.nchars > 4u * c ? g.nchars - 4u * c : 0u;
    const uint32_t implied = left >= 4u ? 0xFFFFFFFFu : (left ? 0xFFFFFFFFu >> (8u * (4u - left)) : 0u);
    const uint32_t stored = mcx_expand_mask(mw >> (4u * (r & 7u)));
    m = g.kind == 0u ? implied : (g.kind == 1u ? stored : g.inline_mask);
}

/* member i < MCX_SET_MEMBERS of a set entry whose words 1..3 are w1, w2, w3 */
MCX_HD uint32_t mcx_set_member(uint32_t w1, uint32_t w2, uint32_t w3, uint32_t i)
{
    const uint32_t w = i < 2u ? (w1 >> 16) : (i < 6u ? w2 : w3);
    const uint32_t sh = i < 2u ? 8u * i : (i < 6u ? 8u * (i - 2u) : 8u * (i - 6u));
    return (w >> sh) & 0xFFu;
}

/* is byte ch accepted by the set entry at word e of p ? */
template <class Prog>
MCX_HD bool mcx_set_has(const Prog &p, uint32_t e, uint32_t ch, const uint32_t *sets)
{
    const uint32_t w0 = p[e], w1 = p[e + 1u], w2 = p[e + 2u], w3 = p[e + 3u];
    const uint32_t n = w1 & 0xFFu;
    if (n == MCX_SET_GENERAL)
        return ((sets[(w0 >> 16) * 8u + (ch >> 5)] >> (ch & 31u)) & 1u) != 0u;
    bool in = false;
    for (uint32_t i = 0; i < n; ++i)
        in = in || mcx_set_member(w1, w2, w3, i) == ch;
    return ((w1 >> 8) & 1u) ? (!in && ch != 0u) : in;
}

template <class Prog, class Str>
MCX_HD bool mcx_seg_rest(const Prog &p, const mcx_seg_t &g, uint32_t first_chunk, const Str &s, uint32_t q,
                         const uint32_t *sets)
{
    for (uint32_t c = first_chunk; c < g.nchunks; ++c) {
        uint32_t v, m;
        mcx_chunk(p, g, c, v, m);
        if ((s.word(q + 4u * c) ^ v) & m)
            return false;
    }
    for (uint32_t k = 0; k < g.nsets; ++k) {
        const uint32_t e = g.setlist + MCX_SET_WORDS * k;
        if (!mcx_set_has(p, e, s.byte(q + (p[e] & 0xFFFFu)), sets))
            return false;
    }
    return true;
}

/*
 * Does the compiled pattern (header hdr, body at word `at` of p) match
 * s[0..len) ?  The anchored first segment is compared at position 0 (from the
 * chunk after the index key on); every other segment is placed at its leftmost
 * fit inside a window [lo, hi] of start positions (a single position for an
 * anchored last segment).  When the segment starts with a literal, the four
 * start positions covered by one string word are screened at once by a
 * zero-byte test on word ^ literal.
 */
MCX_HD bool mcx_is_lower(uint32_t c)
{
    return c - (uint32_t)'a' < 26u;
}

/* the set of positions of lower-case letters in s[0..len), len <= 64 (the host's way; the tile
 * kernel takes it from a ballot of the lanes that build the string's gram set) */
template <class Str>
MCX_HD uint64_t mcx_lower_set(const Str &s, uint32_t len)
{
    uint64_t m = 0;
    for (uint32_t q = 0; q < len && q < 64u; ++q)
        m |= (uint64_t)(mcx_is_lower(s.byte(q)) ? 1u : 0u) << q;
    return m;
}

/* `lower`: the set of positions q < 64 with a lower-case letter at s[q], for a string of at
 * most 64 bytes (pass have_lower = false otherwise).  In an alias table nearly every segment
 * behind a star starts with a field tag ("sv", "bc", "isc"): such a segment can only sit where
 * the string has a lower-case letter, and a modalias has a dozen of those - its start
 * positions are taken from the set, leftmost first, instead of being searched word by word. */
template <class Prog, class Str>
MCX_HD bool mcx_prog_match(const Prog &p, uint32_t at, uint32_t hdr, const Str &s, uint32_t len,
                           const uint32_t *sets, uint64_t lower = 0, bool have_lower = false)
{
    const uint32_t minlen = MCX_HDR_MINLEN(hdr), flags = MCX_HDR_FLAGS(hdr), nseg = MCX_HDR_NSEG(hdr);
    if (flags & MCX_PF_NEVER)
        return false;
    if (len < minlen)
        return false;
    if (!(flags & MCX_PF_HAS_STAR) && len != minlen)
        return false;

    uint32_t w = at;
    uint32_t pos = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        const mcx_seg_t g = mcx_seg_at(p, w, k, hdr);
        w = g.next;
        if (g.nchars == 0u)
            continue; /* only the anchored first segment can be empty */
        if (pos + g.nchars > len)
            return false;
        if (k == 0u) {
            if (!mcx_seg_rest(p, g, g.c0, s, 0u, sets))
                return false;
            pos = g.nchars;
            continue;
        }
        uint32_t lo = pos, hi = len - g.nchars;
        if (k == nseg - 1u && !(flags & MCX_PF_TRAIL_STAR))
            lo = hi;
        uint32_t v0, m0;
        mcx_chunk(p, g, 0u, v0, m0);
        /* first matcher a literal: its byte, repeated, for the screen */
        const bool lit0 = (m0 & 0xFFu) == 0xFFu;
        const uint32_t rep = (v0 & 0xFFu) * 0x01010101u;
        bool found = false;
        if (have_lower && lit0 && mcx_is_lower(v0 & 0xFFu)) {
            /* lo <= hi <= 63 here: the segment fits the string */
            uint64_t cand = lower & (~(uint64_t)0 << lo) & (~(uint64_t)0 >> (63u - hi));
            while (cand != 0u && !found) {
                const uint32_t q = (uint32_t)__builtin_ctzll(cand);
                cand &= cand - 1u;
                if ((s.word(q) ^ v0) & m0)
                    continue;
                if (mcx_seg_rest(p, g, 1u, s, q, sets)) {
                    pos = q + g.nchars;
                    found = true;
                }
            }
            if (!found)
                return false;
            continue;
        }
        for (uint32_t q = lo; q <= hi && !found; q += 4u) {
            const uint32_t w0 = s.word(q);
            uint32_t cand = 0x80808080u; /* bit 8d+7: start position q+d is worth a look */
            if (lit0) {
                const uint32_t x = w0 ^ rep;
                cand = (x - 0x01010101u) & ~x & 0x80808080u;
            }
            const uint32_t room = hi - q; /* positions q .. q+min(3,room) exist */
            if (room < 3u)
                cand &= 0xFFFFFFFFu >> (8u * (3u - room));
            while (cand != 0u && !found) {
                const uint32_t d = (uint32_t)(__builtin_ctz(cand) >> 3);
                cand &= cand - 1u;
                const uint32_t win = d ? s.word(q + d) : w0;
                if ((win ^ v0) & m0)
                    continue;
                if (mcx_seg_rest(p, g, 1u, s, q + d, sets)) {
                    pos = q + d + g.nchars;
                    found = true;
                }
            }
        }
        if (!found)
            return false;
    }
    return true;
}

/*
 * The same decision for a string of at most 64 bytes, with its occurrence
 * table: occ[b] is the 64-bit set of positions q with s[q] == b.  The start
 * positions at which a segment fits are the AND over its matchers k of
 * (positions of that matcher's bytes) >> k: one 64-bit set per segment, which
 * does not depend on the other segments, so the segments of a pattern can be
 * evaluated by different lanes (mcx_seg_mask) and put together afterwards
 * (mcx_step_*): the first segment must fit at 0, an anchored last one at
 * len - nchars, every other one at its first fit at or after the end of the
 * one before it.
 */
#define MCX_OCC_MAX_LEN 64u

/* NEVER flag and the length conditions every matcher checks first */
MCX_HD bool mcx_precheck(uint32_t hdr, uint32_t len)
{
    const uint32_t minlen = MCX_HDR_MINLEN(hdr), flags = MCX_HDR_FLAGS(hdr);
    if (flags & MCX_PF_NEVER)
        return false;
    if (len < minlen)
        return false;
    if (!(flags & MCX_PF_HAS_STAR) && len != minlen)
        return false;
    return true;
}

/* set of start positions at which segment g fits; chunks before g.c0 are the
 * index key's and count as fitting */
template <class Prog, class Occ>
MCX_HD uint64_t mcx_seg_mask_from(const Prog &p, const mcx_seg_t &g, uint32_t v, uint32_t m, const Occ &occ,
                                  uint64_t all, const uint32_t *sets);

template <class Prog, class Occ>
MCX_HD uint64_t mcx_seg_mask(const Prog &p, const mcx_seg_t &g, const Occ &occ, uint64_t all, const uint32_t *sets)
{
    uint32_t v, m;
    mcx_chunk(p, g, g.c0, v, m);
    return mcx_seg_mask_from(p, g, v, m, occ, all, sets);
}

/* v, m: value and mask of the segment's first stored chunk (chunk g.c0) */
template <class Prog, class Occ>
MCX_HD uint64_t mcx_seg_mask_from(const Prog &p, const mcx_seg_t &g, uint32_t v, uint32_t m, const Occ &occ,
                                  uint64_t all, const uint32_t *sets)
{
    /* branch-free and one chunk ahead: the words of chunk c+1 are requested
     * before the four table reads of chunk c are combined (reads of a program
     * word or two past the segment, and of table entry 0 for padding bytes, are
     * harmless: their results are masked) */
    uint64_t M = all;
    for (uint32_t c = g.c0; c < g.nchunks; ++c) {
        uint32_t vn, mn;
        mcx_chunk(p, g, c + 1u, vn, mn);
        const uint64_t o0 = occ[v & 0xFFu], o1 = occ[(v >> 8) & 0xFFu];
        const uint64_t o2 = occ[(v >> 16) & 0xFFu], o3 = occ[v >> 24];
        const uint32_t k0 = 4u * c;
        const uint32_t left = g.nchars - k0; /* matchers in this chunk and after */
        const uint64_t c0 = (m & 0xFFu) ? o0 : all;
        const uint64_t c1 = (m & 0xFF00u) ? o1 : all;
        const uint64_t c2 = (m & 0xFF0000u) ? o2 : all;
        const uint64_t c3 = (m & 0xFF000000u) ? o3 : all;
        const uint64_t keep = ~(uint64_t)0;
        M &= c0 >> (k0 & 63u); /* left >= 1 always */
        M &= (left > 1u) ? (c1 >> ((k0 + 1u) & 63u)) : keep;
        M &= (left > 2u) ? (c2 >> ((k0 + 2u) & 63u)) : keep;
        M &= (left > 3u) ? (c3 >> ((k0 + 3u) & 63u)) : keep;
        v = vn;
        m = mn;
    }
    for (uint32_t e = 0; e < g.nsets; ++e) {
        const uint32_t at = g.setlist + MCX_SET_WORDS * e;
        const uint32_t w0 = p[at], w1 = p[at + 1u], w2 = p[at + 2u], w3 = p[at + 3u];
        const uint32_t n = w1 & 0xFFu;
        uint64_t cls = 0; /* positions holding a byte of the set */
        if (n != MCX_SET_GENERAL) {
            for (uint32_t i = 0; i < n; ++i)
                cls |= occ[mcx_set_member(w1, w2, w3, i)];
            if ((w1 >> 8) & 1u)
                cls = all & ~cls & ~occ[0u];
        } else {
            /* the eight words of the set are asked for together (they live in global
             * memory), and the table is read for up to four members at a time: a
             * member-by-member loop is a chain of waits */
            uint32_t sw[8];
            MCX_UNROLL
            for (uint32_t wi = 0; wi < 8u; ++wi)
                sw[wi] = sets[(w0 >> 16) * 8u + wi];
            MCX_UNROLL
            for (uint32_t wi = 0; wi < 8u; ++wi) {
                uint32_t bits = sw[wi];
                while (bits) {
                    uint64_t o[4];
                    MCX_UNROLL
                    for (uint32_t t = 0; t < 4u; ++t) {
                        const bool any = bits != 0u;
                        const uint32_t b = any ? (uint32_t)__builtin_ctz(bits) : 0u;
                        const uint64_t v = occ[wi * 32u + b];
                        o[t] = any ? v : 0u;
                        bits &= bits - 1u;
                    }
                    cls |= (o[0] | o[1]) | (o[2] | o[3]);
                }
            }
        }
        M &= cls >> ((w0 & 0xFFFFu) & 63u);
    }
    /* a segment longer than the string fits nowhere (its shifts wrapped) */
    return g.nchars > 64u ? 0u : M;
}

/* slot word of the header of segment k < MCX_DIR_SEGS */
template <class Prog>
MCX_HD uint32_t mcx_dir_word(const Prog &slot, uint32_t k)
{
    return (slot[MCX_SLOT_DIR + (k >> 2)] >> (8u * (k & 3u))) & 0xFFu;
}

/*
 * Putting the segments together is a chain over the segments in order.  Its
 * state is {pos = where the next segment may start, ok}; one step places one
 * segment.  What a step needs besides the state is fixed per segment
 * (mcx_step_t), so on the device every lane prepares the step of its own
 * segment and the state travels from lane to lane.
 */
#define MCX_STATE_START 0x100u /* pos 0, ok */
#define MCX_STATE_OK(st) (((st) >> 8) & 1u)

struct mcx_step_t {
    uint64_t M;       /* fit set of the segment                                 */
    uint64_t wantfix; /* anchored segments: the one start position allowed       */
    uint32_t n;       /* segment length; 0 = nothing to place (state unchanged)  */
    uint32_t anchored;
};

/* the step that leaves every state as it is */
MCX_HD mcx_step_t mcx_step_none()
{
    mcx_step_t t;
    t.M = 0;
    t.wantfix = 0;
    t.n = 0;
    t.anchored = 0;
    return t;
}

/* step of segment k < nseg with fit set M and length n */
MCX_HD mcx_step_t mcx_step_make(uint32_t hdr, uint32_t len, uint32_t k, uint64_t M, uint32_t n)
{
    const uint32_t flags = MCX_HDR_FLAGS(hdr), nseg = MCX_HDR_NSEG(hdr);
    mcx_step_t t;
    t.M = M;
    t.n = n; /* only the first segment can be empty */
    const bool last_anchored = k == nseg - 1u && !(flags & MCX_PF_TRAIL_STAR);
    t.anchored = (k == 0u || last_anchored) ? 1u : 0u;
    t.wantfix = k == 0u ? (uint64_t)1 : ((uint64_t)1 << ((len - n) & 63u));
    return t;
}

MCX_HD uint32_t mcx_step_apply(const mcx_step_t &t, uint32_t len, uint32_t state)
{
    const uint32_t pos = state & 0xFFu, ok = state >> 8;
    const bool room = pos + t.n <= len; /* then pos <= 63 for n >= 1 */
    const uint64_t want = t.anchored ? t.wantfix : (~(uint64_t)0 << (pos & 63u));
    const uint64_t fit = t.M & want;
    const bool nz = fit != 0u;
    const uint32_t npos = (nz ? (uint32_t)__builtin_ctzll(fit) : 0u) + t.n;
    const uint32_t nok = (room && nz) ? ok : 0u;
    return t.n ? (npos | (nok << 8)) : state;
}

/* one caller doing all segments in turn: what the lanes of a group do together */
template <class Prog, class Occ>
MCX_HD bool mcx_slot_match_occ(const Prog &slot, const Occ &occ, uint32_t len, const uint32_t *sets)
{
    const uint32_t hdr = slot[0];
    if (!mcx_precheck(hdr, len))
        return false;
    const uint64_t all = len >= 64u ? ~(uint64_t)0 : (((uint64_t)1 << len) - 1u);
    const uint32_t nseg = MCX_HDR_NSEG(hdr);
    /* the whole directory first, then per segment its header word and the word
     * behind it together (the first chunk's value unless the segment is its own
     * value word): a caller reading the slot from far away waits once per
     * segment instead of three times */
    const uint32_t d0 = slot[MCX_SLOT_DIR], d1 = slot[MCX_SLOT_DIR + 1u], d2 = slot[MCX_SLOT_DIR + 2u],
                   d3 = slot[MCX_SLOT_DIR + 3u];
    uint32_t state = MCX_STATE_START;
    for (uint32_t k = 0; k < nseg; ++k) {
        const uint32_t dw = (k >> 2) == 0u ? d0 : ((k >> 2) == 1u ? d1 : ((k >> 2) == 2u ? d2 : d3));
        const uint32_t w = (dw >> (8u * (k & 3u))) & 0xFFu;
        const uint32_t sh = slot[w], nx = slot[w + 1u];
        const mcx_seg_t g = mcx_seg_from(sh, w, k, hdr);
        uint64_t M;
        if (g.kind == 2u) {
            /* most segments between two stars are one to three bytes and are
             * their own value word: three table reads, nothing else to fetch */
            const uint64_t o0 = occ[sh & 0xFFu], o1 = occ[(sh >> 8) & 0xFFu], o2 = occ[(sh >> 16) & 0xFFu];
            const uint32_t im = g.inline_mask;
            M = (im & 0xFFu) ? o0 : all;
            M &= g.nchars > 1u ? (((im & 0xFF00u) ? o1 : all) >> 1) : ~(uint64_t)0;
            M &= g.nchars > 2u ? (((im & 0xFF0000u) ? o2 : all) >> 2) : ~(uint64_t)0;
        } else {
            const uint32_t m = mcx_chunk_mask(slot, g, g.c0);
            M = mcx_seg_mask_from(slot, g, nx, m, occ, all, sets);
        }
        state = mcx_step_apply(mcx_step_make(hdr, len, k, M, g.nchars), len, state);
    }
    return MCX_STATE_OK(state) != 0u;
}

#endif /* MCX_PROG_H */
