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
 *   word 0        minlen | flags | keylen | nseg (MCX_HDR)
 *   per segment   header: nchars (0..15) | nsets (16..31)
 *                 ceil(nchars/4) chunks of two words {value, mask}: string
 *                 bytes q+4c .. q+4c+3, read as one little-endian word w,
 *                 fit iff ((w ^ value) & mask) == 0.  A literal has mask byte
 *                 0xFF, '?' and set positions and the padding of the last
 *                 chunk have mask byte 0.
 *                 nsets words {position in segment (0..15) | set number
 *                 (16..31)}: the byte there must be in that 256-bit set.
 *
 * Storage: every indexed pattern owns one fixed slot of MCX_SLOT_WORDS words.
 * A program that fits lives in its slot; a longer one lives in an overflow
 * array and its slot holds only {header with MCX_PF_EXT set, word offset into
 * the overflow array}.
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
 * prefix of length len whose hash is (h1, h2); empty when count == 0 */
typedef struct {
    uint32_t h1, h2;
    uint32_t start; /* first pattern of the group, in index order */
    uint32_t count;
    uint32_t key[8]; /* the literal prefix itself, zero padded */
} mcx_slot_t;

/* Hash of a key of `len` <= MCX_KEY_MAX bytes held in little-endian words
 * kw[0..7] (bytes past `len` must be zero). */
MCX_HD void mcx_key_hash(const uint32_t *kw, uint32_t len, uint32_t &a, uint32_t &b)
{
    a = 0x811C9DC5u ^ (len * 0x9E3779B1u);
    b = 0x9747B28Cu + len;
    MCX_UNROLL
    for (uint32_t k = 0; k < MCX_KEY_MAX / 4u; ++k) {
        uint32_t w = (4u * k < len) ? kw[k] : 0u;
        a = (a ^ w) * 0x01000193u;
        a ^= a >> 15;
        b = (b + w) * 0x85EBCA6Bu;
        b ^= b >> 13;
    }
    a *= 0x7FEB352Du;
    a ^= a >> 16;
    b *= 0xC2B2AE35u;
    b ^= b >> 16;
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
 * Prog is any type whose operator[](w) returns word w of the program.
 */
template <class Prog, class Str>
MCX_HD bool mcx_seg_rest(const Prog &p, uint32_t chunks, uint32_t first_chunk, uint32_t nchunks,
                         uint32_t setlist, uint32_t nsets, const Str &s, uint32_t q, const uint32_t *sets)
{
    for (uint32_t c = first_chunk; c < nchunks; ++c)
        if ((s.word(q + 4u * c) ^ p[chunks + 2u * c]) & p[chunks + 2u * c + 1u])
            return false;
    for (uint32_t k = 0; k < nsets; ++k) {
        uint32_t e = p[setlist + k];
        uint32_t ch = s.byte(q + (e & 0xFFFFu));
        if (!((sets[(e >> 16) * 8u + (ch >> 5)] >> (ch & 31u)) & 1u))
            return false;
    }
    return true;
}

/*
 * Does the compiled pattern p (header at word `at`) match s[0..len) ?
 * Every segment is placed the same way: leftmost fit inside a window [lo, hi]
 * of start positions (a single position for the two anchored kinds), so that
 * lanes working on different patterns stay in one instruction stream.  When
 * the segment starts with a literal, the four start positions covered by one
 * string word are screened at once by a zero-byte test on word ^ literal.
 */
template <class Prog, class Str>
MCX_HD bool mcx_prog_match(const Prog &p, uint32_t at, const Str &s, uint32_t len, const uint32_t *sets)
{
    const uint32_t hdr = p[at];
    const uint32_t minlen = MCX_HDR_MINLEN(hdr), flags = MCX_HDR_FLAGS(hdr), nseg = MCX_HDR_NSEG(hdr);
    if (flags & MCX_PF_NEVER)
        return false;
    if (len < minlen)
        return false;
    if (!(flags & MCX_PF_HAS_STAR) && len != minlen)
        return false;

    uint32_t w = at + 1u;
    uint32_t pos = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        const uint32_t sh = p[w];
        const uint32_t nchars = sh & 0xFFFFu, nsets = sh >> 16, nchunks = (nchars + 3u) >> 2;
        const uint32_t chunks = w + 1u;
        const uint32_t setlist = chunks + 2u * nchunks;
        w = setlist + nsets;
        if (nchars == 0u)
            continue; /* only the anchored first segment can be empty */
        if (pos + nchars > len)
            return false;
        uint32_t lo = pos, hi = len - nchars;
        if (k == 0u)
            hi = 0u;
        else if (k == nseg - 1u && !(flags & MCX_PF_TRAIL_STAR))
            lo = hi;
        const uint32_t v0 = p[chunks], m0 = p[chunks + 1u];
        /* first matcher a literal: its byte, repeated, for the screen */
        const bool lit0 = (m0 & 0xFFu) == 0xFFu;
        const uint32_t rep = (v0 & 0xFFu) * 0x01010101u;
        bool found = false;
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
                if (mcx_seg_rest(p, chunks, 1u, nchunks, setlist, nsets, s, q + d, sets)) {
                    pos = q + d + nchars;
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
 * table: occ[b] is the 64-bit set of positions q with s[q] == b.  Anchored
 * segments are compared word by word at their one position (the first one
 * from `skip` bytes on: the caller has already compared those).  For a
 * floating segment the start positions at which it fits are the AND over its
 * matchers k of (positions of that matcher's bytes) >> k, and placing it is a
 * find-first-set above the previous segment's end: no loop over string
 * positions, so lanes running different patterns stay together.
 */
#define MCX_OCC_MAX_LEN 64u

template <class Prog, class Occ>
MCX_HD uint64_t mcx_seg_fits(const Prog &p, uint32_t chunks, uint32_t nchars, uint32_t setlist, uint32_t nsets,
                             const Occ &occ, uint64_t all, const uint32_t *sets)
{
    /* branch-free and one chunk ahead: the two words of chunk c+1 are requested
     * before the four table reads of chunk c are combined (reads of a program
     * word or two past the segment, and of table entry 0 for padding bytes, are
     * harmless: their results are masked) */
    const uint32_t nchunks = (nchars + 3u) >> 2;
    uint64_t M = all;
    uint32_t v = p[chunks], m = p[chunks + 1u];
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t vn = p[chunks + 2u * c + 2u], mn = p[chunks + 2u * c + 3u];
        const uint64_t o0 = occ[v & 0xFFu], o1 = occ[(v >> 8) & 0xFFu];
        const uint64_t o2 = occ[(v >> 16) & 0xFFu], o3 = occ[v >> 24];
        const uint32_t k0 = 4u * c;
        const uint32_t left = nchars - k0; /* matchers in this chunk and after */
        const uint64_t c0 = (m & 0xFFu) ? o0 : all;
        const uint64_t c1 = (m & 0xFF00u) ? o1 : all;
        const uint64_t c2 = (m & 0xFF0000u) ? o2 : all;
        const uint64_t c3 = (m & 0xFF000000u) ? o3 : all;
        const uint64_t keep = ~(uint64_t)0;
        M &= c0 >> k0; /* left >= 1 always */
        M &= (left > 1u) ? (c1 >> ((k0 + 1u) & 63u)) : keep;
        M &= (left > 2u) ? (c2 >> ((k0 + 2u) & 63u)) : keep;
        M &= (left > 3u) ? (c3 >> ((k0 + 3u) & 63u)) : keep;
        v = vn;
        m = mn;
    }
    for (uint32_t e = 0; e < nsets; ++e) {
        const uint32_t ent = p[setlist + e];
        uint64_t cls = 0;
        for (uint32_t wi = 0; wi < 8u; ++wi) {
            uint32_t bits = sets[(ent >> 16) * 8u + wi];
            while (bits) {
                cls |= occ[wi * 32u + (uint32_t)__builtin_ctz(bits)];
                bits &= bits - 1u;
            }
        }
        M &= cls >> (ent & 0xFFFFu);
    }
    return M;
}

template <class Prog, class Str, class Occ>
MCX_HD bool mcx_prog_match_occ(const Prog &p, uint32_t at, const Str &s, const Occ &occ, uint32_t len,
                               uint32_t skip, const uint32_t *sets)
{
    const uint32_t hdr = p[at];
    const uint32_t minlen = MCX_HDR_MINLEN(hdr), flags = MCX_HDR_FLAGS(hdr), nseg = MCX_HDR_NSEG(hdr);
    if (flags & MCX_PF_NEVER)
        return false;
    if (len < minlen)
        return false;
    if (!(flags & MCX_PF_HAS_STAR) && len != minlen)
        return false;
    const uint64_t all = len >= 64u ? ~(uint64_t)0 : (((uint64_t)1 << len) - 1u);

    uint32_t w = at + 1u;
    uint32_t pos = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        const uint32_t sh = p[w];
        const uint32_t nchars = sh & 0xFFFFu, nsets = sh >> 16, nchunks = (nchars + 3u) >> 2;
        const uint32_t chunks = w + 1u;
        const uint32_t setlist = chunks + 2u * nchunks;
        w = setlist + nsets;
        if (nchars == 0u)
            continue; /* only the anchored first segment can be empty */
        if (pos + nchars > len)
            return false;
        if (k == 0u) {
            if (!mcx_seg_rest(p, chunks, skip >> 2, nchunks, setlist, nsets, s, 0u, sets))
                return false;
            pos = nchars;
        } else if (k == nseg - 1u && !(flags & MCX_PF_TRAIL_STAR)) {
            if (!mcx_seg_rest(p, chunks, 0u, nchunks, setlist, nsets, s, len - nchars, sets))
                return false;
            pos = len;
        } else {
            uint64_t M = mcx_seg_fits(p, chunks, nchars, setlist, nsets, occ, all, sets);
            M &= ~(uint64_t)0 << pos;
            if (M == 0u)
                return false;
            pos = (uint32_t)__builtin_ctzll(M) + nchars;
        }
    }
    return true;
}

#endif /* MCX_PROG_H */
