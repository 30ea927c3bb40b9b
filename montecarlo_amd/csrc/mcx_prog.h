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
 *   word 0        minlen (bits 0..15) | flags (16..23) | nseg (24..31)
 *   per segment   header: nchars (0..15) | nsets (16..31)
 *                 ceil(nchars/4) chunks of two words {value, mask}: string
 *                 bytes q+4c .. q+4c+3, read as one little-endian word w,
 *                 fit iff ((w ^ value) & mask) == 0.  A literal has mask byte
 *                 0xFF, '?' and set positions and the padding of the last
 *                 chunk have mask byte 0.
 *                 nsets words {position in segment (0..15) | set number
 *                 (16..31)}: the byte there must be in that 256-bit set.
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

#define MCX_PF_TRAIL_STAR 1u
#define MCX_PF_HAS_STAR   2u
#define MCX_PF_NEVER      4u

#define MCX_MAX_PATTERN_BYTES 4096u
#define MCX_MAX_SEGMENTS      255u
#define MCX_KEY_MAX           32u   /* longest literal prefix used as index key */

#define MCX_HDR(minlen, flags, nseg) ((uint32_t)(minlen) | ((uint32_t)(flags) << 16) | ((uint32_t)(nseg) << 24))
#define MCX_HDR_MINLEN(w) ((w) & 0xFFFFu)
#define MCX_HDR_FLAGS(w) (((w) >> 16) & 0xFFu)
#define MCX_HDR_NSEG(w) ((w) >> 24)

/* one slot of the prefix index: a group of patterns that share the literal
 * prefix of length len whose hash is (h1, h2); empty when count == 0 */
typedef struct {
    uint32_t h1, h2;
    uint32_t start; /* first pattern of the group, in index order */
    uint32_t count;
} mcx_slot_t;

/* Hash of a key of `len` bytes: feed the bytes in order, then finish. */
MCX_HD void mcx_hash_init(uint32_t &a, uint32_t &b)
{
    a = 0x811C9DC5u;
    b = 0x9747B28Cu;
}
MCX_HD void mcx_hash_byte(uint32_t &a, uint32_t &b, uint32_t byte)
{
    a = (a ^ byte) * 0x01000193u;
    b = (b + byte + 1u) * 0x85EBCA6Bu;
    b ^= b >> 13;
}
MCX_HD void mcx_hash_finish(uint32_t &a, uint32_t &b, uint32_t len)
{
    a ^= len * 0x9E3779B1u;
    a ^= a >> 16;
    a *= 0x7FEB352Du;
    a ^= a >> 15;
    a *= 0x846CA68Bu;
    a ^= a >> 16;
    b ^= len;
    b *= 0xC2B2AE35u;
    b ^= b >> 16;
}

/*
 * Str is any type with
 *   uint32_t word(uint32_t q)  bytes q..q+3 as a little-endian word; bytes at
 *                              or beyond the string's length may hold anything
 *                              (they are always masked), but the read is safe
 *   uint32_t byte(uint32_t q)  byte q, q < length
 */
template <class Str>
MCX_HD bool mcx_seg_rest(const uint32_t *chunks, uint32_t first_chunk, uint32_t nchunks,
                         const uint32_t *setlist, uint32_t nsets, const Str &s, uint32_t q,
                         const uint32_t *sets)
{
    for (uint32_t c = first_chunk; c < nchunks; ++c)
        if ((s.word(q + 4u * c) ^ chunks[2u * c]) & chunks[2u * c + 1u])
            return false;
    for (uint32_t k = 0; k < nsets; ++k) {
        uint32_t e = setlist[k];
        uint32_t ch = s.byte(q + (e & 0xFFFFu));
        if (!((sets[(e >> 16) * 8u + (ch >> 5)] >> (ch & 31u)) & 1u))
            return false;
    }
    return true;
}

/* Does the compiled pattern at `prog` match the string s[0..len) ? */
template <class Str>
MCX_HD bool mcx_prog_match(const uint32_t *prog, const Str &s, uint32_t len, const uint32_t *sets)
{
    const uint32_t hdr = prog[0];
    const uint32_t minlen = MCX_HDR_MINLEN(hdr), flags = MCX_HDR_FLAGS(hdr), nseg = MCX_HDR_NSEG(hdr);
    if (flags & MCX_PF_NEVER)
        return false;
    if (len < minlen)
        return false;
    if (!(flags & MCX_PF_HAS_STAR) && len != minlen)
        return false;

    const uint32_t *p = prog + 1;
    uint32_t pos = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        const uint32_t sh = p[0];
        const uint32_t nchars = sh & 0xFFFFu, nsets = sh >> 16, nchunks = (nchars + 3u) >> 2;
        const uint32_t *chunks = p + 1;
        const uint32_t *setlist = chunks + 2u * nchunks;
        p = setlist + nsets;
        if (k == 0) {
            /* anchored at the start; len >= minlen >= nchars */
            if (!mcx_seg_rest(chunks, 0u, nchunks, setlist, nsets, s, 0u, sets))
                return false;
            pos = nchars;
        } else if (k == nseg - 1u && !(flags & MCX_PF_TRAIL_STAR)) {
            /* anchored at the end */
            if (pos + nchars > len)
                return false;
            if (!mcx_seg_rest(chunks, 0u, nchunks, setlist, nsets, s, len - nchars, sets))
                return false;
            pos = len;
        } else {
            /* floating: leftmost fit at or after pos (nchars >= 1 here) */
            if (pos + nchars > len)
                return false;
            const uint32_t last = len - nchars;
            const uint32_t v0 = chunks[0], m0 = chunks[1];
            bool found = false;
            for (uint32_t q = pos; q <= last; ++q) {
                if ((s.word(q) ^ v0) & m0)
                    continue;
                if (mcx_seg_rest(chunks, 1u, nchunks, setlist, nsets, s, q, sets)) {
                    pos = q + nchars;
                    found = true;
                    break;
                }
            }
            if (!found)
                return false;
        }
    }
    return true;
}

#endif /* MCX_PROG_H */
