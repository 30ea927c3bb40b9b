/*
 * mcx_prog.h - compiled form of one alias pattern and the routine that runs
 * it against one string.
 *
 * A glob pattern (fnmatch flags = 0, C locale) is a sequence of single-byte
 * matchers (literal byte, '?', bracket set) cut into segments by '*'.  The
 * first segment is anchored at the start of the string, the last one at its
 * end unless the pattern ends in '*', every other segment floats and is placed
 * at its leftmost fit (leftmost placement of a fixed-length segment never
 * loses a match for a plain glob).  This is the decision glibc's fnmatch makes
 * (posix/fnmatch_loop.c, the '*' case), without its recursion.
 *
 * Encoding, little endian, 4-byte aligned per pattern:
 *   u16 minlen   total number of matchers (shortest string that can match)
 *   u8  flags    MCX_PF_*
 *   u8  nseg     number of segments (segment 0 may be empty)
 *   nseg x { u16 nchars; u16 nbytes; u8 code[nbytes]; }
 * Matcher codes: a byte > 3 is that literal; 1 = any byte; 2,lo,hi = set
 * number lo|hi<<8 (256-bit bitmap, 8 words, in the table's set array);
 * 3,b = literal b (for b <= 3).
 *
 * The same source is compiled for the device (the product path) and for the
 * host, where only the test suite's CPU twin of the device routine uses it.
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

#define MCX_OP_ANY 1u
#define MCX_OP_SET 2u
#define MCX_OP_ESC 3u

#define MCX_MAX_PATTERN_BYTES 4096u
#define MCX_MAX_SEGMENTS      255u
#define MCX_PREFIX_BYTES      16u

/* per-pattern record read by the prefilter: the first bytes of the anchored
 * literal prefix, a byte mask saying how many of them are literal, and the
 * length window a matching string must fall in. */
typedef struct {
    uint32_t pfx[4];
    uint32_t mask[4];
} mcx_prefix_t;

/* meta word: minlen in bits 0..15, flags in 16..23 */
#define MCX_META(minlen, flags) ((uint32_t)(minlen) | ((uint32_t)(flags) << 16))
#define MCX_META_MINLEN(m) ((m) & 0xFFFFu)
#define MCX_META_FLAGS(m) (((m) >> 16) & 0xFFu)

/* Run one matcher; pc advances past it. */
template <class Str>
MCX_HD bool mcx_code_match(const uint8_t *&pc, const Str &s, uint32_t at,
                           const uint32_t *sets)
{
    uint32_t op = *pc++;
    uint32_t ch = s[at];
    if (op > 3u)
        return op == ch;
    if (op == MCX_OP_ANY)
        return true;
    if (op == MCX_OP_ESC) {
        uint32_t b = *pc++;
        return b == ch;
    }
    uint32_t id = (uint32_t)pc[0] | ((uint32_t)pc[1] << 8);
    pc += 2;
    return (sets[id * 8u + (ch >> 5)] >> (ch & 31u)) & 1u;
}

template <class Str>
MCX_HD bool mcx_seg_match(const uint8_t *codes, uint32_t nchars, const Str &s,
                          uint32_t at, const uint32_t *sets)
{
    const uint8_t *pc = codes;
    for (uint32_t k = 0; k < nchars; ++k)
        if (!mcx_code_match(pc, s, at + k, sets))
            return false;
    return true;
}

/* Does the compiled pattern at `prog` match the string s[0..len) ? */
template <class Str>
MCX_HD bool mcx_prog_match(const uint8_t *prog, const Str &s, uint32_t len,
                           const uint32_t *sets)
{
    uint32_t minlen = (uint32_t)prog[0] | ((uint32_t)prog[1] << 8);
    uint32_t flags = prog[2];
    uint32_t nseg = prog[3];
    if (flags & MCX_PF_NEVER)
        return false;
    if (len < minlen)
        return false;
    if (!(flags & MCX_PF_HAS_STAR) && len != minlen)
        return false;

    const uint8_t *p = prog + 4;
    uint32_t pos = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
        uint32_t nchars = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
        uint32_t nbytes = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
        const uint8_t *codes = p + 4;
        p = codes + nbytes;
        if (k == 0) {
            /* anchored at the start; len >= minlen >= nchars */
            if (!mcx_seg_match(codes, nchars, s, 0u, sets))
                return false;
            pos = nchars;
        } else if (k == nseg - 1u && !(flags & MCX_PF_TRAIL_STAR)) {
            /* anchored at the end */
            if (pos + nchars > len)
                return false;
            if (!mcx_seg_match(codes, nchars, s, len - nchars, sets))
                return false;
            pos = len;
        } else {
            /* floating: leftmost fit at or after pos */
            bool found = false;
            for (uint32_t q = pos; q + nchars <= len; ++q) {
                if (mcx_seg_match(codes, nchars, s, q, sets)) {
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
