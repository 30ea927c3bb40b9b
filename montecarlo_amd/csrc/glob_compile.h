/*
 * glob_compile.h - host-side compiler from fnmatch(3) pattern text to the
 * segment programs of mcx_prog.h, and builder of the literal-prefix index the
 * device probes.  Runs once per table load.
 */
#ifndef MCX_GLOB_COMPILE_H
#define MCX_GLOB_COMPILE_H

#include <stdint.h>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "mcx_prog.h"

enum { MCX_M_LIT = 0, MCX_M_ANY = 1, MCX_M_SET = 2 };
struct mcx_matcher {
    uint8_t kind;
    uint8_t byte;
    uint16_t set;
};

/* one '*'-free piece of a pattern: a view into its pattern's matchers */
struct mcx_seg {
    const mcx_matcher *p;
    size_t n;
    size_t size() const { return n; }
    const mcx_matcher &operator[](size_t k) const { return p[k]; }
    const mcx_matcher *begin() const { return p; }
    const mcx_matcher *end() const { return p + n; }
};

/* one parsed pattern: its matchers, segment after segment (two allocations a pattern: a
 * table of millions of patterns is parsed in the time the allocator leaves) */
struct mcx_pattern_ir {
    std::vector<mcx_matcher> m;    /* all segments' matchers                      */
    std::vector<uint32_t> seg_end; /* segment k = m[seg_end[k-1] .. seg_end[k])   */
    uint32_t flags = 0;            /* MCX_PF_* */
    uint32_t minlen = 0;
    size_t nsegs() const { return seg_end.size(); }
    mcx_seg seg(size_t k) const /* seg(0) is the anchored one */
    {
        const size_t b = k ? seg_end[k - 1] : 0u;
        return mcx_seg{m.data() + b, seg_end[k] - b};
    }
};

struct mcx_compiled_table {
    uint32_t P = 0;                  /* patterns in the source table          */
    uint32_t n_distinct = 0;         /* distinct pattern texts that can match  */
    std::vector<uint32_t> slot;      /* n_distinct x MCX_SLOT_WORDS, index order */
    std::vector<uint32_t> ext;       /* programs too long for a slot          */
    std::vector<uint32_t> sig;       /* n_indexed gram signatures, index order */
    std::vector<mcx_slot_t> slots;   /* open-addressing hash table, 2^k slots */
    std::vector<uint8_t> tags;       /* per slot: 0 = empty, else mcx_key_tag of the key's hash */
    std::vector<mcx_sub_t> sub;      /* (large group, anchor gram) -> run of patterns, 2^k entries */
    std::vector<uint8_t> sub_tags;   /* per entry of sub: 0 = empty, else mcx_key_tag of its hash */
    uint32_t n_sub_groups = 0;       /* groups that have a second index        */
    uint32_t max_group = 0;          /* patterns in the largest group          */
    std::vector<uint32_t> lens;      /* distinct key lengths, ascending       */
    std::vector<uint32_t> sets;      /* distinct bracket sets, 8 words each   */
    std::map<std::array<uint32_t, 8>, uint32_t> set_ids; /* set bits -> number */
    /* table positions of repeated pattern texts: the copies of the pattern in
     * slot k are dup_idx[dup_start[k] .. dup_start[k] + mult), ascending
     * (dup_start is only meaningful where mult > 1) */
    std::vector<uint32_t> dup_start, dup_idx;
    std::vector<uint8_t> hw_tag;     /* P bytes: pattern text starts with one
                                        of the six hardware bus tags          */
};

/* Parse one pattern.  Returns 0 or MCX_E_PATTERN and sets err.
 * normalize_dash applies kmod's '-' -> '_' rule outside [...]. */
int mcx_parse_pattern(const uint8_t *pat, size_t n, bool normalize_dash, mcx_compiled_table &sets_owner,
                      mcx_pattern_ir &out, std::string &err);

/* Append the program of one parsed pattern to `prog`: the header word, then
 * the body.  keylen = leading literals covered by the pattern's index key
 * (their whole chunks are left out of the body); sets = the table's 256-bit
 * bracket sets, eight words each, by set number.  seg_words, if given, receives
 * for every segment the word of `prog` its header was written to. */
void mcx_emit_program(const mcx_pattern_ir &ir, uint32_t keylen, const std::vector<uint32_t> &sets,
                      std::vector<uint32_t> &prog, std::vector<uint32_t> *seg_words = nullptr);

/* Fill the MCX_SLOT_WORDS words of one pattern's slot (orig = its position in
 * the caller's table, mult = how many table patterns have this text); a body
 * the slot cannot hold is appended to ext. */
void mcx_fill_slot(const mcx_pattern_ir &ir, uint32_t keylen, uint32_t orig, uint32_t mult,
                   const std::vector<uint32_t> &sets, uint32_t *slot, std::vector<uint32_t> &ext);

/* threads: host threads the compile may use (0 = one per core, at most 16; small tables use
 * fewer); sub_min: group size above which a group of the index gets the second (anchor-gram)
 * index (0 = MCX_SUB_MIN).  Neither changes what a table matches; the arrays do not depend
 * on the number of threads. */
struct mcx_compile_options {
    uint32_t threads = 0;
    uint32_t sub_min = 0;
};

/* Whole table: parse, group by literal prefix, emit programs in group order
 * and build the hash index. */
int mcx_compile_table(const char *arena, const uint32_t *a_off, uint32_t P, bool normalize_dash,
                      mcx_compiled_table &out, std::string &err, const mcx_compile_options *opt = nullptr);

#endif
