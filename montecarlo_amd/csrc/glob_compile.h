/*
 * glob_compile.h - host-side compiler from fnmatch(3) pattern text to the
 * segment programs of mcx_prog.h.  Runs once per table load.
 */
#ifndef MCX_GLOB_COMPILE_H
#define MCX_GLOB_COMPILE_H

#include <stdint.h>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "mcx_prog.h"

struct mcx_compiled_table {
    std::vector<uint8_t> prog;        /* all programs, each 4-byte aligned     */
    std::vector<uint32_t> prog_off;   /* P+1 byte offsets into prog            */
    std::vector<mcx_prefix_t> prefix; /* P prefilter records                   */
    std::vector<uint32_t> meta;       /* P words, MCX_META(minlen, flags)      */
    std::vector<uint32_t> sets;       /* distinct bracket sets, 8 words each   */
    std::map<std::array<uint32_t, 8>, uint32_t> set_ids; /* set bits -> number */
    std::vector<uint8_t> hw_tag;      /* P bytes: pattern text starts with one
                                         of the six hardware bus tags          */
};

/* Compile one pattern; appends to out.  Returns 0 or MCX_E_PATTERN and sets
 * err.  normalize_dash applies kmod's '-' -> '_' rule outside [...]. */
int mcx_compile_pattern(const uint8_t *pat, size_t n, bool normalize_dash,
                        mcx_compiled_table &out, std::string &err);

int mcx_compile_table(const char *arena, const uint32_t *a_off, uint32_t P,
                      bool normalize_dash, mcx_compiled_table &out, std::string &err);

#endif
