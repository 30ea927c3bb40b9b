/*
 * k1_census.cpp - development tool: what the large-batch alias matcher has to do
 * per string on a given table (hits, candidates, survivors of the gram screen,
 * matches), counted on the CPU with the product's own table compiler.
 * Reads a table and strings dumped by scripts/k1_census.py.  Not shipped.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../include/mcx.h"
#include "../montecarlo_amd/csrc/glob_compile.h"

static uint64_t g_words = 0; /* string words read: a measure of a program run's length */
struct HostStr {
    const uint8_t *p;
    uint32_t len;
    uint32_t byte(uint32_t q) const { return q < len ? p[q] : 0u; }
    uint32_t word(uint32_t q) const { ++g_words; return byte(q) | (byte(q + 1) << 8) | (byte(q + 2) << 16) | (byte(q + 3) << 24); }
};
struct HostProg {
    const uint32_t *p;
    uint32_t operator[](uint32_t w) const { return p[w]; }
};

static std::vector<uint8_t> slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    std::vector<uint8_t> v;
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    v.resize(ftell(f));
    fseek(f, 0, SEEK_SET);
    if (fread(v.data(), 1, v.size(), f) != v.size()) exit(1);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 1;
    auto A = slurp(argv[1]), AO = slurp(argv[2]), S = slurp(argv[3]), SO = slurp(argv[4]);
    const uint32_t *a_off = (const uint32_t *)AO.data(), *s_off = (const uint32_t *)SO.data();
    const uint32_t P = AO.size() / 4 - 1, N = SO.size() / 4 - 1;
    mcx_compiled_table t;
    std::string err;
    if (mcx_compile_table((const char *)A.data(), a_off, P, false, t, err)) { puts(err.c_str()); return 1; }
    printf("P=%u distinct=%u groups(lens)=%zu sub_groups=%u max_group=%u\n", P, t.n_distinct, t.lens.size(), t.n_sub_groups, t.max_group);
    const size_t mask = t.slots.size() - 1, smask = t.sub.size() - 1;
    uint64_t hits = 0, cand = 0, surv = 0, match = 0, sub_lookups = 0, sub_found = 0, surv_nogram = 0, bigcand = 0;
    std::map<uint32_t, uint64_t> surv_by_nseg, cand_by_gsize, runs_by_size, cand_by_runsize;
    std::map<std::string, uint64_t> fp_shape;
    std::map<uint64_t, uint64_t> cost_hist;
    std::vector<uint32_t> tile_pat;  /* patterns of the survivors of the current tile */
    uint64_t tiles = 0, tile_surv = 0, tile_distinct = 0, shared2 = 0, shared4 = 0, shared8 = 0, shared16 = 0;
    std::vector<uint32_t> tile_cost; /* run lengths of the survivors of the current tile of 64 strings, in order */
    uint64_t wave_sum = 0, wave_max64 = 0;
    uint32_t big_shown = 0;
    uint64_t surv_before = 0;
    for (uint32_t i = 0; i < N; ++i) {
        if (getenv("CENSUS_GREP") && i > 0 && strstr((const char *)S.data() + s_off[i - 1], getenv("CENSUS_GREP")) == (const char *)S.data() + s_off[i - 1])
            printf("  string %u: %llu survivors\n", i - 1, (unsigned long long)(surv - surv_before));
        surv_before = surv;
        if (i % 64 == 0 || i == N - 1) {
            /* a lane per survivor, 64 to a wave: a wave takes as long as its longest run */
            for (size_t a = 0; a < tile_cost.size(); a += 64) {
                uint32_t mx = 0;
                for (size_t b = a; b < std::min(a + 64, tile_cost.size()); ++b) {
                    wave_sum += tile_cost[b];
                    mx = std::max(mx, tile_cost[b]);
                }
                wave_max64 += 64ull * mx;
            }
            tile_cost.clear();
            /* how many of a tile's program runs share their pattern with other runs of the tile */
            if (!tile_pat.empty()) {
                std::map<uint32_t, uint32_t> by;
                for (uint32_t k : tile_pat) by[k]++;
                ++tiles;
                tile_surv += tile_pat.size();
                tile_distinct += by.size();
                for (auto &kv : by) {
                    if (kv.second >= 2) shared2 += kv.second;
                    if (kv.second >= 4) shared4 += kv.second;
                    if (kv.second >= 8) shared8 += kv.second;
                    if (kv.second >= 16) shared16 += kv.second;
                }
                tile_pat.clear();
            }
        }
        const uint32_t len = s_off[i + 1] - s_off[i];
        HostStr s{S.data() + s_off[i], len};
        uint32_t bloom[8] = {0};
        bloom[255 >> 5] |= 1u << 31;
        for (uint32_t q = 0; q + 4 <= len; ++q) { uint32_t b = mcx_gram_bit(s.word(q)); bloom[b >> 5] |= 1u << (b & 31); }
        for (uint32_t L : t.lens) {
            if (L > len) continue;
            uint32_t kw[8] = {0};
            for (uint32_t k = 0; k < L; ++k) kw[k >> 2] |= s.byte(k) << (8 * (k & 3));
            uint32_t a = mcx_key_hash(kw, L);
            for (size_t at = a & mask; t.slots[at].count; at = (at + 1) & mask) {
                const mcx_slot_t &sl = t.slots[at];
                if (sl.hash != a || sl.len != L || memcmp(sl.key, kw, 32)) continue;
                ++hits;
                std::vector<std::pair<uint32_t, uint32_t>> runs;
                if (sl.pad[0]) {
                    if (sl.pad[1]) runs.push_back({sl.start, sl.pad[1]});
                    std::vector<uint32_t> seen;
                    for (uint32_t q = 0; q + 4 <= len; ++q) {
                        uint32_t w = s.word(q), h = mcx_sub_hash(w, sl.pad[0] - 1);
                        ++sub_lookups;
                        for (size_t p = h & smask; t.sub[p].count; p = (p + 1) & smask)
                            if (t.sub[p].gram == w && t.sub[p].gid == sl.pad[0] - 1) {
                                bool dup = false;
                                for (uint32_t x : seen) dup |= x == t.sub[p].start;
                                if (!dup) { seen.push_back(t.sub[p].start); runs.push_back({t.sub[p].start, t.sub[p].count}); ++sub_found; }
                                break;
                            }
                    }
                } else runs.push_back({sl.start, sl.count});
                for (auto r : runs) {
                    if (getenv("CENSUS_BIGRUNS") && r.second > 64 && big_shown < 12) {
                        ++big_shown;
                        printf("  run of %u for %.*s:\n", r.second, (int)len, (const char *)S.data() + s_off[i]);
                        for (uint32_t k = r.first; k < r.first + r.second; k += r.second / 6) {
                            const uint32_t *sl2 = &t.slot[(size_t)k * MCX_SLOT_WORDS];
                            printf("      %.*s\n", (int)(a_off[sl2[1] + 1] - a_off[sl2[1]]), (const char *)A.data() + a_off[sl2[1]]);
                        }
                    }
                    cand += r.second;
                    uint32_t bucket = 1;
                    while (bucket < r.second) bucket *= 2;
                    runs_by_size[bucket]++;
                    cand_by_runsize[bucket] += r.second;
                    cand_by_gsize[sl.count <= 4 ? 4 : sl.count <= 64 ? 64 : sl.count <= 1024 ? 1024 : 100000] += r.second;
                    for (uint32_t k = r.first; k < r.first + r.second; ++k) {
                        if (!mcx_sig_pass(t.sig[k], bloom)) continue;
                        ++surv;
                        const uint32_t *slot = &t.slot[(size_t)k * MCX_SLOT_WORDS];
                        if (t.sig[k] == 0xFFFFFFFFu) ++surv_nogram;
                        surv_by_nseg[MCX_HDR_NSEG(slot[0])]++;
                        const uint64_t words0 = g_words;
                        bool ok = (MCX_HDR_FLAGS(slot[0]) & MCX_PF_EXT)
                                      ? mcx_prog_match(HostProg{t.ext.data()}, slot[MCX_SLOT_DIR], slot[0], s, len, t.sets.data())
                                      : mcx_prog_match(HostProg{slot}, MCX_SLOT_BODY, slot[0], s, len, t.sets.data());
                        tile_cost.push_back((uint32_t)(g_words - words0));
                        tile_pat.push_back(k);
                        cost_hist[std::min<uint64_t>(g_words - words0, 63)]++;
                        if (ok) ++match;
                        else if (fp_shape.size() < 100000) {
                            std::string p((const char *)A.data() + a_off[slot[1]], a_off[slot[1] + 1] - a_off[slot[1]]);
                            if (getenv("CENSUS_EXAMPLES") && (i < 40 || (getenv("CENSUS_GREP") && strstr((const char *)S.data() + s_off[i], getenv("CENSUS_GREP")) == (const char *)S.data() + s_off[i])))
                                printf("  miss: %s  vs  %.*s\n", p.c_str(), (int)len, (const char *)S.data() + s_off[i]);
                            /* shape: digits/hex replaced */
                            for (auto &c : p) if ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'F')) c = 'X';
                            fp_shape[p]++;
                        }
                    }
                }
                break;
            }
        }
    }
    printf("per string: hits %.2f candidates %.1f survivors %.2f (no gram at all %.2f) matches %.2f; sub lookups %.1f found %.2f\n",
           (double)hits / N, (double)cand / N, (double)surv / N, (double)surv_nogram / N, (double)match / N,
           (double)sub_lookups / N, (double)sub_found / N);
    for (auto &kv : cand_by_gsize) printf("  candidates from groups <= %u: %.2f per string\n", kv.first, (double)kv.second / N);
    printf("program runs: %.1f string words read per survivor; lanes busy if a wave waits for its longest run: %.1f of 64\n",
           (double)wave_sum / surv, 64.0 * wave_sum / wave_max64);
    printf("tiles of 64 strings: %.1f runs a tile over %.1f distinct patterns; runs whose pattern has >= 2 / 4 / 8 / 16 runs in the tile: %.1f%% / %.1f%% / %.1f%% / %.1f%%\n",
           (double)tile_surv / tiles, (double)tile_distinct / tiles, 100.0 * shared2 / tile_surv, 100.0 * shared4 / tile_surv,
           100.0 * shared8 / tile_surv, 100.0 * shared16 / tile_surv);
    for (auto &kv : cost_hist) printf("  runs reading %llu words: %.3f per string\n", (unsigned long long)kv.first, (double)kv.second / N);
    for (auto &kv : runs_by_size)
        printf("  runs of up to %u candidates: %.2f per string, %.2f candidates\n", kv.first, (double)kv.second / N,
               (double)cand_by_runsize[kv.first] / N);
    for (auto &kv : surv_by_nseg) printf("  survivors with %u segments: %.2f per string\n", kv.first, (double)kv.second / N);
    std::vector<std::pair<uint64_t, std::string>> v;
    for (auto &kv : fp_shape) v.push_back({kv.second, kv.first});
    std::sort(v.rbegin(), v.rend());
    for (size_t k = 0; k < v.size() && k < 25; ++k) printf("  false positive shape %.3f/string  %s\n", (double)v[k].first / N, v[k].second.c_str());
    return 0;
}
