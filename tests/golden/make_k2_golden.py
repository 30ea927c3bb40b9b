#!/usr/bin/env python3
"""Writes k2_golden.json: three dependency graphs with the flag bytes, refcounts and
closures they must give.  The chain and the diamond are worked out by hand below (the
script checks the oracle against them before it writes anything); the random graph's
expectations are the oracle's, frozen here so that a later change of the oracle shows.

    python tests/golden/make_k2_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from montecarlo_amd import synth  # noqa: E402

LOADED, BOUND, SAFE = 1, 2, 4


def csr(V, edges):
    rowptr = np.zeros(V + 1, dtype=np.uint32)
    for u, _ in edges:
        rowptr[u + 1] += 1
    rowptr = np.cumsum(rowptr, dtype=np.uint64).astype(np.uint32)
    col = np.array([v for _, v in sorted(edges)], dtype=np.uint32)
    return rowptr, col


def bitmap(V, members):
    bm = np.zeros((V + 31) // 32, dtype=np.uint32)
    for v in members:
        bm[v >> 5] |= np.uint32(1 << (v & 31))
    return bm


def case(name, V, edges, flags, seeds):
    rowptr, col = csr(V, edges)
    flags = np.array(flags, dtype=np.uint8)
    out, ref = oracle.dep_flags(rowptr, col, flags)
    rec = {"name": name, "V": V, "rowptr": rowptr.tolist(), "col": col.tolist(), "in_flags": flags.tolist(),
           "seeds": sorted(seeds), "out_flags": out.tolist(), "refcnt": ref.tolist()}
    for direction, key in ((0, "holders"), (1, "deps")):
        clo, edges_scanned, levels = oracle.dep_closure(rowptr, col, flags, bitmap(V, seeds), direction)
        members = [v for v in range(V) if clo[v >> 5] >> (v & 31) & 1]
        rec["closure_" + key] = {"members": members, "edges": int(edges_scanned), "levels": int(levels)}
    return rec


cases = []
# chain 3 -> 2 -> 1 -> 0 (u needs v), all loaded, 3 has a device, everything safe:
#   0, 1, 2 have a holder, so they are in use and not roots; 3 is in use (bound), no holder -> root
c = case("chain", 4, [(1, 0), (2, 1), (3, 2)], [LOADED | SAFE] * 3 + [LOADED | BOUND | SAFE], {1})
assert c["out_flags"] == [3, 3, 3, 6] and c["refcnt"] == [1, 1, 1, 1]
assert c["closure_holders"] == {"members": [1, 2, 3], "edges": 2, "levels": 3}
assert c["closure_deps"] == {"members": [0, 1], "edges": 1, "levels": 2}
cases.append(c)
# diamond 3 -> {1, 2} -> 0, plus 4 (not loaded) -> 0 and an isolated unsafe 5:
#   0 is held by 1 and 2 only (4 is not loaded): refcnt 2; 3 is a root; 5 is idle but not safe
c = case("diamond", 6, [(1, 0), (2, 0), (3, 1), (3, 2), (4, 0)],
         [LOADED | SAFE, LOADED | SAFE, LOADED | SAFE, LOADED | SAFE, SAFE, LOADED], {0})
assert c["out_flags"] == [3, 3, 3, 4, 0, 0] and c["refcnt"] == [2, 1, 1, 0, 0xFFFFFFFF, 0]
assert c["closure_holders"]["members"] == [0, 1, 2, 3] and c["closure_holders"]["levels"] == 3
assert c["closure_deps"]["members"] == [0]
cases.append(c)
rowptr, col = synth.dep_graph(11, 1000, 8)
flags = synth.dep_flags(11, rowptr, col, 300)
edges = [(u, int(col[k])) for u in range(1000) for k in range(rowptr[u], rowptr[u + 1])]
loaded = [v for v in range(1000) if flags[v] & 1]
cases.append(case("random1k", 1000, edges, flags.tolist(), set(loaded[:: max(1, len(loaded) // 5)][:5])))
with open(os.path.join(HERE, "k2_golden.json"), "w") as f:
    json.dump({"source": "make_k2_golden.py: chain and diamond by hand, random1k frozen from the oracle",
               "cases": cases}, f)
print("wrote", len(cases), "cases")
