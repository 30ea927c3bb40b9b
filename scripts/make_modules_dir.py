"""Write a /lib/modules/<version>-like directory (modules.alias, modules.dep as text) from the
synthetic generators, for running tests/test_gpu_parity.py::test_k1_real_host_tables on a box
that has no kernel module tables: MCX_TEST_MODULES_DIR=<dir>."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import synth  # noqa: E402

out, P = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20000
os.makedirs(out, exist_ok=True)
A, a_off, a_mod = synth.alias_table(17, P)
V = synth.alias_modules(P)
names = synth.module_names(18, V)
raw = A.tobytes()
with open(os.path.join(out, "modules.alias"), "w") as f:
    f.write("# Aliases extracted from modules themselves.\n")
    for j in range(P):
        f.write(f"alias {raw[a_off[j]:a_off[j + 1]].decode()} {names[a_mod[j]]}\n")
rowptr, col = synth.dep_graph(19, V, 3)
# modules.dep lists everything a module needs, directly or not: closure of the direct edges


def closure(u, memo={}):
    if u in memo:
        return memo[u]
    memo[u] = []
    acc, seen = [], set()
    for v in col[rowptr[u]:rowptr[u + 1]]:
        for w in [int(v)] + closure(int(v)):
            if w not in seen:
                seen.add(w)
                acc.append(w)
    memo[u] = acc
    return acc


sys.setrecursionlimit(100000)
with open(os.path.join(out, "modules.dep"), "w") as f:
    for u in range(V):
        deps = " ".join(f"kernel/drivers/x/{names[w].replace('_', '-')}.ko.zst" for w in closure(u))
        f.write(f"kernel/drivers/x/{names[u].replace('_', '-')}.ko.zst:{' ' + deps if deps else ''}\n")
print(out, P, V)
