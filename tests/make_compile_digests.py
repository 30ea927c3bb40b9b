"""Writes tests/golden/compile_digests.json: FNV digests of every array of the compiled alias
table (slots, index, second index, signatures, sets, copies ...) for a few seeded tables.
The digests on file were produced by the table compiler before it was split over threads;
the compiler of today, with one thread and with many, has to reproduce them bit for bit
(tests/test_oracle_cpu.py::test_compiled_tables_are_what_they_were).  Entries 3 and 4 of every
list (the prefix index and its tag bytes) were taken again when the key hash changed to mix
only the words a key reaches into: that moves the groups to other places of the hash table and
nothing else (the other eleven digests stayed as they were).

    python tests/make_compile_digests.py            # rewrite the file from today's compiler
                                                    # (sub_min = 5 entries: see CASES_SUB5)
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def splitmix(seed):
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def wild_table(seed, P):
    """metacharacter-heavy patterns: many '*' segments (some past a slot's directory and body),
    bracket sets of every size, '?', escapes, repeats, patterns that can never match, shared
    prefixes of every length"""
    r = splitmix(seed)
    alpha = b"abcdefXYZ0123456789:_-"
    pats = []
    for j in range(P):
        k = next(r)
        if pats and k % 11 == 0:
            pats.append(pats[next(r) % len(pats)])  # a repeat
            continue
        nseg = 1 + next(r) % (24 if k % 7 == 0 else 5)
        out = bytearray(b"bus%d:" % (next(r) % 5) if k % 3 else b"")
        for s in range(nseg):
            for _ in range(next(r) % (40 if k % 13 == 0 else 7)):
                c = next(r) % 20
                if c == 0:
                    out += b"?"
                elif c == 1:
                    n = 1 + next(r) % (14 if next(r) % 4 == 0 else 4)
                    body = bytes(alpha[next(r) % len(alpha)] for _ in range(n)).replace(b"-", b"q")
                    out += b"[" + (b"!" if next(r) % 3 == 0 else b"") + body + b"]"
                elif c == 2:
                    out += b"[0-9A-F]" if next(r) % 2 else b"[a-f]"
                elif c == 3:
                    out += b"\\" + bytes([alpha[next(r) % len(alpha)]])
                else:
                    out.append(alpha[next(r) % len(alpha)])
            if s + 1 < nseg or next(r) % 2:
                out += b"*"
        if k % 97 == 0:
            out += b"\\"  # trailing backslash: never matches
        if k % 101 == 0:
            out += b"[abc"  # unterminated bracket: literal
        pats.append(bytes(out))
    arena = b"".join(pats)
    a_off = np.zeros(P + 1, dtype=np.uint32)
    a_off[1:] = np.cumsum([len(p) for p in pats])
    return np.frombuffer(arena, dtype=np.uint8).copy(), a_off


CASES = (("synth", 2, 30000, False), ("synth", 4, 200000, False), ("synth", 9, 5000, True), ("synth", 11, 1, False),
         ("wild", 21, 4000, False), ("wild", 22, 20000, False))


def table_of(kind, seed, P):
    if kind == "synth":
        from montecarlo_amd import synth
        A, a_off, _ = synth.alias_table(seed, P)
        return A, a_off
    return wild_table(seed, P)


def digest(kind, seed, P, norm, threads=None, sub_min=None):
    import oracle
    tw = oracle.twin()
    A, a_off = table_of(kind, seed, P)
    tt = oracle.TwinTable(A, a_off, norm, **({} if threads is None else {"threads": threads}),
                          **({} if sub_min is None else {"sub_min": sub_min}))
    d = np.zeros(13, dtype=np.uint64)
    tw.twin_table_digest(tt.h, d.ctypes.data_as(ctypes.c_void_p))
    tt.close()
    return [int(x) for x in d]


CASES_SUB5 = (("synth", 2, 30000, False), ("wild", 22, 20000, False), ("synth", 9, 5000, True))

if __name__ == "__main__":
    res = {f"{kind}:{seed}:{P}:{int(norm)}": digest(kind, seed, P, norm) for kind, seed, P, norm in CASES}
    res.update({f"{kind}:{seed}:{P}:{int(norm)}:sub5": digest(kind, seed, P, norm, sub_min=5)
                for kind, seed, P, norm in CASES_SUB5})
    path = os.path.join(ROOT, "tests", "golden", "compile_digests.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=0)
    print("wrote", path)
