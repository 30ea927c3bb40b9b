#!/bin/bash
# SQ instruction counters and HIP-event times of the tile kernel for a list of library builds
# (paths as arguments; 10^6 strings x 30 000 patterns), one line per build.
set -u
OUT=gpurun_out/k1var
mkdir -p "$OUT"
export TMPDIR=/tmp K1_N=1000000 K1_REPS=3
i=0
for lib in "$@"; do
    i=$((i + 1))
    MONTECARLO_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
        SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
        --output-format csv -d "$OUT/run$i" -- python3 scripts/k1time.py > "$OUT/run$i.log" 2>&1 || exit 1
    MONTECARLO_LIB=$lib K1_REPS=30,30 timeout -k 10 120 python3 scripts/k1time.py > "$OUT/time$i.log" 2>&1 || exit 1
    python3 - "$OUT/run$i" "$lib" "$OUT/time$i.log" <<'PY'
import csv, glob, re, sys
d, lib, tlog = sys.argv[1:4]
vals = {}
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith("void k1_match_tiles"):
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
v = {c: sorted(x)[len(x) // 2] for c, x in vals.items()}
us = min(float(x) for x in re.findall(r"us/launch ([0-9.]+)", open(tlog).read()))
print("%-48s %7.1f us  VALU %6.1f  SALU %6.1f  LDS %5.1f  VMEM %5.2f  lanes %4.1f  busy %5.1f Mcyc" % (
    lib.split("/")[-1], us, v["SQ_INSTS_VALU"] / 1e6, v["SQ_INSTS_SALU"] / 1e6, v["SQ_INSTS_LDS"] / 1e6,
    v["SQ_INSTS_VMEM_RD"] / 1e6, v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_ACTIVE_INST_VALU"], 1), v["SQ_BUSY_CYCLES"] / 1e6))
PY
done
