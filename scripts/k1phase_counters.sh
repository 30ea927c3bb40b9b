#!/bin/bash
# SQ instruction counters of the tile kernel with its phases cut off one after the other
# (variant builds libmontecarlo_k1skip<n>.so under scripts/_bin): per-phase instruction counts
# and times follow by subtraction (scripts/k1phase_table.py prints the table).
set -u
OUT=gpurun_out/k1phase
mkdir -p "$OUT"
export TMPDIR=/tmp K1_N=1000000 K1_REPS=3
for k in 1 2 3 4 5; do
    lib=scripts/_bin/libmontecarlo_k1skip$k.so
    [ $k -eq 5 ] && lib=montecarlo_amd/lib/libmontecarlo.so
    MONTECARLO_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
        SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
        --output-format csv -d "$OUT/run$k" -- python3 scripts/k1time.py > "$OUT/run$k.log" 2>&1 || exit 1
    python3 - "$OUT/run$k" "$k" <<'PY'
import csv, glob, sys
d, k = sys.argv[1], sys.argv[2]
vals = {}
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith("void k1_match_tiles"):
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("skip", k, {c: sorted(v)[len(v) // 2] for c, v in sorted(vals.items())})
PY
    MONTECARLO_LIB=$lib K1_REPS=30,30 timeout -k 10 120 python3 scripts/k1time.py > "$OUT/time$k.log" 2>&1 || exit 1
done
python3 scripts/k1phase_table.py
