#!/bin/bash
# Collect the round's profiles on the GPU box (run from the repository root):
#   kernel-trace statistics of the default bench.py run, then the PMC passes
#   (counters only, one pass per group, each a run of its own).
# Results land under gpurun_out/prof_<tag>/; scripts/profile_digest.py turns them
# into the summaries kept under profiles/.
set -u
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { # name, seconds, command...
    local name=$1 secs=$2
    shift 2
    echo "== $name" | tee -a "$OUT/steps.log"
    timeout -s KILL "$secs" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "rc=$rc" | tee -a "$OUT/steps.log"
    [ $rc -eq 0 ] || exit $rc
}
step bench_stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- \
    python3 bench.py --steps 200 --warmup 20
grep "^{" "$OUT/bench_stats.log" | tail -n 1 > "$OUT/bench_line_under_profiler.json" # the bench line (the profiler prints after it)
# HBM traffic of the bench's own kernels: the same command without its CPU legs and large cases
step bench_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_fetch" -- \
    python3 bench.py --no-cpu --no-fleet --no-configs3 --repeats 3
step bench_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_write" -- \
    python3 bench.py --no-cpu --no-fleet --no-configs3 --repeats 3
export K1_N=1000000 K1_REPS=5
step k1_tiles_sq1 180 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES \
    SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/k1_tiles_sq1" -- \
    python3 scripts/k1time.py
step k1_tiles_sq2 180 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    SQ_WAVES SQ_INSTS_VALU --output-format csv -d "$OUT/k1_tiles_sq2" -- python3 scripts/k1time.py
step k1_tiles_fetch 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/k1_tiles_fetch" -- python3 scripts/k1time.py
echo done | tee -a "$OUT/steps.log"
