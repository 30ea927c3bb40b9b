#!/bin/bash
# builds the four variant libraries scripts/k1phase_counters.sh runs (here, on the CPU)
set -e
cd "$(dirname "$0")/../montecarlo_amd/csrc"
for k in 1 2 3 4; do
    make -s variant SRC=mcx_k1.hip TAG=k1skip$k DEFS=-DK1T_SKIP=$k &
done
wait
