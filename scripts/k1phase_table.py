"""Per-phase table of the tile kernel from the output of scripts/k1phase_counters.sh
(gpurun_out/k1phase/run<n>/ counters, gpurun_out/k1phase/time<n>.log HIP-event times):

    python scripts/k1phase_table.py [--write]     # --write: also profiles/k1_tiles_phases.json
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "k1phase")
NAMES = {1: "staging, gram sets, result store", 2: "probe", 3: "second index", 4: "screen", 5: "program runs"}


def main():
    raw, us = {}, {}
    for k in NAMES:
        vals = {}
        for path in glob.glob(os.path.join(SRC, f"run{k}", "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    if r["Kernel_Name"].startswith("void k1_match_tiles"):
                        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        raw[k] = {c: sorted(v)[len(v) // 2] for c, v in sorted(vals.items())}
        with open(os.path.join(SRC, f"time{k}.log")) as f:
            us[k] = min(float(x) for x in re.findall(r"us/launch ([0-9.]+)", f.read()))
    out = {"kernel": "k1_match_tiles<0>",
           "workload": "10^6 strings x 30000 patterns, one launch (scripts/k1time.py, K1_N=1000000)",
           "method": "builds that leave out the phases from n on (-DK1T_SKIP=n, make variant), rocprofv3 --pmc per "
                     "build (scripts/k1phase_counters.sh) and HIP-event time per launch; a phase = the difference "
                     "between two consecutive builds; per-string figures = these divided by 10^6",
           "cumulative": {f"up_to_{NAMES[k]}": dict(raw[k], us_per_launch=us[k]) for k in raw}, "per_phase": {}}
    prev, prev_us = {c: 0.0 for c in raw[1]}, 0.0
    print("%-34s %7s %7s %7s %6s %6s %6s" % ("phase", "us", "VALU", "SALU", "LDS", "VMEM", "lanes"))
    for k in sorted(raw):
        d = {c: raw[k][c] - prev[c] for c in raw[k]}
        row = {"us": round(us[k] - prev_us, 1),
               "valu_instructions_per_string": round(d["SQ_INSTS_VALU"] / 1e6, 1),
               "salu_instructions_per_string": round(d["SQ_INSTS_SALU"] / 1e6, 1),
               "lds_instructions_per_string": round(d["SQ_INSTS_LDS"] / 1e6, 1),
               "vmem_reads_per_string": round(d["SQ_INSTS_VMEM_RD"] / 1e6, 2),
               "lanes_active_of_64": round(d["SQ_THREAD_CYCLES_VALU"] / max(d["SQ_ACTIVE_INST_VALU"], 1.0), 1)}
        out["per_phase"][NAMES[k]] = row
        print("%-34s %7.1f %7.1f %7.1f %6.1f %6.2f %6.1f" % ((NAMES[k],) + tuple(row.values())))
        prev, prev_us = raw[k], us[k]
    print("%-34s %7.1f %7.1f %7.1f" % ("all", us[5], raw[5]["SQ_INSTS_VALU"] / 1e6, raw[5]["SQ_INSTS_SALU"] / 1e6))
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "profiles", "k1_tiles_phases.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
