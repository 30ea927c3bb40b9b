"""Turn the output of scripts/profile_round.sh (gpurun_out/prof_<tag>/) into the summaries kept
under profiles/: <tag>_bench.json, <tag>_bench_kernel_stats.csv, k1_traffic.json, k2_traffic.json,
k1_tiles_sq_counters.json.

    python scripts/profile_digest.py r2

FETCH_SIZE / WRITE_SIZE are KiB per launch; FETCH_SIZE is doubled for bytes (on gfx950 it counts the
128-byte requests of wide streamed reads as 64 bytes, so the result is an upper bound for narrow ones).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(directory):
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            yield from csv.DictReader(f)


def median(vals):
    vals = sorted(vals)
    return vals[len(vals) // 2]


def per_launch(directory, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in rows(directory)
            if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == counter]
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {directory}")
    return median(vals), len(vals)


def kernels_in(directory, prefix):
    return sorted({r["Kernel_Name"].split("(")[0] for r in rows(directory) if r["Kernel_Name"].startswith(prefix)})


def traffic(src, fetch, write, kernel):
    f, nf = per_launch(os.path.join(src, fetch), "FETCH_SIZE", kernel)
    w, nw = per_launch(os.path.join(src, write), "WRITE_SIZE", kernel)
    return {"fetch_size_kib_median": f, "write_size_kib_median": w, "launches_sampled": [nf, nw],
            "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --no-cpu --no-fleet "
            "--no-configs3 (scripts/profile_round.sh), median per "
            "launch in KiB; FETCH_SIZE doubled for bytes (gfx950 counts 128-byte requests as 64 bytes)")

    stats = glob.glob(os.path.join(src, "bench_stats", "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(stats[0], os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_line_under_profiler.json"), os.path.join(dst, tag + "_bench_under_profiler.json"))

    k1 = traffic(src, "bench_fetch", "bench_write", "k1_match_small")
    k1.update({"strings": 1000, "patterns": 30000, "kernel": "k1_match_small", "note": note})
    with open(os.path.join(dst, "k1_traffic.json"), "w") as f:
        json.dump(k1, f, indent=1)

    k2 = {"workload": "configs[2]: V=1000000 modules, E=10007521 edges (the k2 block of bench.py)", "note": note}
    total = 0
    for kern in kernels_in(os.path.join(src, "bench_fetch"), "k2_") + kernels_in(os.path.join(src, "bench_fetch"), "void k2_"):
        if kern.startswith("k2b_"):
            continue  # the closure's kernels: their traffic depends on the seeds, not a per-call figure
        k2[kern] = traffic(src, "bench_fetch", "bench_write", kern)
        total += k2[kern]["hbm_bytes_per_launch"]
    k2["algorithmic_bytes_per_call"] = 4 * (1000000 + 1) + 4 * 10007521 + 2 * 1000000
    k2["hbm_bytes_per_call"] = total
    with open(os.path.join(dst, "k2_traffic.json"), "w") as f:
        json.dump(k2, f, indent=1)

    tiles = [k for k in kernels_in(os.path.join(src, "k1_tiles_sq1"), "void k1_match_tiles")]
    sq = {"kernel": tiles, "workload": "10^6 strings x 30000 patterns, one launch (scripts/k1time.py, K1_N=1000000)",
          "note": "rocprofv3 --pmc, one pass per group of counters, median over the launches of the run; per-string "
                  "figures are these divided by 10^6", "counters": {}}
    for name in ("k1_tiles_sq1", "k1_tiles_sq2"):
        d = os.path.join(src, name)
        counters = sorted({r["Counter_Name"] for r in rows(d)})
        sq["counters"][name] = {c: per_launch(d, c, "void k1_match_tiles")[0] for c in counters}
    c1 = sq["counters"]["k1_tiles_sq1"]
    sq["derived"] = {
        "lanes_active_of_64": c1["SQ_THREAD_CYCLES_VALU"] / c1["SQ_ACTIVE_INST_VALU"],
        "valu_instructions_per_string": c1["SQ_INSTS_VALU"] / 1e6,
        "lds_bank_conflict_share_of_lds_active": c1["SQ_LDS_BANK_CONFLICT"] / max(c1["SQ_ACTIVE_INST_LDS"], 1.0),
    }
    f, n = per_launch(os.path.join(src, "k1_tiles_fetch"), "FETCH_SIZE", "void k1_match_tiles")
    sq["fetch_size_kib_median"] = f
    sq["hbm_read_bytes_per_launch_upper"] = int(2 * f * 1024)
    with open(os.path.join(dst, "k1_tiles_sq_counters.json"), "w") as fo:
        json.dump(sq, fo, indent=1)
    print(json.dumps({"k1": k1["hbm_bytes_per_launch"], "k2": total, "tiles": sq["derived"]}, indent=1))


if __name__ == "__main__":
    main()
