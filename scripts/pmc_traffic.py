"""Turn rocprofv3 PMC passes of scripts/k1time.py into profiles/k1_traffic.json.

On the GPU box (separate passes, counters only, as the MI355X guide prescribes):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 scripts/k1time.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 scripts/k1time.py
then here:  python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 1000 30000
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide streamed read, so it is doubled (an upper bound for narrow accesses).
"""
import csv
import glob
import json
import os
import sys


def per_launch(directory, counter, kernel="k1_match_small"):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {directory}")
    vals.sort()
    return vals[len(vals) // 2], len(vals)


def main():
    fetch_dir, write_dir, n, p = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch_kib, nf = per_launch(fetch_dir, "FETCH_SIZE")
    write_kib, nw = per_launch(write_dir, "WRITE_SIZE")
    rec = {
        "strings": n, "patterns": p, "kernel": "k1_match_small",
        "fetch_size_kib_median": fetch_kib, "write_size_kib_median": write_kib, "launches_sampled": [nf, nw],
        "hbm_bytes_per_launch": int(2 * fetch_kib * 1024 + write_kib * 1024),
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scripts/k1time.py; "
                "FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 bytes)",
    }
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "k1_traffic.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
