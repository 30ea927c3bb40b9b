"""Print the kernel timeline of the last call in a rocprofv3 kernel trace (csv): name, start
relative to the first kernel shown, duration, gap to the previous kernel.  argv: trace directory,
then a piece of the name of the kernel a call starts with (default k2b_seed)."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k2b_seed"
last_seed = max(i for i, r in enumerate(rows) if first in r["Kernel_Name"])
t0 = int(rows[last_seed]["Start_Timestamp"])
prev_end = t0
for r in rows[last_seed:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-22s start %8.1f us  dur %7.1f us  gap %6.1f us  grid %s" % (
        r["Kernel_Name"].split("(")[0].replace("void ", "")[:22], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev_end = e
