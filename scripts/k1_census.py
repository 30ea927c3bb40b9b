"""dump a synthetic table and strings for scripts/k1_census.cpp and run it"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd import synth
seed, P, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A, a_off, _ = synth.alias_table(seed, P)
S, s_off = synth.modalias_strings(int(os.environ.get("CENSUS_STRING_SEED", seed + 3)), P, N)
d = "/tmp/k1census"
os.makedirs(d, exist_ok=True)
for name, arr in (("A", A), ("ao", a_off), ("S", S), ("so", s_off)):
    np.ascontiguousarray(arr).tofile(os.path.join(d, name))
exe = os.path.join(ROOT, "scripts", "_bin", "k1_census")
subprocess.run([exe] + [os.path.join(d, n) for n in ("A", "ao", "S", "so")], check=True)
