"""Write a /lib/modules/<version>-like directory (modules.alias, modules.dep, a /proc/modules
text; real formats, synthetic content) with montecarlo_amd.synth.write_modules_dir:
    python scripts/make_modules_dir.py <dir> [patterns]
tests/test_gpu_parity.py::test_k1_real_host_tables builds one by itself where the box has no
kernel module tables; MCX_TEST_MODULES_DIR=<dir> points it at another."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import synth  # noqa: E402

out, P = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20000
names = synth.write_modules_dir(out, P)
print(out, P, len(names))
