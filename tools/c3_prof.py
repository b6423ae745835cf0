#!/usr/bin/env python3
"""tools/c3_prof.py [ub512|ub2048|ub16384] -- BASELINE C3 (F16 8B, 8 x 2048 prompt tokens) leg of bench.py alone, with the eager per-class profile on stderr."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355X_BENCH_PROFILE", "1")
import bench
pkg = bench.load_pkg()
be = pkg.backend(0)
which = sys.argv[1] if len(sys.argv) > 1 else "ub16384"
if which == "ub16384": r = bench.c3_prefill(pkg, be, one_ubatch=True)
else: r = bench.c3_prefill(pkg, be, n_ubatch=int(which[2:]))
print(json.dumps(r))
