#!/usr/bin/env python3
"""tools/siglip_prof.py -- where a SigLip2 encoder layer's time goes: the mirror graph of vision.cpp (encoders.siglip2, N layers, default 4) submitted eagerly with the
backend's per-class event profile (option `profile`): one line per launch class (n, total us, average us), printed when the backend is freed (MI355X_LOG_STATS)."""
import os, sys
os.environ["MI355X_LOG_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
pkg = bench.load_pkg()
from llama_cpp_omni_amd import encoders as E
be = pkg.backend(0)
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c = pkg.Context(be)
W = E.siglip2_weights(c, E.SIGLIP2, nl); inp, vit = E.siglip2(c, E.SIGLIP2, W)
c.alloc()
rng = np.random.default_rng(1)
def flat(w):
    out = []
    for v in (w.values() if isinstance(w, dict) else w):
        out += flat(v) if isinstance(v, (dict, list)) else [v]
    return out
for t in flat(W) + [inp]:
    n = t.nelements(); v = (rng.standard_normal(n) * 0.05).astype(np.float32)
    be.tensor_set(t, v.astype(np.float16) if t.type == 1 else v)
g = c.graph()
for _ in range(2):
    be.graph_compute(g)
be.synchronize()
be.set_option("profile", 1)
for _ in range(3):
    be.graph_compute(g)
be.synchronize()
be.close()
