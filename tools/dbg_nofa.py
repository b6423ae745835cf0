import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg, nmse
pkg = load_pkg()
from llama_cpp_omni_amd import qwen3
from oracle.ref_backend import make_ref_cpu_backend
from test_round2_gpu import _decode_run, W8
be = pkg.backend(0); ref_be = make_ref_cpu_backend(pkg, 16)
types = qwen3.q4_k_m_types(W8)
rng = np.random.default_rng(3); steps = 4
embd = rng.standard_normal((steps, W8["n_embd"])).astype(np.float32)
ref, _ = _decode_run(pkg, ref_be, W8, types, embd, steps, 32, False)
for opts in ({}, {"mv1": 0}, {"graphs": 0}, {"fusion": 0}):
    got, k = _decode_run(pkg, be, W8, types, embd, steps, 32, False, dict(opts))
    print(opts, k, [("%.1e" % nmse(got[t], ref[t])) if np.isfinite(got[t]).all() else "nan" for t in range(steps)])
