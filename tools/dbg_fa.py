import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg, nmse
pkg = load_pkg()
from llama_cpp_omni_amd import qwen3
from oracle.ref_backend import make_ref_cpu_backend
from test_round2_gpu import _decode_run, W8
be = pkg.backend(0); ref_be = make_ref_cpu_backend(pkg, 16)
types = qwen3.q4_k_m_types(W8)
rng = np.random.default_rng(3); steps = 8
embd = rng.standard_normal((steps, W8["n_embd"])).astype(np.float32)
ref, _ = _decode_run(pkg, ref_be, W8, types, embd, steps, 256, True)
new, k1 = _decode_run(pkg, be, W8, types, embd, steps, 256, True)
old, k2 = _decode_run(pkg, be, W8, types, embd, steps, 256, True, {"fattn_one": 0})
old2, k3 = _decode_run(pkg, be, W8, types, embd, steps, 256, True, {"fattn_one": 0, "mv1": 0})
nofa_ref, _ = _decode_run(pkg, ref_be, W8, types, embd, steps, 32, False)
print("kernels", k1, k2, k3)
for t in range(steps):
    print(t, "new-ref %.2e old-ref %.2e old2-ref %.2e new-old %.2e | fa ref vs nofa ref %.2e new vs nofa ref %.2e" % (nmse(new[t], ref[t]), nmse(old[t], ref[t]), nmse(old2[t], ref[t]), nmse(new[t], old[t]), nmse(ref[t], nofa_ref[t]), nmse(new[t], nofa_ref[t])))
