import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg, nmse
pkg = load_pkg()
from llama_cpp_omni_amd import qwen3
from oracle.ref_backend import make_ref_cpu_backend
from test_round2_gpu import _decode_run, W8
be = pkg.backend(0); ref_be = make_ref_cpu_backend(pkg, 16)
rng = np.random.default_rng(3); steps = 3
for cfg_name, cfg in (("W8", W8), ("W8-1layer", dict(W8, n_layer=1)), ("half", dict(W8, n_embd=2048, n_ff=4096, n_head=16, n_head_kv=4)), ("tinyD128", dict(qwen3.TINY, head_dim=128))):
    types = qwen3.q4_k_m_types(cfg)
    embd = rng.standard_normal((steps, cfg["n_embd"])).astype(np.float32)
    for n_kv in (32, 64, 256):
        ref, _ = _decode_run(pkg, ref_be, cfg, types, embd, steps, n_kv, False)
        got, k = _decode_run(pkg, be, cfg, types, embd, steps, n_kv, False)
        print(cfg_name, n_kv, [("%.1e" % nmse(got[t], ref[t])) if np.isfinite(got[t]).all() else "nan" for t in range(steps)])
