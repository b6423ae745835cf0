import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg, nmse
pkg = load_pkg()
from llama_cpp_omni_amd import qwen3
from oracle.ref_backend import make_ref_cpu_backend
be = pkg.backend(0); ref_be = make_ref_cpu_backend(pkg, 8)
cfg = dict(qwen3.TINY, head_dim=128); types = qwen3.q4_k_m_types(cfg)
rng = np.random.default_rng(3); embd = rng.standard_normal((3, cfg["n_embd"])).astype(np.float32)
res = {}
for name, b in (("ref", ref_be), ("gpu", be)):
    mdl = qwen3.Model(b, cfg, types, n_ctx=32, seed=11, flash_attn=False); mdl.taps = {}
    g, I, logits = mdl.build(1, 32); gr = g.graph()
    out = []
    for t in range(2):
        mdl.set_inputs(I, embd[t:t+1], t, 32); b.graph_compute(gr)
        out.append({k: b.tensor_get(v).copy() for k, v in mdl.taps.items()} | {"logits": b.tensor_get(logits).copy()})
    res[name] = out
for t in range(2):
    for k in res["ref"][t]:
        a, r = res["gpu"][t][k].astype(np.float32), res["ref"][t][k].astype(np.float32)
        print(t, k, a.shape, "finite" if np.isfinite(a).all() else "NAN(%d)" % (~np.isfinite(a)).sum(), "ref finite" if np.isfinite(r).all() else "ref NAN", "%.2e" % nmse(np.nan_to_num(a), np.nan_to_num(r)))
