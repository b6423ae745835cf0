import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg, nmse
pkg = load_pkg()
from oracle.ref_backend import make_ref_cpu_backend
from test_gpu_parity import run_graph, _attn_f64
be = pkg.backend(0); ref_be = make_ref_cpu_backend(pkg, 16)
D, nq, nh, nhkv, nkv = 128, 1, 32, 8, 256
for nvalid in (2, 8, 100):
  for scale_q in (1.0, 8.0):
    rng = np.random.default_rng(nvalid)
    qv = (rng.standard_normal((1, nh, nq, D)) * scale_q).astype(np.float32)
    kv = rng.standard_normal((1, nhkv, nkv, D)).astype(np.float16)
    vv = rng.standard_normal((1, nhkv, nkv, D)).astype(np.float16)
    mask = np.zeros((64, nkv), np.float16); mask[:, nvalid:] = -np.inf
    res = []
    for b in (be, ref_be):
        c = pkg.Context(b)
        q = c.new_tensor(pkg.GGML_TYPE_F32, D, nq, nh, 1); k = c.new_tensor(pkg.GGML_TYPE_F16, D, nkv, nhkv, 1); v = c.new_tensor(pkg.GGML_TYPE_F16, D, nkv, nhkv, 1)
        m = c.new_tensor(pkg.GGML_TYPE_F16, nkv, 64)
        y = c.flash_attn_ext(q, k, v, m, 1.0 / np.sqrt(D))
        (got,) = run_graph(b, c, [y], [(q, qv), (k, kv), (v, vv), (m, mask)])
        res.append(got.reshape(1, nq, nh, D))
    want = _attn_f64(qv, kv, vv, mask, 1.0 / np.sqrt(D))
    print(nvalid, scale_q, "gpu-f64 %.2e ref-f64 %.2e gpu-ref %.2e" % (nmse(res[0], want), nmse(res[1], want), nmse(res[0], res[1])))
