import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg, nmse
pkg = load_pkg()
from llama_cpp_omni_amd import encoders as E
from oracle.ref_backend import make_ref_cpu_backend
from test_round2_gpu import _fill, _flat_weights
which = sys.argv[1] if len(sys.argv) > 1 else "whisper"
be = pkg.backend(0); ref_be = make_ref_cpu_backend(pkg, 16)
res = []
for backend in (be, ref_be):
    rng = np.random.default_rng(7)
    c = pkg.Context(backend)
    if which == "whisper":
        W = E.whisper_weights(c, E.WHISPER, 1); inp, out = E.whisper(c, E.WHISPER, W, 3000)
    else:
        W = E.siglip2_weights(c, E.SIGLIP2, 1); inp, out = E.siglip2(c, E.SIGLIP2, W)
    c.alloc()
    ws = _flat_weights(W)
    sc = []
    for t in ws:
        if t.type == 1 or t.ne[1] > 1 and t.ne[0] > 8:
            fan = t.ne[0] * (t.ne[1] if len([d for d in t.ne if d > 1]) > 2 else 1)
            sc.append(1.0 / np.sqrt(fan))
        else:
            sc.append(0.1)
    _fill(backend, rng, ws, sc)
    backend.tensor_set(inp, rng.standard_normal(inp.nelements()).astype(np.float32))
    backend.graph_compute(c.graph())
    vals = []
    for t in c.nodes:
        if t.t.view_src:           # views alias their base
            vals.append(None); continue
        vals.append((int(t.t.op), tuple(t.ne), int(t.t.type), backend.tensor_get(t).copy()))
    res.append(vals)
    c.free()
for i, (a, b) in enumerate(zip(res[0], res[1])):
    if a is None: continue
    x, y = a[3].astype(np.float64), b[3].astype(np.float64)
    e = nmse(x, y) if np.isfinite(x).all() else float("nan")
    flag = " <<<<" if not (e < 1e-5) else ""
    print(i, "op", a[0], "ne", a[1], "type", a[2], "nmse %.2e%s" % (e, flag))
