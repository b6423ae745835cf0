import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0)
rng = np.random.default_rng(0)
REP = 20
for (nq, nkv, nh, masked) in [(512, 512, 32, True), (512, 512, 32, False), (512, 128, 32, False), (128, 512, 32, False), (512, 512, 8, False), (512, 2048, 32, True)]:
    D = 128; nhkv = max(1, nh // 4)
    c = Context(be)
    q = c.new_tensor(GGML_TYPE_F32, D, nq, nh, 1); k = c.new_tensor(GGML_TYPE_F16, D, nkv, nhkv, 1); v = c.new_tensor(GGML_TYPE_F16, D, nkv, nhkv, 1)
    m = c.new_tensor(GGML_TYPE_F16, nkv, nq) if masked else None
    ys = [c.flash_attn_ext(q, k, v, m, 1.0 / np.sqrt(D)) for _ in range(REP)]
    c.alloc()
    be.tensor_set(q, rng.standard_normal(q.nelements(), dtype=np.float32)); be.tensor_set(k, rng.standard_normal(k.nelements(), dtype=np.float32).astype(np.float16)); be.tensor_set(v, rng.standard_normal(v.nelements(), dtype=np.float32).astype(np.float16))
    if masked:
        mask = np.zeros((nq, nkv), np.float16); off = nkv - nq
        for i in range(nq): mask[i, off + i + 1:] = -np.inf
        be.tensor_set(m, mask)
    g = c.graph()
    for _ in range(2): be.graph_compute(g)
    be.synchronize(); best = 1e9
    for _ in range(5):
        a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
    print(f"nq={nq} nkv={nkv} nh={nh} masked={masked}: {best * 1e3 / REP:7.1f} us / node", flush=True)
    c.free()
