#!/usr/bin/env python3
"""tools/attn_sm_bench.py [D H nq nkv] -- the soft-max attention chain of an encoder (K.q -> SOFT_MAX -> V^T.p -> PERMUTE -> CONT, K / V as f16 tensors [D, nkv, H] / [nkv, D, H]) as the
fused launch runs it: us per chain between two HIP events (REP chains per graph).  MI355X_FA_STAMPS=1 prints workgroup 0's time line."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
be = pkg.backend(0); be.set_option("graphs", 0)
D, H, nq, nkv = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 16, 1500, 1500)
F32, F16 = pkg.GGML_TYPE_F32, pkg.GGML_TYPE_F16
rng = np.random.default_rng(0)
REP = 12
c = pkg.Context(be)
qc = c.new_tensor(F32, D, H, nq); k16 = c.new_tensor(F16, D, H, nkv); v16 = c.new_tensor(F16, nkv, D, H)
outs = []
for _ in range(REP):
    q = c.permute(qc, 0, 2, 1, 3); k = c.permute(k16, 0, 2, 1, 3)
    p = c.soft_max_ext(c.mul_mat(k, q), None, 1.0 / np.sqrt(D), 0.0)
    outs.append(c.cont(c.permute(c.mul_mat(v16, p), 0, 2, 1, 3), D * H, nq))
c.alloc()
be.tensor_set(qc, rng.standard_normal(D * H * nq).astype(np.float32)); be.tensor_set(k16, rng.standard_normal(D * H * nkv).astype(np.float16)); be.tensor_set(v16, rng.standard_normal(D * H * nkv).astype(np.float16))
g = c.graph()
for _ in range(2): be.graph_compute(g)
be.synchronize(); best = 1e9
for _ in range(5):
    a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
fl = 4.0 * D * H * nq * nkv
print(f"D={D} H={H} nq={nq} nkv={nkv}: {best * 1e3 / REP:7.1f} us per chain, {fl / (best * 1e-3 / REP) / 1e12:6.1f} TFLOP/s, {int(be.get_stat('kernels_last_graph'))} launches for {REP} chains", flush=True)
