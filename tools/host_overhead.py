"""tools/host_overhead.py -- where a decode step's wall time goes on the host side: time inside graph_compute (fingerprint + hipGraphLaunch,
asynchronous), the uploads, and the wait for the device."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench
pkg = bench.load_pkg()
from llama_cpp_omni_amd import qwen3
be = pkg.backend(0)
cfg = qwen3.QWEN3_8B; types = qwen3.q4_k_m_types(cfg)
dec = bench.Decoder(pkg, be, cfg, types, n_ctx=256, n_kv=256)
for p in range(8): dec.step(p)
tg = tu = ts = 0.0; N = 100
for p in range(8, 8 + N):
    t0 = time.perf_counter()
    I = dec.I
    dec.h_embd[:] = dec.embd_pool[p % 64]; dec.h_pos[0] = p; dec.h_idx[0] = p; dec.h_mask[:] = -np.inf; dec.h_mask[: p + 1] = 0.0
    be.tensor_set_async(I["inp_embd"], dec.h_embd); be.tensor_set_async(I["inp_pos"], dec.h_pos); be.tensor_set_async(I["k_idxs"], dec.h_idx)
    be.tensor_set_async(I["v_idxs"], dec.h_idx); be.tensor_set_async(I["kq_mask"], dec.h_mask)
    t1 = time.perf_counter()
    be.graph_compute(dec.graph)
    t2 = time.perf_counter()
    be.tensor_get_async(dec.logits, dec.h_logits); be.synchronize()
    t3 = time.perf_counter()
    tu += t1 - t0; tg += t2 - t1; ts += t3 - t2
print("per token (us): uploads + host prep %.1f | graph_compute call %.1f | read-back + wait %.1f | total %.1f; graph nodes %d" % (tu / N * 1e6, tg / N * 1e6, ts / N * 1e6, (tu + tg + ts) / N * 1e6, dec.graph.g.n_nodes))
# MI355X_GRAPH_GPU_TIME=1: the event pair the backend puts around every graph -- printed when the backend is freed
import os
if os.environ.get("MI355X_GRAPH_GPU_TIME"):
    be.synchronize()
    del dec
    be.close()
