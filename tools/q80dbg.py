import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg
pkg = load_pkg(); be = pkg.backend(0)
from llama_cpp_omni_amd import qwen3
from test_gpu_parity import run_graph
from oracle import oracle_py as orc
K, M = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 70000
ty = pkg.GGML_TYPE_Q8_0
rng = np.random.default_rng(1)
x = rng.standard_normal((1, K)).astype(np.float32)
wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
c = pkg.Context(be)
w = c.new_tensor(ty, K, M); xt = c.new_tensor(pkg.GGML_TYPE_F32, K, 1)
y = c.mul_mat(w, xt)
(res,) = run_graph(be, c, [y], [(w, wv), (xt, x)])
res = res.reshape(-1)
want = orc.mul_mat(ty, wv.view(np.uint8).reshape(M, -1), x).reshape(-1)
bad = np.where(~np.isfinite(res) | (np.abs(res - want) > 1e-3 * (np.abs(want) + 1e-2)))[0]
print("M", M, "kernels", be.get_stat("kernels_last_graph"), "bad", bad.size, "first", bad[:8], "last", bad[-8:])
if bad.size:
    d = np.diff(bad); print("runs:", bad[0], [int(v) for v in bad[1:][d > 1][:10]])
