"""tg128 (llama-bench test_gen analogue, bench.py's Decoder) at the widths of the target's siblings, with the batch-1 kernels (mmv1.hip; the
TAIL instances when K % 4096 != 0) and with them switched off (the round-1 multi-column family).  Random Q4_K_M-mapped weights of each shape.
  python tools/decode_widths.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

MODELS = {
    "qwen3-4b":  dict(n_embd=2560, n_layer=36, n_head=32, n_head_kv=8, head_dim=128, n_ff=9728,  n_vocab=151936, rms_eps=1e-6, rope_base=1e6, n_ctx_orig=40960),
    "qwen3-8b":  dict(n_embd=4096, n_layer=36, n_head=32, n_head_kv=8, head_dim=128, n_ff=12288, n_vocab=151936, rms_eps=1e-6, rope_base=1e6, n_ctx_orig=40960),
    "llama3-8b-shape": dict(n_embd=4096, n_layer=32, n_head=32, n_head_kv=8, head_dim=128, n_ff=14336, n_vocab=128256, rms_eps=1e-5, rope_base=5e5, n_ctx_orig=8192),
    "qwen3-14b": dict(n_embd=5120, n_layer=40, n_head=40, n_head_kv=8, head_dim=128, n_ff=17408, n_vocab=151936, rms_eps=1e-6, rope_base=1e6, n_ctx_orig=40960),
}


def main():
    pkg = bench.load_pkg()
    from llama_cpp_omni_amd import qwen3
    be = pkg.Backend()
    for name, cfg in MODELS.items():
        types = qwen3.q4_k_m_types(cfg)
        for mv1 in (1, 0):
            be.set_option("mv1", mv1)
            d = bench.Decoder(pkg, be, cfg, types, n_ctx=512, n_kv=256)
            for t in range(16):
                d.step(t)
            t0 = time.perf_counter()
            for t in range(16, 144):
                d.step(t)
            dt = time.perf_counter() - t0
            print(f"{name:18s} mv1={mv1}  tg128 {128 / dt:7.1f} tok/s  ({dt / 128 * 1e3:.3f} ms/token, {be.get_stat('kernels_last_graph'):.0f} launches)", flush=True)
            d.g.free(); d.model.wctx.free()
    be.set_option("mv1", 1)


if __name__ == "__main__":
    main()
