"""oracle/ref_flags_probe.py LIB [OUT.npz] -- TEST INFRASTRUCTURE.  Loads ONE build of the reference CPU backend (oracle/_ref/libggml-ref.so, or oracle/_ref/native/libggml-ref.so =
the reference's own default flag set: -march=native, gnu11 / gnu++17, default floating-point contraction) and prints, as JSON, SHA-1 digests of what its quantisers write on
the golden signal of tests/golden/quant.npz (quantize_row_q8_K / _q8_0 of y and of the edge rows; ggml_quantize_chunk Q4_K / Q6_K / Q8_0 of x) and its vec_dot scalars on the
golden blocks.  One build per process: both libraries export the same symbols.  Used by tests/test_oracle.py::test_reference_build_flags_give_the_same_bytes."""
import ctypes as C, hashlib, json, os, sys
import numpy as np

lib = C.CDLL(sys.argv[1], mode=C.RTLD_LOCAL)
lib.ggml_cpu_init()                                              # the f16 -> f32 table the x86 vec_dot kernels read (as oracle/ref_backend.py does)
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "quant.npz"))
x, y, edge = g["x"].copy(), g["y"].copy(), g["edge"].copy()
K = x.size
out = {}
sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
vp = lambda a: a.ctypes.data_as(C.c_void_p)

q8k = np.zeros(K // 256 * 292, np.uint8); lib.quantize_row_q8_K(vp(y), vp(q8k), C.c_int64(K))
q80 = np.zeros(K // 32 * 34, np.uint8);   lib.quantize_row_q8_0(vp(y), vp(q80), C.c_int64(K))
e8k = np.zeros(4 * 292, np.uint8);        lib.quantize_row_q8_K(vp(edge), vp(e8k), C.c_int64(1024)); e8k.reshape(4, 292)[0, 260:] = 0
out["y_q8_K"], out["y_q8_0"], out["edge_q8_K"] = sha(q8k), sha(q80), sha(e8k)
out["y_q8_K_diff_bytes_vs_golden"] = int((q8k != g["y_q8_K"]).sum())
out["y_q8_0_diff_bytes_vs_golden"] = int((q80 != g["y_q8_0"]).sum())
out["edge_q8_K_diff_bytes_vs_golden"] = int((e8k != g["edge_q8_K"]).sum())

lib.ggml_quantize_chunk.restype = C.c_size_t
lib.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
lib.ggml_quantize_init.argtypes = [C.c_int]
vdot = {"q4_K": lib.ggml_vec_dot_q4_K_q8_K, "q6_K": lib.ggml_vec_dot_q6_K_q8_K, "q8_0": lib.ggml_vec_dot_q8_0_q8_0}
for name, (ty, blck, bsz) in {"q4_K": (12, 256, 144), "q6_K": (14, 256, 210), "q8_0": (8, 32, 34)}.items():
    lib.ggml_quantize_init(ty)
    blocks = np.zeros(K // blck * bsz, np.uint8)
    lib.ggml_quantize_chunk(ty, x.ctypes.data, blocks.ctypes.data, 0, 1, K, None)
    out[f"{name}_blocks"] = sha(blocks)
    out[f"{name}_blocks_diff_bytes_vs_golden"] = int((blocks != g[f"{name}_blocks"]).sum())
    f = vdot[name]
    f.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    gb = g[f"{name}_blocks"].copy(); act = (g["y_q8_0"] if name == "q8_0" else g["y_q8_K"]).copy()      # the GOLDEN operands: only the dot's own float arithmetic is compared
    dots = []
    for k in (256, 4096, 12288):
        s = C.c_float(0); f(k, C.byref(s), 0, gb.ctypes.data, 0, act.ctypes.data, 0, 1); dots.append(s.value)
    out[f"{name}_dots"] = [float(np.float32(d)) for d in dots]
    out[f"{name}_dots_rel_vs_golden"] = [float(abs(np.float32(d) - gd) / max(abs(gd), 1e-30)) for d, gd in zip(dots, g[f"{name}_dots"])]
# a wider net than the cosine: 2^20 seeded normal values at block-wise scales 2^-20 .. 2^20 (ties under nearest_int's magic-number rounding are where a contracted
# fma could differ from a separately rounded multiply + add)
rng = np.random.default_rng(20240601)
N = 1 << 20
r = (rng.standard_normal(N) * np.exp2(rng.integers(-20, 21, N // 256).repeat(256))).astype(np.float32)
r8k = np.zeros(N // 256 * 292, np.uint8); lib.quantize_row_q8_K(vp(r), vp(r8k), C.c_int64(N))
r80 = np.zeros(N // 32 * 34, np.uint8);   lib.quantize_row_q8_0(vp(r), vp(r80), C.c_int64(N))
out["rand_q8_K"], out["rand_q8_0"] = sha(r8k), sha(r80)
for name, (ty, blck, bsz) in {"q4_K": (12, 256, 144), "q6_K": (14, 256, 210)}.items():
    blocks = np.zeros((N // 16) // blck * bsz, np.uint8)
    lib.ggml_quantize_chunk(ty, r.ctypes.data, blocks.ctypes.data, 0, 1, N // 16, None)
    out[f"rand_{name}_blocks"] = sha(blocks)
if len(sys.argv) > 2:                                            # the arrays themselves, for a byte-wise comparison of two builds
    np.savez(sys.argv[2], rand=r, rand_q8_K=r8k, rand_q8_0=r80)
print(json.dumps(out))
