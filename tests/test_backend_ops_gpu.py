"""The reference's OWN op-parity harness as driver-run tests (`-m gpu`): `oracle/_ref/test-backend-ops test -b MI355X0 -o <op>` -- built by oracle/Makefile.ref from
the reference's tests/test-backend-ops.cpp (the pin SURVEY.md section 8c calls authoritative: every case is computed by the reference CPU backend and by the
plug-in, NMSE bars of the reference, tests/test-backend-ops.cpp:1130-1290, 7149-7223) -- against libggml-mi355x.so as the reference's loader finds it through
GGML_BACKEND_PATH, every fusion on (the defaults).  One test per op; an op passes when the harness exits 0, reports no FAIL line and ran at least one supported
case.  FLASH_ATTN_EXT is 8240 cases of which the reference CPU backend's share takes ~1000 s, so it is sampled with the harness' own `-p` parameter filter:
every F16-cache case of head sizes 64 / 128 (the MFMA / one-token / DMA-ring kernels), the odd head sizes and quantised / BF16 / F32 caches of fattn_any.hip
at one depth.  tools/run_tbo.sh runs the unsampled set and writes profiles/rNN_test_backend_ops.txt.  Nothing here reads /root/reference."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TBO = os.path.join(ROOT, "oracle", "_ref", "test-backend-ops")
LIB = os.path.join(ROOT, "llama.cpp-omni_amd", "lib", "libggml-mi355x.so")

OPS = ["MUL_MAT", "ADD", "SUB", "MUL", "DIV", "RMS_NORM", "SCALE", "ROPE", "SOFT_MAX", "CPY", "CONT", "DUP", "GET_ROWS", "SET_ROWS",
       "SWIGLU", "REGLU", "GEGLU", "GEGLU_ERF", "GEGLU_QUICK", "ABS", "SGN", "NEG", "STEP", "TANH", "ELU", "RELU", "SIGMOID", "GELU", "GELU_QUICK", "SILU",
       "HARDSWISH", "HARDSIGMOID", "EXP", "GELU_ERF", "NORM", "IM2COL", "POOL_2D", "SQR", "SQRT", "LOG", "SIN", "COS", "CLAMP", "LEAKY_RELU", "CONCAT", "REPEAT",
       "PAD", "PAD_REFLECT_1D", "ARANGE", "TIMESTEP_EMBEDDING", "SUM_ROWS", "CONV_TRANSPOSE_1D"]
# (name, -p regex over the case's parameter string "hsk=..,hsv=..,nh=..,nr23=[..],kv=..,nb=..,mask=..,sinks=..,max_bias=..,logit_softcap=..,prec=..,type_KV=..,permute=[..]")
FA_SAMPLES = [("f16_heads_64_128", r"hsk=(64|128),.*type_KV=f16"),
              ("odd_head_sizes_f16", r"hsk=(40|80|96|192|256|576),.*kv=512,.*type_KV=f16"),
              ("other_cache_types_d128", r"hsk=128,hsv=128,.*kv=113,.*type_KV=(f32|bf16|q8_0|q4_0)")]

ANSI = re.compile(r"\x1b\[[0-9;]*m")


def _run(op, params=None, timeout=1500):
    if not os.path.exists(TBO):
        pytest.skip("oracle/_ref/test-backend-ops was not built (it is compiled where /root/reference exists and travels with the snapshot)")
    env = dict(os.environ)
    env["GGML_BACKEND_PATH"] = LIB
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "oracle", "_ref") + ":" + env.get("LD_LIBRARY_PATH", "")
    cmd = [TBO, "test", "-b", "MI355X0", "-o", op] + (["-p", params] if params else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, env=env, cwd=ROOT)
    out = ANSI.sub("", r.stdout)
    cases = [l for l in out.splitlines() if "): " in l]
    ok = sum(1 for l in cases if l.rstrip().endswith("OK"))
    fail = [l for l in cases if "FAIL" in l]
    return r.returncode, ok, fail, out


@pytest.mark.parametrize("op", OPS)
def test_reference_test_backend_ops(op):
    rc, ok, fail, out = _run(op)
    assert not fail, "\n".join(fail[:8])
    assert rc == 0, out[-2000:]
    assert ok > 0, f"no supported case of {op} ran on the plug-in:\n" + out[-1500:]


@pytest.mark.parametrize("name,params", FA_SAMPLES, ids=[s[0] for s in FA_SAMPLES])
def test_reference_test_backend_ops_flash_attn_sampled(name, params):
    rc, ok, fail, out = _run("FLASH_ATTN_EXT", params)
    assert not fail, "\n".join(fail[:8])
    assert rc == 0, out[-2000:]
    assert ok >= 20, (name, ok, out[-1500:])
