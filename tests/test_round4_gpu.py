"""Round-4 GPU tests (`-m gpu`).  Nothing here reads /root/reference.

* the persistent role-split decode engine LAB (tools/mmv3_engine.hip, tools/mmv3_lab.hip: the four mat-vec stages of a decode layer as ONE launch
  whose weight stream runs across the stage boundaries) produces every stage's output bit for bit as the product's four k_mv2 launches do --
  the evidence behind profiles/r04_engine_lab.txt.  The lab is compiled on the box (hipcc is part of the image) because it is not part of the
  product library: it measured slower than the launch form ("Round 4")."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_lab(tmp):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(tmp, "mmv3_lab")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=14", "-Wno-inline-asm",
           os.path.join(ROOT, "tools", "mmv3_lab.hip"), "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return exe


@pytest.mark.parametrize("stages", [(0, 3), (1, 2)])
def test_persistent_engine_lab_is_bit_identical_to_the_launch_form(tmp_path, stages):
    """wo + resid -> [norm] gate / up + SwiGLU -> down + resid -> [norm] wq / wk / wv, Q4_K and Q6_K layer variants, three weight sets each: every
    element of x2, h, x3, q, k, v equal between one k_mv3 launch and the k_mv2 launches; no bounded wait of the engine may give up."""
    exe = _build_lab(str(tmp_path))
    r = subprocess.run([exe, str(stages[0]), str(stages[1])], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 0, out[-3000:]
    assert "GAVE UP" not in out and "MISMATCH" not in out and "RESULTS DIFFER" not in out, out[-3000:]
    assert out.count("results identical") == 2, out[-3000:]                   # both layer variants
    assert out.count(" identical (0 /") == 2 * 3 * (stages[1] - stages[0] + 1), out[-3000:]
