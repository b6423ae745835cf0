#!/bin/bash
# tools/encoder_node_diff.sh -- node lists of the omni encoder graphs: the REFERENCE's builders (audition.cpp / vision.cpp through oracle/_ref/omni-enc-min) against this
# repo's Python mirrors (llama.cpp-omni_amd/encoders.py, used by bench.py's `extras` and the round-2 tests), both as submitted to the plug-in (MI355X_DUMP_GRAPH).  GPU box.
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so MTMD_BACKEND_DEVICE=MI355X0
rm -f /tmp/ref_apm.dump /tmp/ref_vpm.dump /tmp/mir_apm.dump /tmp/mir_vpm.dump
python tools/make_synth_omni_gguf.py --module apm -o /tmp/apm.gguf > /dev/null; python tools/make_synth_omni_gguf.py --module vpm -o /tmp/vpm.gguf > /dev/null
MI355X_DUMP_GRAPH=/tmp/ref_apm.dump oracle/_ref/omni-enc-min apm /tmp/apm.gguf /tmp/a.bin --gpu --chunks 1 --frames 3000 > /dev/null 2>&1
MI355X_DUMP_GRAPH=/tmp/ref_vpm.dump oracle/_ref/omni-enc-min vpm /tmp/vpm.gguf /tmp/v.bin --gpu --chunks 1 > /dev/null 2>&1
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd import encoders as E
be = pkg.backend(0)
def run(path, build):
    os.environ["MI355X_DUMP_GRAPH"] = path
    c = pkg.Context(be); build(c); c.alloc(); be.graph_compute(c.graph()); be.synchronize(); c.free()
PY
MI355X_DUMP_GRAPH=/tmp/mir_apm.dump python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd import encoders as E
be = pkg.backend(0)
c = pkg.Context(be); W = E.whisper_weights(c, E.WHISPER, 24); E.whisper(c, E.WHISPER, W, 3000); c.alloc(); be.graph_compute(c.graph()); be.synchronize()
PY
MI355X_DUMP_GRAPH=/tmp/mir_vpm.dump python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd import encoders as E
be = pkg.backend(0)
c = pkg.Context(be); W = E.siglip2_weights(c, E.SIGLIP2, 27); inp, vit = E.siglip2(c, E.SIGLIP2, W)
Wr = E.resampler_weights(c, E.RESAMPLER); E.resampler(c, E.RESAMPLER, Wr, vit, (E.SIGLIP2["image"] // E.SIGLIP2["patch"]) ** 2)
c.alloc(); be.graph_compute(c.graph()); be.synchronize()
PY
{
echo "# tools/encoder_node_diff.sh: A = the reference's builder through ggml_backend_sched (last graph the plug-in received), B = this repo's Python mirror"
echo "## apm: Whisper-medium, 3000 mel frames in one call (A: audition.cpp build_whisper, first call of the streaming K/V cache; B: encoders.whisper)"
python tools/graph_diff.py /tmp/ref_apm.dump /tmp/mir_apm.dump
echo
echo "## vpm: SigLip2 27 blocks + resampler, one 448x448 slice (A: vision.cpp build_minicpmv; B: encoders.siglip2 + encoders.resampler)"
python tools/graph_diff.py /tmp/ref_vpm.dump /tmp/mir_vpm.dump
} > $OUT/encoder_node_diff.txt 2>&1
cat $OUT/encoder_node_diff.txt
