#!/bin/bash
# tools/t2w_report.sh -- the Token2Wav evidence of a round in one file (stdout): the reference's Token2WavSession (oracle/_ref/t2w-min) on the plug-in, device time per
# window graph, each round-5 matcher switched off in turn, every launch of one DiT block timed in-graph, the small f32 GEMM shapes, eager rocprofv3 kernel stats.  GPU box.
set -u
cd "$(dirname "$0")/.."
python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
echo "# commit $(cat gpurun_out/.commit 2>/dev/null)"
echo "## 1. oracle/_ref/t2w-min /tmp/t2w out.f32 gpu --windows 8 (MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1): device time per graph as nodes:ms -- 27194 = a window's DiT graph, 2373 = its vocoder graph"
MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1 oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 8 2>&1 | grep -E "device time|kernels in last|host time|replayed graphs|^\{" | cut -c1-400
echo
echo "## 2. one matcher off at a time (6 windows; read the replayed 27194-node graphs)"
for e in MI355X_NO_CONV_FUSE=1 MI355X_NO_CONCAT_TAIL=1 MI355X_NO_ATTN_F32=1 MI355X_NO_GEMM_F32_T16=1 MI355X_NO_NORM_FUSE=1 MI355X_NO_GATE_NORM=1 MI355X_NO_CONV1D_TC=1 MI355X_NO_EW_CHAIN=1 MI355X_EW_CHAIN_NO_V1=1 MI355X_NO_CONT_SINK=1; do
  echo "-- $e"; env $e MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1 oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 6 2>&1 | grep -E "device time|kernels in last|^\{" | cut -c1-330
done
echo
echo "## 3. tools/t2w_slices.sh: every launch of one DiT block as its own capture (10 copies per replay, 20 replays): us | node op launches folded shape sources"
tools/t2w_slices.sh 2>&1 | cut -c1-200
echo
echo "## 4. tools/gemm_f32_bench.py (in-graph, 48 different weight tensors per shape); then with the 16 x 16-tile kernel off"
python tools/gemm_f32_bench.py
echo "-- MI355X_NO_GEMM_F32_T16=1"; MI355X_NO_GEMM_F32_T16=1 python tools/gemm_f32_bench.py
echo
echo "## 5. tools/t2w_profile.sh: eager (MI355X_GRAPHS=0) rocprofv3 --kernel-trace --stats, prompt set-up + 4 windows"
tools/t2w_profile.sh t2w_r05 > /dev/null 2>&1; head -32 gpurun_out/t2w_r05/kernels.txt
