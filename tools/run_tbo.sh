#!/bin/bash
# Run the reference's own op-parity harness (oracle/_ref/test-backend-ops, built by oracle/Makefile.ref from
# /root/reference/tests/test-backend-ops.cpp) against libggml-mi355x.so on the GPU box.
# usage: tools/run_tbo.sh [op ...]   (default: every op the backend claims)
cd "$(dirname "$0")/.."
export GGML_BACKEND_PATH="$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so"
export LD_LIBRARY_PATH="$PWD/oracle/_ref:$LD_LIBRARY_PATH"
mkdir -p gpurun_out
OPS="$@"
[ -z "$OPS" ] && OPS="MUL_MAT ADD SUB MUL DIV RMS_NORM SCALE ROPE SOFT_MAX CPY CONT DUP GET_ROWS SET_ROWS FLASH_ATTN_EXT SWIGLU REGLU GEGLU GEGLU_ERF GEGLU_QUICK ABS SGN NEG STEP TANH ELU RELU SIGMOID GELU GELU_QUICK SILU HARDSWISH HARDSIGMOID EXP GELU_ERF NORM IM2COL POOL_2D SQR SQRT LOG SIN COS CLAMP LEAKY_RELU CONCAT REPEAT PAD PAD_REFLECT_1D ARANGE TIMESTEP_EMBEDDING SUM_ROWS CONV_TRANSPOSE_1D"
rc=0
for op in $OPS; do
  timeout ${TBO_TIMEOUT:-2400} ./oracle/_ref/test-backend-ops test -b MI355X0 -o $op > gpurun_out/tbo_$op.log 2>&1
  r=$?
  echo "== $op rc=$r: $(grep -E 'tests passed' gpurun_out/tbo_$op.log | tail -1)  fails: $(grep -c 'FAIL' gpurun_out/tbo_$op.log)"
  grep -E "FAIL" gpurun_out/tbo_$op.log | sed -E 's/\x1b\[[0-9;]*m//g' | head -6
  [ $r -ne 0 ] && rc=1
done
# summary table (what profiles/rNN_test_backend_ops.txt holds): per op, cases OK / FAIL / left to the CPU backend by supports_op
{
  echo "# reference tests/test-backend-ops.cpp (built by oracle/Makefile.ref) against libggml-mi355x.so on MI355X: tools/run_tbo.sh"
  echo "# op | OK | FAIL | not supported (left to the CPU backend by supports_op)"
  for op in $OPS; do
    f=gpurun_out/tbo_$op.log
    ok=$(sed -E 's/\x1b\[[0-9;]*m//g' $f | grep -E "\): " | grep -c "OK$")
    fail=$(sed -E 's/\x1b\[[0-9;]*m//g' $f | grep -E "\): " | grep -c "FAIL")
    ns=$(sed -E 's/\x1b\[[0-9;]*m//g' $f | grep -E "\): " | grep -c "not supported")
    echo "$op | $ok | $fail | $ns"
  done
} > gpurun_out/tbo_summary.txt
exit $rc
