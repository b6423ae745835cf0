#!/bin/bash
# Run the reference's own op-parity harness (oracle/_ref/test-backend-ops, built by oracle/Makefile.ref from
# /root/reference/tests/test-backend-ops.cpp) against libggml-mi355x.so on the GPU box.
# usage: tools/run_tbo.sh [op ...]   (default: every op the backend claims)
cd "$(dirname "$0")/.."
export GGML_BACKEND_PATH="$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so"
export LD_LIBRARY_PATH="$PWD/oracle/_ref:$LD_LIBRARY_PATH"
mkdir -p gpurun_out
OPS="$@"
[ -z "$OPS" ] && OPS="MUL_MAT ADD SUB MUL DIV RMS_NORM SCALE UNARY GLU ROPE SOFT_MAX CPY CONT DUP GET_ROWS SET_ROWS FLASH_ATTN_EXT"
rc=0
for op in $OPS; do
  timeout 600 ./oracle/_ref/test-backend-ops test -b MI355X0 -o $op > gpurun_out/tbo_$op.log 2>&1
  r=$?
  echo "== $op rc=$r: $(grep -E 'tests passed' gpurun_out/tbo_$op.log | tail -1)  fails: $(grep -c 'FAIL' gpurun_out/tbo_$op.log)"
  grep -E "FAIL" gpurun_out/tbo_$op.log | sed -E 's/\x1b\[[0-9;]*m//g' | head -6
  [ $r -ne 0 ] && rc=1
done
exit $rc
