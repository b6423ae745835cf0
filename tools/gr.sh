#!/bin/bash
# tools/gr.sh TIMEOUT 'command' -- gpurun with retries while every GPU slot of the pod is busy (exit 3 / "transient": nothing charged)
T=$1; shift
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$T" -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit $rc
done
echo "$out"; exit 3
