#!/bin/bash
# sample clocks / power while a long GEMM loop runs
cd /root/repo
(python tools/gemm_one.py 8192 8192 8192 400 > /tmp/g.log 2>&1) &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i -E "sclk|power|busy|mclk|fclk" | tr '\n' ' '; echo; sleep 1.5; done
wait $PID
cat /tmp/g.log | grep TFLOP
rocm-smi --showmaxpower 2>/dev/null | grep -i power
