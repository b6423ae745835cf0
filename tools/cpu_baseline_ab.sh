#!/bin/bash
# tools/cpu_baseline_ab.sh -- bench.py's cpu_baseline leg alone, with and without the OpenMP thread placement (run on the GPU box: its host CPUs are what is measured)
cd "$(dirname "$0")/.."
for v in pinned unpinned; do
  if [ $v = unpinned ]; then export MI355X_CPU_BASELINE_NO_PIN=1; fi
  s=$(date +%s.%N)
  python bench.py --cpu-baseline-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], 'tok/s at', d['cores'], 'threads =', d['gb_per_s'], 'GB/s; sweep', d['thread_sweep_tok_s'], d.get('build'), d.get('topology'))"
  e=$(date +%s.%N); echo "  wall $(echo "$e - $s" | bc) s"
done
