#!/bin/bash
# Same-box A/B of builds on the forward step of several workloads:  tools/ab_fwd.sh "c3 c2 c5" libA.so libB.so
WL=$1; shift
for round in 1 2; do
  for W in $WL; do
    for v in "$@"; do
      GNC_LIB_PATH=$v python bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$(basename $v)', round(d['ms_per_step'],3), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items() if x>0.5})"
    done
  done
done
