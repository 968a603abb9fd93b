#!/bin/bash
# same-box A/B of one environment switch: tools/ab_env.sh <tag> <ENVVAR> "<bench args>"  (B = the switch set to 1)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=$1; VAR=$2; shift 2
for arm in A B A B; do
  if [ $arm = B ]; then export $VAR=1; else unset $VAR; fi
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$arm', '$VAR=' + ('1' if '$arm' == 'B' else '-'), 'ms_per_step', round(d['ms_per_step'], 3), 'train', round(d.get('train', {}).get('ms_per_step', 0), 2), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})" | tee -a gpurun_out/${TAG}.log
done
