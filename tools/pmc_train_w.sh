#!/bin/bash
# SQ instruction / wait counters of the training step's kernels for one workload: bash tools/pmc_train_w.sh c5 r02_r
# (one --pmc pass with --kernel-trace only; program directly after `--`).  Output: gpurun_out/<tag>_pmc_train_<W>.json
set -o pipefail
W=${1:-c3}; TAG=${2:-r02}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d gpurun_out/${TAG}_pmc_train_$W -o run -- python3 bench.py --workload $W --mode train --steps 2 --warmup 1 --preheat-ms 0 --no-cpu-baseline \
  > gpurun_out/${TAG}_pmc_train_$W.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_train_$W.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_train_$W.json --workload $W-train gpurun_out/${TAG}_pmc_train_$W
rm -rf gpurun_out/${TAG}_pmc_train_$W
python3 - <<P
import json
d = json.load(open('gpurun_out/${TAG}_pmc_train_$W.json'))
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['avg_duration_us_under_pmc'] * kv[1]['dispatches_per_pass']):
    if v['avg_duration_us_under_pmc'] * v['dispatches_per_pass'] < 2000:
        continue
    c = v['counters_avg_per_dispatch']
    m = max(c.get('SQ_INSTS_MFMA', 1), 1)
    print(k[:70], v['dispatches_per_pass'], round(v['avg_duration_us_under_pmc']), {x[8:]: round(c[x] / m, 2) for x in c if x.startswith('SQ_INSTS')},
          'wait_any', round(v.get('wait_any_frac', 0), 3), 'wait_inst', round(v.get('wait_inst_any_frac', 0), 3))
P
