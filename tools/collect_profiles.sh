#!/bin/bash
# Collects, on the GPU box, everything profiles/ keeps for one round tag:
#   tools/collect_profiles.sh r02_b "c3 c2 c5"
# per workload: the bench line, the rocprofv3 kernel-trace stats of the same command, and three --pmc passes
# (SQ set, FETCH_SIZE, WRITE_SIZE: separate passes, never together with trace domains other than --kernel-trace)
# condensed by tools/pmc_summary.py.  The profiled program comes directly after `--` (no env/bash hop).
# Output: gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r02}
WORKLOADS=${2:-"c3 c2 c5"}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for W in $WORKLOADS; do
  echo "== $W bench" >&2
  timeout -k 10 400 python3 bench.py --workload $W > $OUT/${TAG}_bench_$W.jsonl 2> $OUT/${TAG}_bench_$W.err || exit 1
  echo "== $W kernel stats" >&2
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$W -o run -- \
    python3 bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof_$W.log 2>&1 || exit 1
  cp "$(find $OUT/${TAG}_prof_$W -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_${W}_kernel_stats.csv
  for P in sq fetch write; do
    case $P in sq) C="$SQ";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
    echo "== $W pmc $P" >&2
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_${W}_$P -o run -- \
      python3 bench.py --workload $W --steps 2 --warmup 1 --preheat-ms 0 --no-cpu-baseline --train-steps 0 \
      > $OUT/${TAG}_pmc_${W}_$P.log 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py $OUT/${TAG}_pmc_$W.json --workload $W $OUT/${TAG}_pmc_${W}_sq $OUT/${TAG}_pmc_${W}_fetch $OUT/${TAG}_pmc_${W}_write || exit 1
  rm -rf $OUT/${TAG}_pmc_${W}_sq $OUT/${TAG}_pmc_${W}_fetch $OUT/${TAG}_pmc_${W}_write $OUT/${TAG}_prof_$W
done
echo done >&2
# training legs: bench line + kernel stats of the training step (set TRAIN="c3 c5 c2" to collect)
for W in ${TRAIN:-}; do
  echo "== $W train" >&2
  timeout -k 10 400 python3 bench.py --workload $W --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_train_$W.jsonl 2> $OUT/${TAG}_bench_train_$W.err || exit 1
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train_$W -o run -- \
    python3 bench.py --workload $W --mode train --steps 3 --warmup 1 --preheat-ms 0 --no-cpu-baseline > $OUT/${TAG}_prof_train_$W.log 2>&1 || exit 1
  cp "$(find $OUT/${TAG}_prof_train_$W -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_train_${W}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_train_$W
done
echo done-train >&2
