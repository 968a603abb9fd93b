#!/bin/bash
# Same-box A/B of the INFERENCE forward of two library builds, all three single-GPU workloads (A B A B per workload):
#   gpurun -- 'bash tools/ab_fwd_libs.sh graphnet_classifier_amd/libgnc_hip.so build/libgnc_ab.so "c3 c2 c5"'
A=$1; B=$2; WL=${3:-"c3 c2 c5"}
mkdir -p gpurun_out
for W in $WL; do
  for v in A B A B; do
    GNC_LIB_PATH="${!v}" python bench.py --workload $W --steps 20 --warmup 3 --preheat-ms 300 --no-cpu-baseline --train-steps 0 > gpurun_out/abf_${W}_$v.log 2>&1 || { tail -3 gpurun_out/abf_${W}_$v.log; exit 1; }
    echo "$W $v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abf_${W}_$v.log | head -1)"
  done
done
