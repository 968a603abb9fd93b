#!/bin/bash
# Same-box A/B of two builds of libgnc_hip.so (box-to-box spread on the pool is +-3-5 %, more than most kernel
# changes are worth, so variants are compared inside ONE gpurun call):
#   gpurun -- 'bash tools/ab_same_box.sh build/libgnc_base.so build/libgnc_new.so'
# Runs the c3 forward bench twice per variant (A B A B) and the training bench once each; leaves A installed.
set -e
A=$1; B=$2; LIB=graphnet_classifier_amd/libgnc_hip.so
mkdir -p gpurun_out
for v in A B A B; do
  cp "${!v}" $LIB
  python bench.py --no-cpu-baseline --preheat-ms 300 > gpurun_out/ab_$v.log 2>&1
  echo "$v fwd   $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$v.log | head -1) $(grep -o '"kernel_ms_per_step": {[^}]*}' gpurun_out/ab_$v.log | cut -c1-400)"
done
for v in A B; do
  cp "${!v}" $LIB
  python bench.py --mode train --steps 5 --warmup 2 --preheat-ms 300 > gpurun_out/abt_$v.log 2>&1
  echo "$v train $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abt_$v.log | head -1)"
done
cp "$A" $LIB
