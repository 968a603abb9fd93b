#!/bin/bash
# Same-box A/B/... of several builds of libgnc_hip.so on the c3 TRAINING step (see tools/ab_same_box.sh for why):
#   gpurun -- 'bash tools/ab_train.sh build/libgnc_A.so build/libgnc_B.so [...]'
# Two rounds over all variants; leaves the first one installed.
LIB=graphnet_classifier_amd/libgnc_hip.so
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    cp "$v" $LIB
    python bench.py --mode train --steps 6 --warmup 2 --preheat-ms 300 --no-cpu-baseline > gpurun_out/abt_$(basename $v).log 2>&1
    echo "$(basename $v) train $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abt_$(basename $v).log | head -1) $(grep -o '"mlp_backward_fused[^,]*' gpurun_out/abt_$(basename $v).log | tr '\n' ' ')"
  done
done
cp "$1" $LIB
