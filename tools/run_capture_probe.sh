cd $GRAFT_REPO_ROOT
for m in fresh_default stale_side alias_default stale_default; do
  echo "=== repro3 MODE=$m"; MODE=$m timeout -k 10 120 python tools/repro_capture3.py > gpurun_out/r03_cap3_$m.log 2>&1; echo "rc=$?" >> gpurun_out/r03_cap3_$m.log; tail -5 gpurun_out/r03_cap3_$m.log
done
for m in eager_first_del eager_first; do
  echo "=== repro2 MODE=$m"; MODE=$m timeout -k 10 180 python tools/repro_capture2.py > gpurun_out/r03_cap2_$m.log 2>&1; echo "rc=$?" >> gpurun_out/r03_cap2_$m.log; tail -8 gpurun_out/r03_cap2_$m.log
done
exit 0
