set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_d_train_c3.jsonl 2> gpurun_out/r02_d_train_c3.err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r02_d_train_c3.jsonl').read().strip().splitlines()[-1]); print('train ms', d['ms_per_step']); print({k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r02_d_pmc_train -o run -- python3 bench.py --mode train --steps 2 --warmup 1 --preheat-ms 0 --no-cpu-baseline > gpurun_out/r02_d_pmc_train.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/r02_d_pmc_train.json --workload c3-train gpurun_out/r02_d_pmc_train
rm -rf gpurun_out/r02_d_pmc_train
python -c "
import json; d=json.load(open('gpurun_out/r02_d_pmc_train.json'))
for k,v in d['kernels'].items():
    if 'backward' in k or 'xty' in k:
        c=v['counters_avg_per_dispatch']; print(k[:50], v['dispatches_per_pass'], round(v['avg_duration_us_under_pmc']), {x:round(c[x]/max(c.get('SQ_INSTS_MFMA',1),1),2) for x in c if x.startswith('SQ_INSTS')}, round(v.get('wait_any_frac',0),3), round(v.get('wait_inst_any_frac',0),3))"
