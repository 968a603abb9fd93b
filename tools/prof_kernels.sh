#!/bin/bash
# rocprofv3 kernel-trace stats of one python tool: tools/prof_kernels.sh <tag> <script.py> [args...]  (output: gpurun_out/<tag>_kernel_stats.csv)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o run -- python3 "$@" > gpurun_out/${TAG}_prof.log 2>&1 || { tail -5 gpurun_out/${TAG}_prof.log; exit 1; }
cp "$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv
cut -d, -f1-4 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160 | head -${LINES_SHOWN:-25}
