#!/bin/bash
# rocprofv3 kernel trace of the default bench (per-GPU batch 4) -> gpurun_out/<set>_fwdbwd_b4_{kernel_stats.csv,step_breakdown.txt}
set -e
SET=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${SET}_b4 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${SET}_fwdbwd_b4.log 2>&1
cp "$(ls -t gpurun_out/prof_${SET}_b4/*/*_kernel_stats.csv | head -1)" gpurun_out/${SET}_fwdbwd_b4_kernel_stats.csv
python tools/step_breakdown.py "$(ls -t gpurun_out/prof_${SET}_b4/*/*_kernel_trace.csv | head -1)" 60 --by-grid > gpurun_out/${SET}_fwdbwd_b4_step_breakdown.txt
rm -rf gpurun_out/prof_${SET}_b4
