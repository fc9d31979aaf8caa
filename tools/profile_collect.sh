#!/bin/bash
# rocprofv3 kernel-trace passes of bench.py for one profiles/ set (run on the GPU box from the repo root):
#     bash tools/profile_collect.sh r02_c
# writes gpurun_out/<set>_{fwdbwd_b1,fwd_f32,fwd_bf16,fwd_fp16}_{kernel_stats.csv,step_breakdown.txt}
set -e
SET=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${SET}_$name -- python bench.py "$@" --no-cpu-baseline > gpurun_out/${SET}_${name}.log 2>&1
  cp "$(ls -t gpurun_out/prof_${SET}_$name/*/*_kernel_stats.csv | head -1)" gpurun_out/${SET}_${name}_kernel_stats.csv
  python tools/step_breakdown.py "$(ls -t gpurun_out/prof_${SET}_$name/*/*_kernel_trace.csv | head -1)" 40 > gpurun_out/${SET}_${name}_step_breakdown.txt
  rm -rf gpurun_out/prof_${SET}_$name
  echo "$name done"
}
run fwdbwd_b1 --batch 1 --steps 8 --warmup 3
run fwd_f32 --mode fwd --steps 20 --warmup 5
run fwd_bf16 --mode fwd --dtype bf16 --steps 20 --warmup 5
run fwd_fp16 --mode fwd --dtype fp16 --steps 20 --warmup 5
