#!/bin/bash
# A/B of the training step with / without non-temporal conv output stores (does the following BatchNorm pass read y from the infinity cache?)
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/conv3d_bf16x3.hip cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/deconv3d_x3.hip cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/volume_fused.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  python bench.py --no-cpu-baseline $NT_BENCH_ARGS > gpurun_out/nt_ab.log 2>&1
  echo "$v: $(python - <<PY
import json
l=[x for x in open("gpurun_out/nt_ab.log") if x.startswith("{")][-1]; d=json.loads(l); print(d["ms_per_step"])
PY
)"
done
