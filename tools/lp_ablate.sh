#!/bin/bash
# Ablation of conv3_lp_kernel on the GPU box: rebuilds conv3d_lp.hip with the given -D flags (e.g. -DLP_ABL=3) and times the BASELINE-shape launch.
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/conv3d_lp.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  echo "$v: $(DCA_LP_ONLY=1 python tools/lp_time.py | grep 'bfloat16  lp ->lp')"
done
