#!/bin/bash
# Ablation of deconv3_bf16x3_kernel on the GPU box: rebuilds deconv3d_x3.hip with the given -D flags (e.g. -DDX3_ABL=3) and times the 64->32 launch.
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/deconv3d_x3.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  echo "$v: $(timeout -k 5 90 python tools/deconv_time.py | grep 'bf16x3' | sed 's/(.*fp32-equivalent), //' | tr '\n' '|')"
done
