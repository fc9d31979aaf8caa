#!/bin/bash
# A/B of conv3_bf16x3_kernel builds on the GPU box: rebuilds conv3d_bf16x3.hip with the given -D flags and times the network's shapes.
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/conv3d_bf16x3.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  echo "$v: $(timeout -k 5 120 python tools/x3_vs_fp32.py | awk '{print $3, $5}' | head -4 | tr '\n' '|')"
done
