#!/bin/bash
# A/B of wgrad3_bf16x3_kernel builds on the GPU box: rebuilds conv3d_wgrad_bf16x3.hip with the given -D flags and times the 32->32 weight gradient at 48x136x240.
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/conv3d_wgrad_bf16x3.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  echo "$v: $(timeout -k 5 120 python tools/wx3_time.py | tr '\n' '|')"
done
