#!/bin/bash
# A/B of wgrad3s2_f16x2_kernel builds on the GPU box
cd "$(dirname "$0")/.."
P=cost-volume-aggregation-in-stereo-matching-revisited_amd
for v in "$@"; do
  touch $P/csrc/conv3d_wgrad_s2_f16x2.hip
  DCA_EXTRA_CFLAGS="$v" python $P/_build.py > /dev/null 2>&1 || { echo "$v: build failed"; continue; }
  echo "== $v"; timeout -k 5 200 python tools/ws2_time.py 2>&1 | grep -E "^N=|f16x2 max"
done
touch $P/csrc/conv3d_wgrad_s2_f16x2.hip; python $P/_build.py > /dev/null 2>&1
