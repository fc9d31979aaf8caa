#!/bin/bash
# A/B of conv3_f16x2_kernel builds on the GPU box: rebuilds conv3d_f16x2.hip with the given -D flags, times the network's shapes
cd "$(dirname "$0")/.."
P=cost-volume-aggregation-in-stereo-matching-revisited_amd
for v in "$@"; do
  touch $P/csrc/conv3d_f16x2.hip $P/csrc/conv3d_wgrad_f16x2.hip
  DCA_EXTRA_CFLAGS="$v" python $P/_build.py > /dev/null 2>&1 || { echo "$v: build failed"; continue; }
  echo "== $v"; timeout -k 5 200 python tools/x2_check.py 2>&1 | grep -E "^N=|dw [0-9.e-]+ \| x3" | sed 's/| x3.*//'
done
touch $P/csrc/conv3d_f16x2.hip $P/csrc/conv3d_wgrad_f16x2.hip; python $P/_build.py > /dev/null 2>&1
