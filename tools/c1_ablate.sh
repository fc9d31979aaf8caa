#!/bin/bash
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/conv1_lp.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  echo "$v: $(python tools/lp_time.py | grep 'conv1 lp bfloat16' | tr '\n' '|')"
done
