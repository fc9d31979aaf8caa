#!/bin/bash
cd "$(dirname "$0")/.."
for v in "$@"; do
  touch cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc/conv3d_mfma.hip
  DCA_EXTRA_CFLAGS="$v" python cost-volume-aggregation-in-stereo-matching-revisited_amd/_build.py > /dev/null 2>&1 || exit 1
  echo "$v: $(python tools/conv1_time.py | tr '\n' '|')"
done
