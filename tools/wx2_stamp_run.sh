#!/bin/bash
# stamp build of conv3d_wgrad_f16x2.hip on the GPU box, tile timeline, then restore the normal build
cd "$(dirname "$0")/.."
P=cost-volume-aggregation-in-stereo-matching-revisited_amd
touch $P/csrc/conv3d_wgrad_f16x2.hip
DCA_EXTRA_CFLAGS="-DWX2_STAMP=1 $1" python $P/_build.py > /dev/null 2>&1 || exit 1
timeout -k 5 120 python tools/wx2_stamps.py
touch $P/csrc/conv3d_wgrad_f16x2.hip
python $P/_build.py > /dev/null 2>&1
