#!/bin/bash
# builds the X2_STAMP debug library (tools/bin/libdca_stamp.so: the product sources with -DX2_STAMP=1), here in the container
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
SRC=cost-volume-aggregation-in-stereo-matching-revisited_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DX2_STAMP=1 -shared -o tools/bin/libdca_stamp.so $SRC/conv3d_f16x2.hip
echo built tools/bin/libdca_stamp.so
