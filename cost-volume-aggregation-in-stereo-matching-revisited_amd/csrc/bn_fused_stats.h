// BatchNorm batch statistics emitted by the convolution kernels themselves (conv3d_bf16x3.hip, conv1_x3.hip,
// deconv3d_x3.hip: the *_forward_stats entry points), so that the training-mode convbn_3d (models/submodule.py:121-124)
// needs no separate pass over the convolution output.
//
// The MFMA accumulator layout puts output channel cu(r) + 4*half in register r of the lanes of wave half `half`
// (lanes 0-31 / 32-63), one position per lane.  Every (wave, half, r) keeps {K, s, q} in a wave-private LDS slot and n per
// (wave, half):  K = the first value that wave half produced for the channel (a shift taken from the data itself, so a
// channel with |mean| >> std keeps its variance), s = sum of (y - K), q = sum of (y - K)^2, n = number of positions.
// At the end of the kernel the waves of a workgroup are re-centred on wave 0's K in double and summed in wave order; the
// workgroup writes ONE partial {K, n, s, q} (four doubles) per channel: part[(c * nchunk + blockIdx.x) * 4 + ...], nchunk =
// gridDim.x.  dca_bn_finalize_centered (pointwise.hip) re-centres the partials of a channel the same way.  Everything is
// order-fixed (no atomics between waves): bitwise reproducible, and a function of the data alone.
#pragma once
#include "dca_common.h"

#define FS_WAVE_FLOATS 100   // [half][r][K, s, q] = 96 floats, then n[half] (2), padded

// sum over the 32 lanes of a wave half, result in every lane of the half
__device__ __forceinline__ float fs_half_sum(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));   // row_mirror
  x += __shfl_xor(x, 16, 64);                                                                       // the other 16-lane row
  return x;
}

// wave-private slot of (half, r); call with the wave's base pointer (stat_lds + wave * FS_WAVE_FLOATS)
__device__ __forceinline__ float* fs_slot(float* wbase, int half, int r) { return wbase + (half * 16 + r) * 3; }

// the value lane 0 of each wave half holds, in every lane of that half (wave-uniform control flow required)
__device__ __forceinline__ float fs_half_first(float v, int half) {
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  return half ? b : a;
}

// The workgroup's partial of every channel: thread i < 32 handles channel slot (half = i >> 4, r = i & 15) = local channel
// cu(r) + 4 * half.  Call after a __syncthreads() that follows the last slot update.  c_base: first channel of the block,
// c_limit: number of channels that exist (c_base + local < c_limit is written).
__device__ __forceinline__ void fs_flush(const float* stat_lds, int nwaves, int tid, int c_base, int c_limit,
                                         double* part, int nchunk, int chunk) {
  if (tid >= 32) return;
  const int half = tid >> 4, r = tid & 15, c = c_base + (r & 3) + 8 * (r >> 2) + 4 * half;
  const double k0 = (double)stat_lds[(half * 16 + r) * 3];
  double n = 0.0, s = 0.0, q = 0.0;
  for (int w = 0; w < nwaves; ++w) {
    const float* wb = stat_lds + w * FS_WAVE_FLOATS;
    const double kw = (double)wb[(half * 16 + r) * 3], sw = (double)wb[(half * 16 + r) * 3 + 1];
    const double qw = (double)wb[(half * 16 + r) * 3 + 2], nw = (double)wb[96 + half];
    const double dk = kw - k0;
    n += nw;
    s += sw + nw * dk;
    q += qw + 2.0 * dk * sw + nw * dk * dk;
  }
  if (c < c_limit) {
    double* p = part + ((long)c * nchunk + chunk) * 4;
    p[0] = k0; p[1] = n; p[2] = s; p[3] = q;
  }
}
