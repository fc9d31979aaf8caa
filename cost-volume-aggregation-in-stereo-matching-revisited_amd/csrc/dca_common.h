// Shared device/host helpers for the DCANet hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DCA_WAVE 64

// Every C-ABI entry point returns a hipError_t cast to int (0 = success); invalid arguments are
// reported as hipErrorInvalidValue before anything is launched.
#define DCA_REQUIRE(cond)                      \
  do {                                         \
    if (!(cond)) return (int)hipErrorInvalidValue; \
  } while (0)

static inline int dca_launch_status() { return (int)hipGetLastError(); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Blocks are dealt round-robin over the 8 XCDs (private L2 each): give every XCD a contiguous
// chunk of the logical tile space so neighbouring tiles (which share halos) hit the same L2.
// Bijective for any grid size (cdna_hip_programming.md, T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Activation codes used by every epilogue / BN-apply kernel: slope 1 = identity, 0 = ReLU,
// 0.1 = LeakyReLU(0.1).
__device__ __forceinline__ float act_apply(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
