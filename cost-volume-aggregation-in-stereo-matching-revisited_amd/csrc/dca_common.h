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

// Hardware-predicated loads (T8): a raw buffer load whose byte offset lies outside the descriptor's range returns 0,
// so "load if in bounds else 0" costs one v_cndmask on a 32-bit offset -- no branch, no wait.  (A predicated plain
// load makes hipcc wrap every load in its own branch, in the worst case with an s_waitcnt vmcnt(0) behind it.)
// The descriptor base must be wave-uniform (kernel argument / blockIdx arithmetic) and the range < 2 GiB.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define DCA_OOB_OFFSET ((int)0x7ffffff0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dca_rsrc(const void* base, long bytes) {
  // readfirstlane makes the wave-uniformity of the descriptor PROVABLE; otherwise hipcc wraps every buffer op in a
  // waterfall loop (cdna_hip_programming.md T20)
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
// `ok` is an INTEGER 0/1 mask built with `&` (never `&&`: short-circuit evaluation comes back as control flow around
// every load); the select is bitwise so it cannot be turned into a branch either.
__device__ __forceinline__ int dca_pred_off(int byte_off, int ok) {
  const int m = -ok;
  return (byte_off & m) | (DCA_OOB_OFFSET & ~m);
}
__device__ __forceinline__ float dca_bload1(__amdgpu_buffer_rsrc_t r, int byte_off, int ok) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, dca_pred_off(byte_off, ok), 0, 0));
}
// masked store: an out-of-range offset is dropped by the hardware
__device__ __forceinline__ void dca_bstore1(__amdgpu_buffer_rsrc_t r, float v, int byte_off, int ok) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, dca_pred_off(byte_off, ok), 0, 0);
}
// non-temporal variant (gfx950 cache-policy bit 1 = nt): for outputs that are written once and read much later, so that
// they do not displace the halo / weight lines the kernel re-reads from L2
__device__ __forceinline__ void dca_bstore1_nt(__amdgpu_buffer_rsrc_t r, float v, int byte_off, int ok) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, dca_pred_off(byte_off, ok), 0, 2);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 dca_bload2(__amdgpu_buffer_rsrc_t r, int byte_off, int ok) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, dca_pred_off(byte_off, ok), 0, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ void dca_bstore2(__amdgpu_buffer_rsrc_t r, float2 v, int byte_off, int ok) {
  const u32x2 u = {__float_as_uint(v.x), __float_as_uint(v.y)};
  __builtin_amdgcn_raw_buffer_store_b64(u, r, dca_pred_off(byte_off, ok), 0, 0);
}
__device__ __forceinline__ float4 dca_bload4(__amdgpu_buffer_rsrc_t r, int byte_off, int ok) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, dca_pred_off(byte_off, ok), 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ---- operand scales of the f16x2 kernels (conv3d_f16x2.hip, conv3d_wgrad_f16x2.hip, conv3d_wgrad_s2_f16x2.hip) ----------
// Every operand CHANNEL c is scaled by its own power of two 2^exps[c] that brings the channel's max |.| into [2^14, 2^15)
// before the split into two f16 terms (round 3: per channel, not per tensor -- a channel 10^8 below the tensor's maximum
// keeps its 22 bits).  `exps` (C device ints) come from
//  * per-channel maxima: slots[c * DCA_AMAX_CSLOTS + s], s < nslots -- every workgroup of the producing kernel STORES the
//    bit pattern of its own maximum over channel c (a non-negative fp32 number, so unsigned order = float order) into a
//    slot of its own; dca_amax_exps / dca_conv3d_x2_prep_weight take the maximum over the nslots written slots.  Plain
//    stores and loads only, one writer per slot and launch, nothing to zero-initialise:
//      - atomicMax on ONE word serialised thousands of atomics on one address (+45 us on a 150 us BatchNorm pass);
//      - atomicMax spread over 64 words was read WRONGLY inside hipGraph replays (root cause: DESIGN.md section 3,
//        tools/graph_amax_repro.hip);
//  * or a bound the producer computes before it writes (the packed "px2" operand format below).
#define DCA_AMAX_CSLOTS 1024
// a value another kernel wrote shortly before, read with a device-coherent vector load instead of an s_load through the
// scalar data cache
__device__ __forceinline__ float dca_coherent_loadf(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int dca_coherent_loadi(const int* p) {
  return (int)__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exponent e such that a value whose max |.| has the bit pattern `bits` lands in [2^14, 2^15) after * 2^e; 0 for zero.
// Clamped to [-100, 126]: scaling is done with v_ldexp_f32 (exact for any e), fp32 denormals are not resolved.
__device__ __forceinline__ int x2_scale_exp(unsigned bits) {
  const int e = (int)((bits >> 23) & 255);
  int ex = e == 0 ? 0 : 141 - e;
  ex = ex > 126 ? 126 : (ex < -100 ? -100 : ex);
  return ex;
}
__device__ __forceinline__ float x2_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }   // -126 <= e <= 127
// maximum of a non-negative float over the workgroup -> slot (thread 0 writes); every thread must call it
__device__ __forceinline__ void dca_cmax_put(float m, unsigned* slot) {
  m = wave_max(m);
  __shared__ float dca_cmax_red[16];
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();                       // the array may still be read by a previous call
  if ((threadIdx.x & 63) == 0) dca_cmax_red[w] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) m = fmaxf(m, dca_cmax_red[i]);
    *slot = __float_as_uint(m);
  }
}

// ---- packed operand format "px2" -------------------------------------------------------------------------------------
// An fp32 tensor (N, C, D, H, W), C % 8 == 0, as the two f16 terms the f16x2 kernels multiply, written ONCE by its producer
// (BatchNorm apply / backward: bandwidth-bound kernels with vector-ALU slack) instead of being scaled, split and transposed
// by every consumer's staging code:  per sample  [term 2][C/8][D][H][W][8] f16  with  h = f16(x 2^exps[c]), l = f16(x 2^exps[c] - h)
// -- the same 4 bytes per element as fp32.  A halo voxel's 8 channels are one 16-byte word (the MFMA B fragment of the
// convolution as it sits in LDS), a halo row is one contiguous run.
__device__ __forceinline__ long px2_term_bytes(int C, long S) { return (long)C * S * 2; }
