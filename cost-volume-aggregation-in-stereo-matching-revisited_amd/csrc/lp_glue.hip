// Bandwidth-bound glue of the reduced-precision inference path, at the boundary between the 2-byte 1/4-resolution
// tensors and the fp32 1/8-resolution interior of a DCA block:
//   avgpool3d_lp   nn.AvgPool3d((3,3,3), stride 2, padding 1) (models/augment/cva.py:39), 2-byte in -> fp32 out
//   trilinear_lp   F.interpolate(scale_factor=(2,2,2), mode='trilinear') (cva.py:64), fp32 in -> 2-byte out
// Same arithmetic as the fp32 kernels of pointwise.hip (count_include_pad=True: always / 27; align_corners=False weights
// .25 / .75 with clamped ends); only the storage type of the large side differs.
#include "dca_common.h"
#include "../../include/dca_hip.h"

namespace {

template <typename MT> __device__ __forceinline__ float g_lo(unsigned w) {
  return (float)__builtin_bit_cast(MT, (unsigned short)(w & 0xffffu));
}
template <typename MT> __device__ __forceinline__ float g_hi(unsigned w) {
  return (float)__builtin_bit_cast(MT, (unsigned short)(w >> 16));
}
template <typename MT> __device__ __forceinline__ unsigned g_pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef MT mtx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mtx2));
}

// One thread per PAIR of outputs (ow0 = 2p, 2p + 1): per (kd, kh) row it needs the inputs w = 4p-1 .. 4p+3 = one
// aligned 8-byte quad + the 2-byte sample left of it (hardware-predicated buffer loads: out-of-range rows / columns
// read as the zero padding).  Wi % 4 == 0.  grid (ceil(Wo/2 / 256), Ho, NC * Do).
template <typename MT>
__global__ __launch_bounds__(256) void avgpool3d_lp_kernel(const unsigned short* __restrict__ x, float* __restrict__ y,
                                                           int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  const int p = blockIdx.x * 64 + (threadIdx.x & 63), oh = blockIdx.y * 4 + (threadIdx.x >> 6);   // a wave = 64 pairs of one row
  const int od = blockIdx.z % Do;
  const long nc = blockIdx.z / Do;
  if (2 * p >= Wo || oh >= Ho) return;
  const long plane = (long)Di * Hi * Wi;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(x + nc * plane, plane * 2);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd) {
    const int d = 2 * od - 1 + kd;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * oh - 1 + kh;
      const int okr = (int)((unsigned)d < (unsigned)Di) & (int)((unsigned)h < (unsigned)Hi);
      const int row = (d * Hi + h) * Wi;
      const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(xr, dca_pred_off((row + 4 * p) * 2, okr), 0, 0);
      const unsigned e = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, dca_pred_off((row + 4 * p - 1) * 2, okr & (int)(p > 0)), 0, 0);
      const float a0 = g_lo<MT>(q.x), a1 = g_hi<MT>(q.x), a2 = g_lo<MT>(q.y), a3 = g_hi<MT>(q.y);
      s0 += g_lo<MT>(e) + a0 + a1;
      s1 += a1 + a2 + a3;
    }
  }
  float* o = y + ((nc * Do + od) * Ho + oh) * (long)Wo + 2 * p;
  *(float2*)o = make_float2(s0 * (1.0f / 27.0f), s1 * (1.0f / 27.0f));
}

// x2 up-sampling: one thread per coarse cell writes its 2x2x2 outputs, a packed pair of 2-byte values per fine row.
template <typename MT>
__global__ __launch_bounds__(256) void trilinear_up2_lp_kernel(const float* __restrict__ x, unsigned* __restrict__ y,
                                                               int Di, int Hi, int Wi) {
  const int mw = blockIdx.x * 64 + (threadIdx.x & 63), mh = blockIdx.y * 4 + (threadIdx.x >> 6);   // a wave = 64 w of one row
  const int md = blockIdx.z % Di;
  const long nc = blockIdx.z / Di;
  if (mw >= Wi || mh >= Hi) return;
  const float* p = x + nc * Di * Hi * Wi;
  const int dm = max(md - 1, 0), dp = min(md + 1, Di - 1), hm = max(mh - 1, 0), hp = min(mh + 1, Hi - 1);
  const int wm = max(mw - 1, 0), wp = min(mw + 1, Wi - 1);
  float r0[3][3], r1[3][3];
  const int ds[3] = {dm, md, dp}, hs[3] = {hm, mh, hp};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float* row = p + ((long)ds[a] * Hi + hs[b]) * Wi;
      const float xm = row[wm], x0 = row[mw], xp = row[wp];
      r0[a][b] = 0.25f * xm + 0.75f * x0;
      r1[a][b] = 0.75f * x0 + 0.25f * xp;
    }
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  unsigned* q = y + nc * (2L * Di) * Ho * Wi;     // y as dwords: a fine row of Wo 2-byte values = Wi dwords
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int a0 = pd ? 1 : 0, a1 = pd ? 2 : 1, b0 = ph ? 1 : 0, b1 = ph ? 2 : 1;
      const float wa0 = pd ? 0.75f : 0.25f, wa1 = pd ? 0.25f : 0.75f, wb0 = ph ? 0.75f : 0.25f, wb1 = ph ? 0.25f : 0.75f;
      const float ox = wa0 * (wb0 * r0[a0][b0] + wb1 * r0[a0][b1]) + wa1 * (wb0 * r0[a1][b0] + wb1 * r0[a1][b1]);
      const float oy = wa0 * (wb0 * r1[a0][b0] + wb1 * r1[a0][b1]) + wa1 * (wb0 * r1[a1][b0] + wb1 * r1[a1][b1]);
      q[((long)(2 * md + pd) * Ho + 2 * mh + ph) * Wi + mw] = g_pack2<MT>(ox, oy);
    }
}

}  // namespace

extern "C" int dca_avgpool3d_lp_fwd(const void* x, float* y, long NC, int Di, int Hi, int Wi, int dtype,
                                    hipStream_t stream) {
  DCA_REQUIRE(x && y && NC > 0 && Di > 0 && Hi > 0 && Wi > 0 && (dtype == DCA_BF16 || dtype == DCA_FP16));
  DCA_REQUIRE(Wi % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 7) == 0);
  const int Do = (Di + 1) / 2, Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
  DCA_REQUIRE(Ho <= 65535 && NC * Do <= 65535 && (long)Di * Hi * Wi * 2 < 0x7ffffff0L);
  const dim3 grid(cdiv(Wo / 2, 64), cdiv(Ho, 4), (unsigned)(NC * Do));
  if (dtype == DCA_BF16)
    hipLaunchKernelGGL(avgpool3d_lp_kernel<__bf16>, grid, dim3(256), 0, stream, (const unsigned short*)x, y, Di, Hi, Wi,
                       Do, Ho, Wo);
  else
    hipLaunchKernelGGL(avgpool3d_lp_kernel<_Float16>, grid, dim3(256), 0, stream, (const unsigned short*)x, y, Di, Hi, Wi,
                       Do, Ho, Wo);
  return dca_launch_status();
}

extern "C" int dca_trilinear_up2_lp_fwd(const float* x, void* y, long NC, int Di, int Hi, int Wi, int dtype,
                                        hipStream_t stream) {
  DCA_REQUIRE(x && y && NC > 0 && Di > 0 && Hi > 0 && Wi > 0 && (dtype == DCA_BF16 || dtype == DCA_FP16));
  DCA_REQUIRE(Hi <= 65535 && NC * Di <= 65535 && (((uintptr_t)y) & 3) == 0);
  const dim3 grid(cdiv(Wi, 64), cdiv(Hi, 4), (unsigned)(NC * Di));
  if (dtype == DCA_BF16)
    hipLaunchKernelGGL(trilinear_up2_lp_kernel<__bf16>, grid, dim3(256), 0, stream, x, (unsigned*)y, Di, Hi, Wi);
  else
    hipLaunchKernelGGL(trilinear_up2_lp_kernel<_Float16>, grid, dim3(256), 0, stream, x, (unsigned*)y, Di, Hi, Wi);
  return dca_launch_status();
}
