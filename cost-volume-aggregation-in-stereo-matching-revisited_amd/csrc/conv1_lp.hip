// Reduced-precision 1x1x1 convolution (pointwise GEMM) of the inference path: bf16 / fp16 activations, one native MFMA
// product per multiply, fp32 accumulation and fp32 epilogue (folded BatchNorm affine, activation, residuals).
//
// Reference operators served (eval mode): the `fuse` conv on cat([aug, cost_volume], 1) -- two inputs, no concat
// (models/augment/cva.py:55,69), `cost_agg.redir` (cva.py:23), and the tap-expansion GEMM of the 32 -> 1 logit heads
// (fp32 output; see conv3d_c1.hip).
//
// Bandwidth bound by construction.  A wave owns 128 consecutive voxels; lane (r = lane & 31, h = lane >> 5) loads, for
// each of the 8 channels 8h..8h+7 of a 16-channel chunk, the four voxels 4r..4r+3 as one 8-byte load (32 lanes = 256
// contiguous bytes per channel).  MFMA column tile j is the voxel set {4r + j}: the lane's B fragment for tile j is the
// 8 channels of its own voxel 4r + j, assembled from the loaded words with four v_perm -- no LDS.  The results of the
// four tiles give, per output channel, the lane's four voxels again: one 8-byte (or 16-byte fp32) store per channel.
#include "dca_common.h"
#include "../../include/dca_hip.h"

typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c1_f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct C1LpArgs {
  const void* x;       // (N, C1, S)  2-byte
  const void* x2;      // (N, C2, S)  2-byte or null
  const unsigned short* wfrag;   // [chunk][lane][8] A fragments (2-byte), chunk = 16 input channels
  void* y;             // (N, Cout, S) 2-byte or fp32
  const float* scale;
  const float* shift;
  const void* res_pre;
  const void* res_post;
  float slope;
  int N, C1, C2, Cout, NCH1, NCH2;
  long S;
};

template <typename MT> struct C1;
template <> struct C1<__bf16> {
  typedef c1_bf16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct C1<_Float16> {
  typedef c1_f16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <typename MT> __device__ __forceinline__ unsigned c1_pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef MT mtx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mtx2));
}
template <typename MT> __device__ __forceinline__ float c1_lo(unsigned w) {
  return (float)__builtin_bit_cast(MT, (unsigned short)(w & 0xffffu));
}
template <typename MT> __device__ __forceinline__ float c1_hi(unsigned w) {
  return (float)__builtin_bit_cast(MT, (unsigned short)(w >> 16));
}

constexpr int MAXCH = 8;   // 16-channel chunks over both inputs (Cin <= 128)

// S % 4 == 0 and 8-byte aligned bases (the caller checks); OUT32: fp32 output and residuals
// two workgroups per CU (<= 256 registers per lane): 59.6 -> 41.9 us (32->32) and 66.5 -> 53.4 us (64->32, two inputs) at
// 48x136x240 -- 4.8 / 5.6 TB/s; three or four per CU make hipcc spill (82 / 161 us)
#ifndef C1_OCC
#define C1_OCC 2
#endif
template <typename MT, bool OUT32>
__global__ __launch_bounds__(256, C1_OCC) void conv1_lp_kernel(C1LpArgs a) {
  typedef typename C1<MT>::vec8 vec8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int nch = a.NCH1 + a.NCH2;
  vec8 wf[MAXCH];
#pragma unroll
  for (int c = 0; c < MAXCH; ++c)
    if (c < nch) wf[c] = *(const vec8*)(a.wfrag + ((long)c * 64 + lane) * 8);
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  float sc[16], sh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = min((r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
    sc[r] = has_aff ? a.scale[co] : 1.f;
    sh[r] = has_aff ? a.shift[co] : 0.f;
  }
  constexpr int OSZ = OUT32 ? 4 : 2;
  const long ngroups = (a.S + 127) / 128, total = (long)a.N * ngroups;
  for (long g = (long)blockIdx.x * 4 + wv; g < total; g += (long)gridDim.x * 4) {
    const int n = (int)(g / ngroups);
    const long v = (g - (long)n * ngroups) * 128 + 4 * l31;          // this lane's first voxel
    const int inr = (int)(v < a.S);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
      if (c < nch) {
        const bool first = c < a.NCH1;
        const int cc = first ? c : c - a.NCH1, C = first ? a.C1 : a.C2;
        const __amdgpu_buffer_rsrc_t xr = dca_rsrc((const char*)(first ? a.x : a.x2) + (long)n * C * a.S * 2, (long)C * a.S * 2);
        // channel >= C lands beyond the descriptor's range -> zero (partial last chunk)
        const int base = dca_pred_off((int)((((long)(cc * 16 + 8 * half)) * a.S + v) * 2), inr);
        u32x2 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = __builtin_amdgcn_raw_buffer_load_b64(xr, base + (int)(j * a.S * 2), 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned sel = (t & 1) ? 0x07060302u : 0x05040100u;
          u32x4 f;
          f.x = __builtin_amdgcn_perm(t < 2 ? q[1].x : q[1].y, t < 2 ? q[0].x : q[0].y, sel);
          f.y = __builtin_amdgcn_perm(t < 2 ? q[3].x : q[3].y, t < 2 ? q[2].x : q[2].y, sel);
          f.z = __builtin_amdgcn_perm(t < 2 ? q[5].x : q[5].y, t < 2 ? q[4].x : q[4].y, sel);
          f.w = __builtin_amdgcn_perm(t < 2 ? q[7].x : q[7].y, t < 2 ? q[6].x : q[6].y, sel);
          acc[t] = C1<MT>::mfma(wf[c], __builtin_bit_cast(vec8, f), acc[t]);
        }
      }
    }
    // epilogue: one predicated base per lane; an output channel >= Cout is beyond the descriptor's range
    const long osample = (long)a.Cout * a.S;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc((char*)a.y + (long)n * osample * OSZ, osample * OSZ);
    const __amdgpu_buffer_rsrc_t pr = dca_rsrc((const char*)(has_pre ? a.res_pre : a.y) + (long)n * osample * OSZ, osample * OSZ);
    const __amdgpu_buffer_rsrc_t qr = dca_rsrc((const char*)(has_post ? a.res_post : a.y) + (long)n * osample * OSZ, osample * OSZ);
    const int obase = dca_pred_off((int)(((long)(4 * half) * a.S + v) * OSZ), inr);
#pragma unroll
    for (int rc = 0; rc < 16; rc += 4) {
      float rp[4][4], rq[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) rp[q][j] = rq[q][j] = 0.f;
      if (has_pre) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = rc + q, off = obase + (int)(((r & 3) + 8 * (r >> 2)) * a.S * OSZ);
          if constexpr (OUT32) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(pr, off, 0, 0);
            rp[q][0] = __uint_as_float(w.x); rp[q][1] = __uint_as_float(w.y); rp[q][2] = __uint_as_float(w.z); rp[q][3] = __uint_as_float(w.w);
          } else {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(pr, off, 0, 0);
            rp[q][0] = c1_lo<MT>(w.x); rp[q][1] = c1_hi<MT>(w.x); rp[q][2] = c1_lo<MT>(w.y); rp[q][3] = c1_hi<MT>(w.y);
          }
        }
      }
      if (has_post) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = rc + q, off = obase + (int)(((r & 3) + 8 * (r >> 2)) * a.S * OSZ);
          if constexpr (OUT32) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(qr, off, 0, 0);
            rq[q][0] = __uint_as_float(w.x); rq[q][1] = __uint_as_float(w.y); rq[q][2] = __uint_as_float(w.z); rq[q][3] = __uint_as_float(w.w);
          } else {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(qr, off, 0, 0);
            rq[q][0] = c1_lo<MT>(w.x); rq[q][1] = c1_hi<MT>(w.x); rq[q][2] = c1_lo<MT>(w.y); rq[q][3] = c1_hi<MT>(w.y);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rc + q, off = obase + (int)(((r & 3) + 8 * (r >> 2)) * a.S * OSZ);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = act_apply(acc[j][r] * sc[r] + sh[r] + rp[q][j], a.slope) + rq[q][j];
        if constexpr (OUT32) {
          const u32x4 w = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
          __builtin_amdgcn_raw_buffer_store_b128(w, yr, off, 0, 0);
        } else {
          const u32x2 w = {c1_pack2<MT>(o[0], o[1]), c1_pack2<MT>(o[2], o[3])};
          __builtin_amdgcn_raw_buffer_store_b64(w, yr, off, 0, 0);
        }
      }
    }
  }
}

// wfrag[chunk][lane][j] = W[cout = lane & 31][cin = chunk_base + 8 * (lane >> 5) + j] as MT; the chunks of the first
// input (C1 channels, padded to 16) come first, then those of the second input.  w: (Cout, C1 + C2) row major fp32.
template <typename MT>
__global__ void conv1_lp_prep_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout, int C1,
                                     int C2, int NCH1, int NCH2) {
  const int total = (NCH1 + NCH2) * 512;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, c = idx >> 9;
    const bool first = c < NCH1;
    const int ci = (first ? c : c - NCH1) * 16 + 8 * (lane >> 5) + j, co = lane & 31;
    float v = 0.f;
    if (co < Cout && ci < (first ? C1 : C2)) v = w[(long)co * (C1 + C2) + (first ? ci : C1 + ci)];
    const MT m = (MT)v;
    dst[idx] = __builtin_bit_cast(unsigned short, m);
  }
}

}  // namespace

extern "C" long dca_conv1_lp_weight_bytes(int C1, int C2) {
  if (C1 <= 0 || C2 < 0) return 0;
  return (long)((C1 + 15) / 16 + (C2 + 15) / 16) * 1024;
}

extern "C" int dca_conv1_lp_prep_weight(const float* w, void* wfrag, int Cout, int C1, int C2, int dtype,
                                        hipStream_t stream) {
  DCA_REQUIRE(w && wfrag && Cout > 0 && Cout <= 32 && C1 > 0 && C2 >= 0 && (dtype == DCA_BF16 || dtype == DCA_FP16));
  const int n1 = (C1 + 15) / 16, n2 = (C2 + 15) / 16;
  DCA_REQUIRE(n1 + n2 <= MAXCH);
  if (dtype == DCA_BF16)
    hipLaunchKernelGGL(conv1_lp_prep_kernel<__bf16>, dim3(n1 + n2), dim3(256), 0, stream, w, (unsigned short*)wfrag, Cout,
                       C1, C2, n1, n2);
  else
    hipLaunchKernelGGL(conv1_lp_prep_kernel<_Float16>, dim3(n1 + n2), dim3(256), 0, stream, w, (unsigned short*)wfrag,
                       Cout, C1, C2, n1, n2);
  return dca_launch_status();
}

extern "C" int dca_conv1_lp_forward(const void* x, const void* x2, const void* wfrag, void* y, const float* scale,
                                    const float* shift, const void* res_pre, const void* res_post, float slope, int N,
                                    int C1, int C2, int Cout, long S, int dtype, int out_f32, hipStream_t stream) {
  DCA_REQUIRE(x && wfrag && y && N > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && Cout <= 32 && S > 0);
  DCA_REQUIRE((x2 != nullptr) == (C2 > 0));
  DCA_REQUIRE(dtype == DCA_BF16 || dtype == DCA_FP16);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE(S % 4 == 0 && (((uintptr_t)x) & 7) == 0 && (((uintptr_t)x2) & 7) == 0 && (((uintptr_t)y) & 15) == 0 &&
              (((uintptr_t)res_pre) & 15) == 0 && (((uintptr_t)res_post) & 15) == 0 && (((uintptr_t)wfrag) & 15) == 0);
  // 32-bit byte offsets inside one sample, with room for the out-of-range marker + 7 channel strides
  const long cmax = (C1 > C2 ? C1 : C2) > 32 ? (C1 > C2 ? C1 : C2) : 32;
  DCA_REQUIRE(cmax * S * 4 < 0x7ffffff0L);
  C1LpArgs a;
  a.x = x; a.x2 = x2; a.wfrag = (const unsigned short*)wfrag; a.y = y;
  a.scale = scale; a.shift = shift; a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.C1 = C1; a.C2 = C2; a.Cout = Cout; a.NCH1 = (C1 + 15) / 16; a.NCH2 = (C2 + 15) / 16; a.S = S;
  DCA_REQUIRE(a.NCH1 + a.NCH2 <= MAXCH);
  const long groups = (long)N * ((S + 127) / 128);
  long blocks = (groups + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (dtype == DCA_BF16) {
    if (out_f32) hipLaunchKernelGGL((conv1_lp_kernel<__bf16, true>), dim3((int)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1_lp_kernel<__bf16, false>), dim3((int)blocks), dim3(256), 0, stream, a);
  } else {
    if (out_f32) hipLaunchKernelGGL((conv1_lp_kernel<_Float16, true>), dim3((int)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1_lp_kernel<_Float16, false>), dim3((int)blocks), dim3(256), 0, stream, a);
  }
  return dca_launch_status();
}
