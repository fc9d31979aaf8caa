// 3D convolutions of the DCANet aggregation path on the CDNA4 matrix cores, exact fp32.
//
// Replaces (reference, PyTorch/ATen): nn.Conv3d / nn.ConvTranspose3d inside `convbn_3d`
// (models/submodule.py:121-124), `dres0/dres1/classif*` (models/gwcnet_dca_g.py:141-168),
// `cva.downsample/classify/fuse` (models/augment/cva.py:39-55), `Multi_Aggregation`
// (cva.py:13-31) and the 1x1x1 projections of SelfAttention_bn.py:136-160.
//
// Layout: NCDHW fp32 in HBM (the reference's layout, so the Python boundary needs no permutes).
// Contraction: v_mfma_f32_32x32x2_f32 with D[cout][voxel] += W[cout][k] * X[k][voxel]; the 32 MFMA
// columns are 32 consecutive W positions, so both the LDS operand reads and the global stores are
// contiguous along W (128 B per half wave).  K runs over (tap, cin) with the input halo tile and
// the per-chunk weights staged in LDS.  The same three kernels serve the backward-data passes with
// re-laid-out weights (see dca_conv3d_prep_weight).
#include "dca_common.h"
#include <stdlib.h>
#include "../../include/dca_hip.h"

struct ConvArgs {
  const float* x;
  const float* x2;
  const float* wt;
  float* y;
  const float* scale;
  const float* shift;
  const float* res_pre;
  const float* res_post;
  float slope;
  int N, Cin, Cout, CinPad;
  int CoutTotal, co_off;  // this launch writes channels [co_off, co_off+Cout) of a CoutTotal-channel output
  int Di, Hi, Wi, Do, Ho, Wo;
  int nTD, nTH, nTW;
  long xs_n, x2s_n;  // batch strides (floats) of x / x2 (1x1 kernel only)
};

__device__ __forceinline__ float epilogue(const ConvArgs& a, float v, int co, long idx) {
  if (a.scale) v = v * a.scale[a.co_off + co] + a.shift[a.co_off + co];
  if (a.res_pre) v += a.res_pre[idx];
  v = act_apply(v, a.slope);
  if (a.res_post) v += a.res_post[idx];
  return v;
}

// 2-byte storage types of the reduced-precision inference path (include/dca_hip.h: DCA_BF16 = 1, DCA_FP16 = 2); 0 = fp32.
// The stride-2 convolution can READ them (XT) and the transposed convolution can WRITE them (YT, residuals included);
// the arithmetic stays the exact-fp32 MFMA of this file.
template <int DT> struct Two;
template <> struct Two<1> { typedef __bf16 T; };
template <> struct Two<2> { typedef _Float16 T; };
template <int DT> __device__ __forceinline__ float two_lo(unsigned w) {
  return (float)__builtin_bit_cast(typename Two<DT>::T, (unsigned short)(w & 0xffffu));
}
template <int DT> __device__ __forceinline__ float two_hi(unsigned w) {
  return (float)__builtin_bit_cast(typename Two<DT>::T, (unsigned short)(w >> 16));
}
template <int DT> __device__ __forceinline__ unsigned two_pack(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef typename Two<DT>::T mtx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mtx2));
}

// ---------------------------------------------------------------------------------------------
// 3x3x3, pad 1, stride S.  Block = 4 waves; output tile TD x TH x TW; an MFMA column tile is 32 voxels =
// (32/TW) consecutive H rows x TW consecutive W positions (TW = 32: one row; TW = 16: two rows, which lets
// W = 240 tile with no wasted lanes).  Each wave owns NT = TD*TH*TW/128 such tiles and all CT*32 output
// channels.  The host picks (TD, TH, TW) per problem to minimise padding waste and grid tail.
// Aligned inputs (VEC) run a software pipeline: the next channel chunk's global loads (input halo tile and
// weights) are issued into registers before the MFMA loop of the current chunk and written to LDS after it.
// ---------------------------------------------------------------------------------------------
template <int S, int CT, int CK, int TD, int TH, int TW, bool VEC, int XT = 0>
__global__ __launch_bounds__(256, 2) void conv3_mfma_kernel(ConvArgs a) {
  static_assert(TW == 32 || (TW == 16 && S == 1), "tile width");
  static_assert(XT == 0 || VEC, "2-byte inputs take the aligned path");
  constexpr int XSZ = XT == 0 ? 4 : 2;
  constexpr int RPT = 32 / TW;                 // H rows per MFMA tile
  constexpr int NT = TD * TH / RPT / 4;        // MFMA tiles per wave
  static_assert(NT * 4 * RPT == TD * TH, "tile shape");
  constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int IWP = ((3 + IW + 3) / 4) * 4;  // interior starts at column 4 (16-byte aligned)
  constexpr int CO = CT * 32;
  constexpr int ROWS = CK * ID * IH;
  constexpr int IN_ELEMS = ROWS * IWP;
  constexpr int QPR = TW * S / 4;              // float4 per interior row
  constexpr int NH = (S == 1) ? 2 : 1;         // halo scalars per row
  constexpr int WQ = CK * CO / 4;              // weight float4 per tap
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;
  float* w_lds = smem + IN_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int wl = l31 & (TW - 1), hsel = l31 / TW;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % a.nTW; bid /= a.nTW;
  const int th = bid % a.nTH; bid /= a.nTH;
  const int td = bid % a.nTD;
  const int n = bid / a.nTD;
  const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
  const int di0 = d0 * S - 1, hi0 = h0 * S - 1, wi0 = w0 * S - 1;

  f32x16 acc[NT][CT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][ct][r] = 0.f;

  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = (wv * NT + t) * RPT + hsel, dl = r / TH, hl = r % TH;
    boff[t] = ((half * ID + dl * S) * IH + hl * S) * IWP + 3 + wl * S;
  }

  constexpr int KX = (ROWS * QPR + 255) / 256, KH = (ROWS * NH + 255) / 256, KW = (27 * WQ + 255) / 256;
  float4 rx[VEC ? KX : 1], rw[VEC ? KW : 1];
  float rh[VEC ? KH : 1];

  // Prefetch loads are hardware-predicated buffer loads (dca_common.h): straight-line code, masked elements -> 0.
  const long sample = (long)a.Cin * a.Di * a.Hi * a.Wi;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc((const char*)a.x + (long)n * sample * XSZ, sample * XSZ);
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc(a.wt, (long)27 * a.CinPad * CO * 4);
  const int cstride = a.Di * a.Hi * a.Wi;
  auto load_regs = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 256 * k;
      const int row = it / QPR, q = it % QPR;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = ci0 + c, di = di0 + id, hi = hi0 + ih, wi = wi0 + 1 + 4 * q;
      const int ok = (int)(it < ROWS * QPR) & (int)(ci < a.Cin) & (int)((unsigned)di < (unsigned)a.Di) &
                     (int)((unsigned)hi < (unsigned)a.Hi) & (int)(wi < a.Wi);
      if constexpr (XT == 0) {
        rx[k] = dca_bload4(xr, (ci * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4, ok);
      } else {   // four 2-byte values, kept raw until store_regs (a conversion here would wait for the load)
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xr, dca_pred_off((ci * cstride + (di * a.Hi + hi) * a.Wi + wi) * 2, ok), 0, 0);
        rx[k].x = __uint_as_float(v.x); rx[k].y = __uint_as_float(v.y);
      }
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int it = tid + 256 * k;
      const int row = it / NH, j = (it % NH) ? (IW - 1) : 0;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = ci0 + c, di = di0 + id, hi = hi0 + ih, wi = wi0 + j;
      const int ok = (int)(it < ROWS * NH) & (int)(ci < a.Cin) & (int)((unsigned)di < (unsigned)a.Di) &
                     (int)((unsigned)hi < (unsigned)a.Hi) & (int)((unsigned)wi < (unsigned)a.Wi);
      if constexpr (XT == 0) rh[k] = dca_bload1(xr, (ci * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4, ok);
      else rh[k] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
               xr, dca_pred_off((ci * cstride + (di * a.Hi + hi) * a.Wi + wi) * 2, ok), 0, 0));
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int it = tid + 256 * k;
      const int tap = it / WQ, q = it % WQ;
      rw[k] = dca_bload4(wr, ((tap * a.CinPad + ci0) * CO + 4 * q) * 4, (int)(it < 27 * WQ));
    }
  };
  auto store_regs = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 256 * k;
      if constexpr (XT != 0) {
        const unsigned w0_ = __float_as_uint(rx[k].x), w1_ = __float_as_uint(rx[k].y);
        rx[k] = make_float4(two_lo<XT ? XT : 1>(w0_), two_hi<XT ? XT : 1>(w0_), two_lo<XT ? XT : 1>(w1_), two_hi<XT ? XT : 1>(w1_));
      }
      if (it < ROWS * QPR) *(float4*)(in_lds + (it / QPR) * IWP + 4 + 4 * (it % QPR)) = rx[k];
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int it = tid + 256 * k;
      if constexpr (XT != 0) rh[k] = two_lo<XT ? XT : 1>(__float_as_uint(rh[k]));
      if (it < ROWS * NH) in_lds[(it / NH) * IWP + 3 + ((it % NH) ? (IW - 1) : 0)] = rh[k];
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int it = tid + 256 * k;
      if (it < 27 * WQ) *(float4*)(w_lds + (it / WQ) * CK * CO + 4 * (it % WQ)) = rw[k];
    }
  };

  if constexpr (VEC) load_regs(0);
  for (int ci0 = 0; ci0 < a.CinPad; ci0 += CK) {
    __syncthreads();
    if constexpr (VEC) {
      store_regs();
    } else {
      for (int it = tid; it < ROWS * IW; it += 256) {
        const int row = it / IW, j = it % IW;
        const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
        const int ci = ci0 + c, di = di0 + id, hi = hi0 + ih, wi = wi0 + j;
        float v = 0.f;
        if (ci < a.Cin && (unsigned)di < (unsigned)a.Di && (unsigned)hi < (unsigned)a.Hi &&
            (unsigned)wi < (unsigned)a.Wi)
          v = a.x[((((long)n * a.Cin + ci) * a.Di + di) * a.Hi + hi) * a.Wi + wi];
        in_lds[row * IWP + 3 + j] = v;
      }
      for (int it = tid; it < 27 * WQ; it += 256) {
        const int tap = it / WQ, q = it % WQ;
        *(float4*)(w_lds + tap * CK * CO + 4 * q) =
            *(const float4*)(a.wt + ((long)tap * a.CinPad + ci0) * CO + 4 * q);
      }
    }
    __syncthreads();
    if constexpr (VEC) {
      if (ci0 + CK < a.CinPad) load_regs(ci0 + CK);
    }
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
        const float* inb = in_lds + (kd * IH + kh) * IWP;
        const float* wb = w_lds + (kd * 3 + kh) * 3 * CK * CO + half * CO + l31;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
          for (int kk = 0; kk < CK / 2; ++kk) {
            float av[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[ct] = wb[(kw * CK + 2 * kk) * CO + ct * 32];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float bv = inb[boff[t] + kk * 2 * ID * IH * IWP + kw];
#pragma unroll
              for (int ct = 0; ct < CT; ++ct)
                acc[t][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct], bv, acc[t][ct], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // Epilogue.  The optional affine / residual operands are loaded in batches behind uniform null checks (a per-
  // element `if (ptr) v += ptr[idx]` compiles to one dependent load + s_waitcnt vmcnt(0) per output element).
  const int w = w0 + wl;
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = min(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
      sc[r] = has_aff ? a.scale[a.co_off + co] : 1.f;
      sh[r] = has_aff ? a.shift[a.co_off + co] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int r0 = (wv * NT + t) * RPT + hsel, d = d0 + r0 / TH, h = h0 + r0 % TH;
      const bool ok = d < a.Do && h < a.Ho && w < a.Wo;
      const long plane = (long)a.Do * a.Ho * a.Wo;
      const long base = (((long)n * a.CoutTotal + a.co_off) * a.Do + (ok ? d : 0)) * a.Ho * (long)a.Wo +
                        (long)(ok ? h : 0) * a.Wo + (ok ? w : 0);
      float rp[16], rq[16];
      if (has_pre) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = min(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
          rp[r] = a.res_pre[base + co * plane];
        }
      }
      if (has_post) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = min(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
          rq[r] = a.res_post[base + co * plane];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[t][ct][r] * sc[r] + sh[r];
        if (has_pre) v += rp[r];
        v = act_apply(v, a.slope);
        if (has_post) v += rq[r];
        if (ok && co < a.Cout) a.y[base + co * plane] = v;
      }
    }
  }
}

template <int S, int CT, int CK, int TD, int TH, int TW>
static int launch_conv3(ConvArgs& a, bool vec, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Transposed 3x3x3, stride 2, pad 1, output_padding 1 (out = 2*in).  out[o] += x[m] W[k] with
// o = 2m - 1 + k: per dim, k=1 feeds even outputs from x[m]; k=0 / k=2 feed odd outputs 2m+1 from
// x[m+1] / x[m].  Each wave owns one row of 32 coarse positions and keeps the 8 output-parity
// classes in 8 accumulators; the 27 taps are distributed over them at compile time.
// ---------------------------------------------------------------------------------------------
template <int CK, bool VEC, int YT = 0>
__global__ __launch_bounds__(256, 2) void deconv3_mfma_kernel(ConvArgs a) {
  constexpr int ID = 3, IH = 3, IWP = 40, CO = 32;
  constexpr int ROWS = CK * ID * IH;
  constexpr int IN_ELEMS = ROWS * IWP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;
  float* w_lds = smem + IN_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % a.nTW; bid /= a.nTW;
  const int th = bid % a.nTH; bid /= a.nTH;
  const int td = bid % a.nTD;
  const int n = bid / a.nTD;
  const int md0 = td * 2, mh0 = th * 2, mw0 = tw * 32;
  const int mdl = wv >> 1, mhl = wv & 1;

  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  const int boff = ((half * ID + mdl) * IH + mhl) * IWP + 4 + l31;

  // software pipeline (aligned inputs): next chunk's global loads are in flight during the MFMA block
  constexpr int WQ = CK * CO / 4;
  constexpr int KX = (ROWS * 8 + 255) / 256, KH = (ROWS + 255) / 256, KW = (27 * WQ + 255) / 256;
  float4 rx[VEC ? KX : 1], rw[VEC ? KW : 1];
  float rh[VEC ? KH : 1];
  const long sample = (long)a.Cin * a.Di * a.Hi * a.Wi;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc(a.wt, (long)27 * a.CinPad * CO * 4);
  const int cstride = a.Di * a.Hi * a.Wi;
  auto load_regs = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 256 * k;
      const int row = it >> 3, q = it & 7;
      const int c = row / 9, rem = row % 9, id = rem / 3, ih = rem % 3;
      const int ci = ci0 + c, di = md0 + id, hi = mh0 + ih, wi = mw0 + 4 * q;
      const int ok = (int)(it < ROWS * 8) & (int)(ci < a.Cin) & (int)(di < a.Di) & (int)(hi < a.Hi) & (int)(wi < a.Wi);
      rx[k] = dca_bload4(xr, (ci * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4, ok);
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int row = tid + 256 * k;
      const int c = row / 9, rem = row % 9, id = rem / 3, ih = rem % 3;
      const int ci = ci0 + c, di = md0 + id, hi = mh0 + ih, wi = mw0 + 32;
      const int ok = (int)(row < ROWS) & (int)(ci < a.Cin) & (int)(di < a.Di) & (int)(hi < a.Hi) & (int)(wi < a.Wi);
      rh[k] = dca_bload1(xr, (ci * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4, ok);
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int it = tid + 256 * k;
      rw[k] = dca_bload4(wr, (((it / WQ) * a.CinPad + ci0) * CO + 4 * (it % WQ)) * 4, (int)(it < 27 * WQ));
    }
  };
  auto store_regs = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 256 * k;
      if (it < ROWS * 8) *(float4*)(in_lds + (it >> 3) * IWP + 4 + 4 * (it & 7)) = rx[k];
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int row = tid + 256 * k;
      if (row < ROWS) in_lds[row * IWP + 36] = rh[k];
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int it = tid + 256 * k;
      if (it < 27 * WQ) *(float4*)(w_lds + (it / WQ) * CK * CO + 4 * (it % WQ)) = rw[k];
    }
  };

  if constexpr (VEC) load_regs(0);
  for (int ci0 = 0; ci0 < a.CinPad; ci0 += CK) {
    __syncthreads();
    if constexpr (VEC) {
      store_regs();
    } else {
      for (int it = tid; it < ROWS * 33; it += 256) {
        const int row = it / 33, j = it % 33;
        const int c = row / 9, rem = row % 9, id = rem / 3, ih = rem % 3;
        const int ci = ci0 + c, di = md0 + id, hi = mh0 + ih, wi = mw0 + j;
        float v = 0.f;
        if (ci < a.Cin && di < a.Di && hi < a.Hi && wi < a.Wi)
          v = a.x[((((long)n * a.Cin + ci) * a.Di + di) * a.Hi + hi) * a.Wi + wi];
        in_lds[row * IWP + 4 + j] = v;
      }
      for (int it = tid; it < 27 * WQ; it += 256) {
        const int tap = it / WQ, q = it % WQ;
        *(float4*)(w_lds + tap * CK * CO + 4 * q) =
            *(const float4*)(a.wt + ((long)tap * a.CinPad + ci0) * CO + 4 * q);
      }
    }
    __syncthreads();
    if constexpr (VEC) {
      if (ci0 + CK < a.CinPad) load_regs(ci0 + CK);
    }
    const float* wb = w_lds + half * CO + l31;
    const float* inb = in_lds + boff;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int pc = ((kd != 1) * 2 + (kh != 1)) * 2 + (kw != 1);
          const int off = ((kd == 0) * IH + (kh == 0)) * IWP + (kw == 0);
          const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int kk = 0; kk < CK / 2; ++kk) {
            const float av = wb[(tap * CK + 2 * kk) * CO];
            const float bv = inb[kk * 2 * ID * IH * IWP + off];
            acc[pc] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[pc], 0, 0, 0);
          }
        }
  }

  const int md = md0 + mdl, mh = mh0 + mhl, mw = mw0 + l31;
  const bool ok = md < a.Di && mh < a.Hi && mw < a.Wi;
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  float sc[16], sh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = min((r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
    sc[r] = has_aff ? a.scale[a.co_off + co] : 1.f;
    sh[r] = has_aff ? a.shift[a.co_off + co] : 0.f;
  }
  const long plane = (long)a.Do * a.Ho * a.Wo;
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int d = ok ? 2 * md + pd : 0, h = ok ? 2 * mh + ph : 0, w = ok ? 2 * mw : 0;
      const long base = (((long)n * a.CoutTotal + a.co_off) * a.Do + d) * a.Ho * (long)a.Wo + (long)h * a.Wo + w;
#pragma unroll
      for (int rc = 0; rc < 16; rc += 8) {
        float2 rp[8], rq[8];
        if (has_pre) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = rc + q;
            const long off = base + min((r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1) * plane;
            if constexpr (YT == 0) rp[q] = *(const float2*)(a.res_pre + off);
            else {
              const unsigned w_ = *(const unsigned*)((const unsigned short*)a.res_pre + off);
              rp[q] = make_float2(two_lo<YT ? YT : 1>(w_), two_hi<YT ? YT : 1>(w_));
            }
          }
        }
        if (has_post) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = rc + q;
            const long off = base + min((r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1) * plane;
            if constexpr (YT == 0) rq[q] = *(const float2*)(a.res_post + off);
            else {
              const unsigned w_ = *(const unsigned*)((const unsigned short*)a.res_post + off);
              rq[q] = make_float2(two_lo<YT ? YT : 1>(w_), two_hi<YT ? YT : 1>(w_));
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = rc + q, co = (r & 3) + 8 * (r >> 2) + 4 * half;
          float2 o = make_float2(acc[(pd * 2 + ph) * 2 + 0][r] * sc[r] + sh[r], acc[(pd * 2 + ph) * 2 + 1][r] * sc[r] + sh[r]);
          if (has_pre) { o.x += rp[q].x; o.y += rp[q].y; }
          o.x = act_apply(o.x, a.slope); o.y = act_apply(o.y, a.slope);
          if (has_post) { o.x += rq[q].x; o.y += rq[q].y; }
          if (ok && co < a.Cout) {
            if constexpr (YT == 0) *(float2*)(a.y + base + co * plane) = o;
            else *(unsigned*)((unsigned short*)a.y + base + co * plane) = two_pack<YT ? YT : 1>(o.x, o.y);
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1x1 convolution (pointwise GEMM), Cout <= 32, Cin = 32*NG taken from x (first 32) and x2.
// Each wave streams groups of 128 consecutive voxels: lane (i, half) loads float4 x[ci=2kk+half]
// [v0+4i..]; MFMA tile j is the voxel set {v0+4i+j}, so loads and stores are 16 B per lane.
// The weight fragments stay in registers.
// ---------------------------------------------------------------------------------------------
// Round 2: the channels of a group go in two batches of 8 k-pairs, scale / shift live in LDS and the epilogue handles two
// output rows at a time (124 -> 120 us for 32->32 at 48x136x240).  Forcing two workgroups per CU (CONV1_OCC=2) still
// spills (131 us): a wave's 64 fp32 MFMAs (4096 matrix-pipe cycles per 128 voxels) are the latency this kernel cannot
// hide, which is what the reduced-precision conv1_lp.hip (1536 cycles, 4.8-5.6 TB/s) does not have.
#ifndef CONV1_OCC
#define CONV1_OCC 1
#endif
template <int NG, bool VEC>
__global__ __launch_bounds__(256, CONV1_OCC) void conv1_mfma_kernel(ConvArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const long DHW = (long)a.Do * a.Ho * a.Wo;
  const long ngroups = (DHW + 127) / 128, total = (long)a.N * ngroups;
  __shared__ float aff[64];
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  if (threadIdx.x < 64) {
    const int co = min((int)(threadIdx.x & 31), a.Cout - 1);
    aff[threadIdx.x] = has_aff ? (threadIdx.x < 32 ? a.scale[a.co_off + co] : a.shift[a.co_off + co])
                               : (threadIdx.x < 32 ? 1.f : 0.f);
  }
  float aw[NG * 16];
#pragma unroll
  for (int kk = 0; kk < NG * 16; ++kk) aw[kk] = a.wt[(2 * kk + half) * 32 + l31];
  __syncthreads();
  // hardware-predicated loads of the activations (dca_common.h): 32-bit offsets inside one (sample, tensor)
  const bool boff_ok = a.xs_n * 4 < 0x7ffffff0L && a.x2s_n * 4 < 0x7ffffff0L;
  for (long g = (long)blockIdx.x * 4 + wv; g < total; g += (long)gridDim.x * 4) {
    const int n = (int)(g / ngroups);
    const long v = (g % ngroups) * 128 + 4 * l31;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      const float* xbase = (grp == 0 ? a.x + n * a.xs_n : a.x2 + n * a.x2s_n);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {       // two batches of 8 k-pairs (16 channels)
        float4 b[8];
        if (VEC && boff_ok) {
          const __amdgpu_buffer_rsrc_t xr = dca_rsrc(xbase, (grp == 0 ? a.xs_n : a.x2s_n) * 4);
#pragma unroll
          for (int kk = 0; kk < 8; ++kk)
            b[kk] = dca_bload4(xr, (int)(((long)(2 * (hb * 8 + kk) + half) * DHW + v) * 4), (int)(v < DHW));
        } else {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const float* p = xbase + v + (long)(2 * (hb * 8 + kk) + half) * DHW;
            b[kk].x = (v + 0 < DHW) ? p[0] : 0.f;
            b[kk].y = (v + 1 < DHW) ? p[1] : 0.f;
            b[kk].z = (v + 2 < DHW) ? p[2] : 0.f;
            b[kk].w = (v + 3 < DHW) ? p[3] : 0.f;
          }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float w = aw[grp * 16 + hb * 8 + kk];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b[kk].x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b[kk].y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b[kk].z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b[kk].w, acc[3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the next batch's 8 loads behind this batch's MFMAs (register budget)
      }
    }
    const bool inr = v < DHW;
    const long base = ((long)n * a.CoutTotal + a.co_off) * DHW + (inr ? v : 0);
    if (VEC) {
#pragma unroll
      for (int rc = 0; rc < 16; rc += 2) {  // two rows at a time
        float4 rp[2], rq[2];
        if (has_pre) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int r = rc + q;
            rp[q] = *(const float4*)(a.res_pre + base + min((r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1) * DHW);
          }
        }
        if (has_post) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int r = rc + q;
            rq[q] = *(const float4*)(a.res_post + base + min((r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1) * DHW);
          }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = rc + q, cl = (r & 3) + 8 * (r >> 2) + 4 * half, co = cl;
          const float sc = aff[cl], sh = aff[32 + cl];
          float o[4] = {acc[0][r] * sc + sh, acc[1][r] * sc + sh, acc[2][r] * sc + sh, acc[3][r] * sc + sh};
          if (has_pre) { o[0] += rp[q].x; o[1] += rp[q].y; o[2] += rp[q].z; o[3] += rp[q].w; }
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = act_apply(o[j], a.slope);
          if (has_post) { o[0] += rq[q].x; o[1] += rq[q].y; o[2] += rq[q].z; o[3] += rq[q].w; }
          if (inr && co < a.Cout) *(float4*)(a.y + base + co * DHW) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= a.Cout) continue;
        const long idx = ((long)n * a.CoutTotal + a.co_off + co) * DHW + v;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v + j < DHW) a.y[idx + j] = epilogue(a, acc[j][r], co, idx + j);
      }
    }
  }
}

// dst[tap][a][b] (a < Apad rows = contraction channels, b < Bpad = output channels, zero padded)
// from a PyTorch weight: src_ab ? src[a][b][K] : src[b][a][K]; flip reverses the tap order.
__global__ void prep_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int Bn,
                                   int Apad, int Bpad, int K, int src_ab, int flip, int Btotal, int b_off) {
  const int total = K * Apad * Bpad;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int tap = idx / (Apad * Bpad), ai = (idx / Bpad) % Apad, bi = idx % Bpad;
    float v = 0.f;
    if (ai < A && bi < Bn) {
      const int st = flip ? K - 1 - tap : tap;
      v = src_ab ? src[((long)ai * Btotal + b_off + bi) * K + st] : src[((long)(b_off + bi) * A + ai) * K + st];
    }
    dst[idx] = v;
  }
}

template <typename KernelT>
static int launch_conv(KernelT kernel, const ConvArgs& a, int grid, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
  return dca_launch_status();
}

template <int S, int CT, int CK, int TD, int TH, int TW>
static int launch_conv3(ConvArgs& a, bool vec, hipStream_t stream) {
  constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int IWP = ((3 + IW + 3) / 4) * 4;
  const size_t lds = (size_t)(CK * ID * IH * IWP + 27 * CK * CT * 32) * 4;
  a.nTD = cdiv(a.Do, TD); a.nTH = cdiv(a.Ho, TH); a.nTW = cdiv(a.Wo, TW);
  const int grid = a.N * a.nTD * a.nTH * a.nTW;
  return vec ? launch_conv(conv3_mfma_kernel<S, CT, CK, TD, TH, TW, true>, a, grid, lds, stream)
             : launch_conv(conv3_mfma_kernel<S, CT, CK, TD, TH, TW, false>, a, grid, lds, stream);
}

extern "C" int dca_conv3d_prep_weight(const float* w, float* wt, int A, int B, int Apad, int Bpad, int K,
                                      int src_ab, int flip, int Btotal, int b_off, hipStream_t stream) {
  DCA_REQUIRE(w && wt && A > 0 && B > 0 && Apad >= A && Bpad >= B && (K == 1 || K == 27));
  DCA_REQUIRE(b_off >= 0 && b_off + B <= Btotal);
  const int total = K * Apad * Bpad;
  hipLaunchKernelGGL(prep_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, w, wt, A, B, Apad, Bpad, K,
                     src_ab, flip, Btotal, b_off);
  return dca_launch_status();
}

extern "C" int dca_conv3d_forward(const float* x, const float* x2, const float* wt, float* y, const float* scale,
                                  const float* shift, const float* res_pre, const float* res_post, float slope,
                                  int N, int Cin, int C1, int Cout, int CinPad, int CoutTotal, int co_off, int Di,
                                  int Hi, int Wi, int Do, int Ho, int Wo, int ksize, int stride, int transposed,
                                  hipStream_t stream) {
  DCA_REQUIRE(x && wt && y && N > 0 && Cin > 0 && Cout > 0 && CinPad >= Cin);
  DCA_REQUIRE(co_off >= 0 && co_off + Cout <= CoutTotal);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  ConvArgs a;
  a.x = x; a.x2 = x2; a.wt = wt; a.y = y; a.scale = scale; a.shift = shift;
  a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.CinPad = CinPad; a.CoutTotal = CoutTotal; a.co_off = co_off;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.nTD = a.nTH = a.nTW = 1; a.xs_n = a.x2s_n = 0;
  const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wt) & 15) == 0;

  if (ksize == 1) {
    DCA_REQUIRE(stride == 1 && !transposed && Cout <= 32 && Di == Do && Hi == Ho && Wi == Wo);
    const long DHW = (long)Do * Ho * Wo;
    int NG;
    if (x2) {
      DCA_REQUIRE(C1 == 32 && Cin == 64 && CinPad == 64);
      NG = 2; a.xs_n = 32 * DHW; a.x2s_n = 32 * DHW;
    } else {
      DCA_REQUIRE((Cin == 32 && CinPad == 32) || (Cin == 64 && CinPad == 64));
      NG = Cin / 32; a.xs_n = (long)Cin * DHW; a.x2s_n = a.xs_n; a.x2 = x + 32 * DHW;
    }
    const bool vec = aligned && (DHW % 4 == 0) && ((((uintptr_t)a.x2) & 15) == 0);
    const long total = (long)N * ((DHW + 127) / 128);
    const int grid = (int)((total + 3) / 4 < 2048 ? (total + 3) / 4 : 2048);
    if (NG == 1) return vec ? launch_conv(conv1_mfma_kernel<1, true>, a, grid, 0, stream)
                            : launch_conv(conv1_mfma_kernel<1, false>, a, grid, 0, stream);
    return vec ? launch_conv(conv1_mfma_kernel<2, true>, a, grid, 0, stream)
               : launch_conv(conv1_mfma_kernel<2, false>, a, grid, 0, stream);
  }
  DCA_REQUIRE(ksize == 3 && x2 == nullptr);
  const bool vec = aligned && (Wi % 4 == 0) && ((long)Cin * Di * Hi * Wi * 4 < 0x7ffffff0L);
  if (transposed) {
    DCA_REQUIRE(stride == 2 && Cout <= 32 && Do == 2 * Di && Ho == 2 * Hi && Wo == 2 * Wi && CinPad % 8 == 0);
    a.nTD = cdiv(Di, 2); a.nTH = cdiv(Hi, 2); a.nTW = cdiv(Wi, 32);
    const int grid = N * a.nTD * a.nTH * a.nTW;
    const size_t lds = (size_t)(8 * 9 * 40 + 27 * 8 * 32) * 4;
    return vec ? launch_conv(deconv3_mfma_kernel<8, true>, a, grid, lds, stream)
               : launch_conv(deconv3_mfma_kernel<8, false>, a, grid, lds, stream);
  }
  DCA_REQUIRE(Cout <= 64 && CinPad % 8 == 0);
  if (stride == 1) {
    DCA_REQUIRE(Do == Di && Ho == Hi && Wo == Wi);
    // candidate tile shapes (TD, TH, TW); pick the one with the fewest padded voxels, then the fewest blocks
    struct Shape { int td, th, tw; };
    const Shape shapes[3] = {{2, 8, 32}, {4, 4, 32}, {4, 8, 16}};
    int best = 0;
    long best_cost = -1;
    for (int i = 0; i < 3; ++i) {
      const long cost = (long)cdiv(Do, shapes[i].td) * shapes[i].td * cdiv(Ho, shapes[i].th) * shapes[i].th *
                        cdiv(Wo, shapes[i].tw) * shapes[i].tw;
      if (best_cost < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    // small volumes: fewer than two workgroups per CU with 512-voxel tiles -> 128-voxel tiles (4x the workgroups)
    const long nblk = (long)N * cdiv(Do, shapes[best].td) * cdiv(Ho, shapes[best].th) * cdiv(Wo, shapes[best].tw);
    const bool small = nblk < 512;
    if (Cout <= 32) {
      if (small) return launch_conv3<1, 1, 8, 2, 4, 16>(a, vec, stream);
      if (best == 0) return launch_conv3<1, 1, 8, 2, 8, 32>(a, vec, stream);
      if (best == 1) return launch_conv3<1, 1, 8, 4, 4, 32>(a, vec, stream);
      return launch_conv3<1, 1, 8, 4, 8, 16>(a, vec, stream);
    }
    if (small) return launch_conv3<1, 2, 4, 2, 4, 16>(a, vec, stream);
    if (best == 0) return launch_conv3<1, 2, 4, 2, 8, 32>(a, vec, stream);
    if (best == 1) return launch_conv3<1, 2, 4, 4, 4, 32>(a, vec, stream);
    return launch_conv3<1, 2, 4, 4, 8, 16>(a, vec, stream);
  }
  DCA_REQUIRE(stride == 2 && Do == (Di + 1) / 2 && Ho == (Hi + 1) / 2 && Wo == (Wi + 1) / 2);
  // Few big tiles quantise badly (816 workgroups of 2 x 4 x 32 on 512 slots at 24 x 68 x 120): below ~4 rounds use
  // 1 x 4 x 32 tiles with 2-channel chunks (29 KB of LDS, 123 registers -> four workgroups per CU): 265 -> 243 us.
  if ((long)N * cdiv(Do, 2) * cdiv(Ho, 4) * cdiv(Wo, 32) < 2048) return launch_conv3<2, 2, 2, 1, 4, 32>(a, vec, stream);
  return launch_conv3<2, 2, 4, 2, 4, 32>(a, vec, stream);
}


// Mixed-storage launches of the reduced-precision inference path (the arithmetic is this file's exact-fp32 MFMA):
//   transposed == 0: 3x3x3 stride-2 convolution reading a 2-byte x (dtype), writing fp32 y (cost_agg.conv1, cva.py:16-17)
//   transposed == 1: ConvTranspose3d(3, s2, p1, op1) reading fp32 x, writing y and reading res_pre / res_post in the
//                    2-byte type (cost_agg.conv3 + ReLU(. + redir) [+ outer residual], cva.py:21-29)
extern "C" int dca_conv3d_forward_mixed(const void* x, const float* wt, void* y, const float* scale, const float* shift,
                                        const void* res_pre, const void* res_post, float slope, int N, int Cin, int Cout,
                                        int CinPad, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int transposed,
                                        int dtype, hipStream_t stream) {
  DCA_REQUIRE(x && wt && y && N > 0 && Cin > 0 && Cout > 0 && CinPad >= Cin && CinPad % 8 == 0);
  DCA_REQUIRE(dtype == DCA_BF16 || dtype == DCA_FP16);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE(Wi % 4 == 0 && (long)Cin * Di * Hi * Wi * 4 < 0x7ffffff0L);
  DCA_REQUIRE(((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wt | (uintptr_t)res_pre | (uintptr_t)res_post)) & 15) == 0);
  ConvArgs a;
  a.x = (const float*)x; a.x2 = nullptr; a.wt = wt; a.y = (float*)y; a.scale = scale; a.shift = shift;
  a.res_pre = (const float*)res_pre; a.res_post = (const float*)res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.CinPad = CinPad; a.CoutTotal = Cout; a.co_off = 0;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.nTD = a.nTH = a.nTW = 1; a.xs_n = a.x2s_n = 0;
  if (transposed) {
    DCA_REQUIRE(Cout <= 32 && Do == 2 * Di && Ho == 2 * Hi && Wo == 2 * Wi);
    a.nTD = cdiv(Di, 2); a.nTH = cdiv(Hi, 2); a.nTW = cdiv(Wi, 32);
    const int grid = N * a.nTD * a.nTH * a.nTW;
    const size_t lds = (size_t)(8 * 9 * 40 + 27 * 8 * 32) * 4;
    return dtype == DCA_BF16 ? launch_conv(deconv3_mfma_kernel<8, true, 1>, a, grid, lds, stream)
                             : launch_conv(deconv3_mfma_kernel<8, true, 2>, a, grid, lds, stream);
  }
  DCA_REQUIRE(res_pre == nullptr && res_post == nullptr);   // (fp32 residual reads are not wired for this form)
  DCA_REQUIRE(Cout <= 64 && Do == (Di + 1) / 2 && Ho == (Hi + 1) / 2 && Wo == (Wi + 1) / 2);
  constexpr int CT = 2;
  const bool small = (long)N * cdiv(Do, 2) * cdiv(Ho, 4) * cdiv(Wo, 32) < 2048;
  if (small) {
    constexpr int CK = 2, TD = 1, TH = 4, TW = 32, S = 2;
    constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, IWP = ((3 + IW + 3) / 4) * 4;
    const size_t lds = (size_t)(CK * ID * IH * IWP + 27 * CK * CT * 32) * 4;
    a.nTD = cdiv(Do, TD); a.nTH = cdiv(Ho, TH); a.nTW = cdiv(Wo, TW);
    const int grid = N * a.nTD * a.nTH * a.nTW;
    return dtype == DCA_BF16 ? launch_conv(conv3_mfma_kernel<S, CT, CK, TD, TH, TW, true, 1>, a, grid, lds, stream)
                             : launch_conv(conv3_mfma_kernel<S, CT, CK, TD, TH, TW, true, 2>, a, grid, lds, stream);
  }
  constexpr int CK = 4, TD = 2, TH = 4, TW = 32, S = 2;
  constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, IWP = ((3 + IW + 3) / 4) * 4;
  const size_t lds = (size_t)(CK * ID * IH * IWP + 27 * CK * CT * 32) * 4;
  a.nTD = cdiv(Do, TD); a.nTH = cdiv(Ho, TH); a.nTW = cdiv(Wo, TW);
  const int grid = N * a.nTD * a.nTH * a.nTW;
  return dtype == DCA_BF16 ? launch_conv(conv3_mfma_kernel<S, CT, CK, TD, TH, TW, true, 1>, a, grid, lds, stream)
                           : launch_conv(conv3_mfma_kernel<S, CT, CK, TD, TH, TW, true, 2>, a, grid, lds, stream);
}
