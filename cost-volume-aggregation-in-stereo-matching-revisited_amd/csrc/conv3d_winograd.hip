// Winograd F(2x2, 3x3) over (H, W), direct over D, for the dominant 3x3x3 stride-1 convolutions (Cout <= 32) of the
// DCANet aggregation path on the fp32 matrix cores (gfx950).
//
// Replaces the same reference ops as conv3d_mfma.hip (nn.Conv3d k3 s1 p1 inside convbn_3d, models/submodule.py:
// 121-124, gwcnet_dca_g.py:141-168, cva.py:39-53) and their stride-1 backward-data passes.
//
//   Y(2x2) = A^T [ sum_{kd, ci} U_{kd,co,ci} (.) V_{ci, d+kd-1} ] A,   V = B^T x(4x4) B,   U = G g(3x3) G^T
//
// 16 independent GEMMs (one per Winograd coefficient xi) of K = 3*Cin instead of one GEMM of K = 27*Cin:
// 2.25x fewer MFMAs per output.  Block = 4 waves, 2 output planes x (4 x 8) tiles of 2x2 outputs (= 2 x 8 x 16
// voxels).  Wave w owns xi_h = w (4 coefficients) for both planes = 8 accumulators.  Per 4-channel chunk every
// thread fetches two 4x4 input patches straight from global memory into registers (prefetched one chunk ahead),
// transforms them (32 add/sub) and writes the 16 coefficients to LDS with the tile index on the lanes; the chunk's
// transformed weights U are copied alongside.  The inverse transform is wave-local along W and goes through LDS
// once along H.  Same epilogue (affine + activation + residuals, channel offset) as the direct kernels.
#include "dca_common.h"
#include "../../include/dca_hip.h"

namespace {
constexpr int CK = 4, PD = 4, V_ELEMS = 16 * CK * PD * 32, U_ELEMS = 16 * 3 * CK * 32;

struct WinoArgs {
  const float* x;
  const float* ug;   // [16*3][CinPad][32]
  float* y;
  const float* scale;
  const float* shift;
  const float* res_pre;
  const float* res_post;
  float slope;
  int N, Cin, Cout, CinPad, CoutTotal, co_off;
  int D, H, W;
  int nTD, nTH, nTW;
};

__device__ __forceinline__ float wepilogue(const WinoArgs& a, float v, int co, long idx) {
  if (a.scale) v = v * a.scale[a.co_off + co] + a.shift[a.co_off + co];
  if (a.res_pre) v += a.res_pre[idx];
  v = act_apply(v, a.slope);
  if (a.res_post) v += a.res_post[idx];
  return v;
}

__global__ __launch_bounds__(256, 2) void wino_conv3_kernel(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;             // [16 xi][CK][PD][32 tiles]
  float* Us = smem + V_ELEMS;   // [16 xi][3 kd][CK][32 co]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % a.nTW; bid /= a.nTW;
  const int th = bid % a.nTH; bid /= a.nTH;
  const int td = bid % a.nTD;
  const int n = bid / a.nTD;
  const int d0 = td * 2, h0 = th * 8, w0 = tw * 16;

  // staging role: patch (c = tid>>7 (+2 for the second), plane pp, tile pt)
  const int pt = tid & 31, pp = (tid >> 5) & 3, pc0 = tid >> 7;
  const int ph0 = h0 + 2 * (pt >> 3) - 1, pw0 = w0 + 2 * (pt & 7) - 1, pd = d0 - 1 + pp;
  const bool pd_ok = (unsigned)pd < (unsigned)a.D;
  float px[2][16];
  float4 ru[6];

  // patch loads are hardware-predicated buffer loads (dca_common.h): out-of-volume elements come back as 0
  const long sample = (long)a.Cin * a.D * a.H * a.W;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
  int hoff[4], hok[4], wok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    hok[r] = (int)pd_ok & (int)((unsigned)(ph0 + r) < (unsigned)a.H);
    hoff[r] = (pd * a.H + ph0 + r) * a.W + pw0;
    wok[r] = (int)((unsigned)(pw0 + r) < (unsigned)a.W);
  }
  const int cstride = a.D * a.H * a.W;
  auto load_regs = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ci = ci0 + pc0 + 2 * k;
      const int cok = (int)(ci < a.Cin);
      const int cbase = ci * cstride;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          px[k][r * 4 + s] = dca_bload1(xr, (cbase + hoff[r] + s) * 4, cok & hok[r] & wok[s]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int it = tid + 256 * k;              // 1536 float4: (xi*3+kd) in 0..47, c in 0..3, co4 in 0..7
      const int xk = it >> 5, c = (it >> 3) & 3, q = it & 7;
      ru[k] = *(const float4*)(a.ug + ((long)xk * a.CinPad + ci0 + c) * 32 + 4 * q);
    }
  };
  auto store_regs = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float t[16], v[16];
#pragma unroll
      for (int s = 0; s < 4; ++s) {              // B^T along rows (h)
        t[0 * 4 + s] = px[k][0 * 4 + s] - px[k][2 * 4 + s];
        t[1 * 4 + s] = px[k][1 * 4 + s] + px[k][2 * 4 + s];
        t[2 * 4 + s] = px[k][2 * 4 + s] - px[k][1 * 4 + s];
        t[3 * 4 + s] = px[k][1 * 4 + s] - px[k][3 * 4 + s];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {              // ... and along columns (w)
        v[r * 4 + 0] = t[r * 4 + 0] - t[r * 4 + 2];
        v[r * 4 + 1] = t[r * 4 + 1] + t[r * 4 + 2];
        v[r * 4 + 2] = t[r * 4 + 2] - t[r * 4 + 1];
        v[r * 4 + 3] = t[r * 4 + 1] - t[r * 4 + 3];
      }
      const int c = pc0 + 2 * k;
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) Vs[((xi * CK + c) * PD + pp) * 32 + pt] = v[xi];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int it = tid + 256 * k;
      const int xk = it >> 5, c = (it >> 3) & 3, q = it & 7;
      *(float4*)(Us + (xk * CK + c) * 32 + 4 * q) = ru[k];
    }
  };

  f32x16 acc[4][2];   // [xi_w][plane]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

  load_regs(0);
  for (int ci0 = 0; ci0 < a.CinPad; ci0 += CK) {
    __syncthreads();
    store_regs();
    __syncthreads();
    if (ci0 + CK < a.CinPad) load_regs(ci0 + CK);
#pragma unroll
    for (int xw = 0; xw < 4; ++xw) {
      const int xi = wv * 4 + xw;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk) {
          const float av = Us[((xi * 3 + kd) * CK + 2 * kk + half) * 32 + l31];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const float bv = Vs[((xi * CK + 2 * kk + half) * PD + p + kd) * 32 + l31];
            acc[xw][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[xw][p], 0, 0, 0);
          }
        }
    }
  }

  // inverse transform: along W inside the wave, along H across the 4 waves through LDS (one plane per round)
  float* Ts = smem;   // [xi_h 4][b 2][16 regs][64 lanes] = 32 KB
  const int ti = l31 >> 3, tj = l31 & 7;
  const int oa = wv & 1, rbase = (wv >> 1) * 8;
  const int oh = h0 + 2 * ti + oa, ow = w0 + 2 * tj;
  const bool wpair = (a.W & 1) == 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][p][r], m1 = acc[1][p][r], m2 = acc[2][p][r], m3 = acc[3][p][r];
      Ts[((wv * 2 + 0) * 16 + r) * 64 + lane] = m0 + m1 + m2;
      Ts[((wv * 2 + 1) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
    __syncthreads();
    const int od = d0 + p;
    if (od < a.D && oh < a.H && ow < a.W) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = rbase + rr;
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        float yv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float t0 = Ts[((0 * 2 + b) * 16 + r) * 64 + lane], t1 = Ts[((1 * 2 + b) * 16 + r) * 64 + lane];
          const float t2 = Ts[((2 * 2 + b) * 16 + r) * 64 + lane], t3 = Ts[((3 * 2 + b) * 16 + r) * 64 + lane];
          yv[b] = oa == 0 ? (t0 + t1 + t2) : (t1 - t2 - t3);
        }
        if (co < a.Cout) {
          const long idx = ((((long)n * a.CoutTotal + a.co_off + co) * a.D + od) * a.H + oh) * a.W + ow;
          if (wpair) {
            float2 o;
            o.x = wepilogue(a, yv[0], co, idx);
            o.y = wepilogue(a, yv[1], co, idx + 1);
            *(float2*)(a.y + idx) = o;
          } else {
            a.y[idx] = wepilogue(a, yv[0], co, idx);
            if (ow + 1 < a.W) a.y[idx + 1] = wepilogue(a, yv[1], co, idx + 1);
          }
        }
      }
    }
  }
}

// U[(xi*3 + kd)][a][b] = sum_{kh,kw} G[xi_h][kh] G[xi_w][kw] g(a, b, kd, kh, kw), zero padded to (Apad, 32);
// g(a,b,tap) = src_ab ? src[a][b][tap] : src[b][a][tap], taps reversed when flip (stride-1 backward-data).
__global__ void wino_prep_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int Bn, int Apad,
                                 int src_ab, int flip, int Btotal, int b_off) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int total = 48 * Apad * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int bi = idx & 31, ai = (idx >> 5) % Apad, xk = idx / (32 * Apad);
    const int xi = xk / 3, kd = xk % 3, xh = xi >> 2, xw = xi & 3;
    float v = 0.f;
    if (ai < A && bi < Bn) {
      const float* g = src_ab ? src + ((long)ai * Btotal + b_off + bi) * 27 : src + ((long)(b_off + bi) * A + ai) * 27;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (kd * 3 + kh) * 3 + kw;
          v += G[xh][kh] * G[xw][kw] * g[flip ? 26 - tap : tap];
        }
    }
    dst[idx] = v;
  }
}
}  // namespace

extern "C" int dca_conv3d_wino_prep_weight(const float* w, float* ug, int A, int B, int Apad, int src_ab, int flip,
                                           int Btotal, int b_off, hipStream_t stream) {
  DCA_REQUIRE(w && ug && A > 0 && B > 0 && B <= 32 && Apad >= A && Apad % 4 == 0 && b_off >= 0 && b_off + B <= Btotal);
  const int total = 48 * Apad * 32;
  hipLaunchKernelGGL(wino_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, w, ug, A, B, Apad, src_ab, flip,
                     Btotal, b_off);
  return dca_launch_status();
}

extern "C" int dca_conv3d_wino_forward(const float* x, const float* ug, float* y, const float* scale,
                                       const float* shift, const float* res_pre, const float* res_post, float slope,
                                       int N, int Cin, int Cout, int CinPad, int CoutTotal, int co_off, int D, int H,
                                       int W, hipStream_t stream) {
  DCA_REQUIRE(x && ug && y && N > 0 && Cin > 0 && Cout > 0 && Cout <= 32 && CinPad >= Cin && CinPad % 4 == 0);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr) && co_off >= 0 && co_off + Cout <= CoutTotal);
  DCA_REQUIRE((((uintptr_t)ug | (uintptr_t)y) & 15) == 0);
  DCA_REQUIRE((long)Cin * D * H * W * 4 < 0x7ffffff0L);   // 32-bit byte offsets inside one sample
  WinoArgs a;
  a.x = x; a.ug = ug; a.y = y; a.scale = scale; a.shift = shift; a.res_pre = res_pre; a.res_post = res_post;
  a.slope = slope; a.N = N; a.Cin = Cin; a.Cout = Cout; a.CinPad = CinPad; a.CoutTotal = CoutTotal; a.co_off = co_off;
  a.D = D; a.H = H; a.W = W;
  a.nTD = cdiv(D, 2); a.nTH = cdiv(H, 8); a.nTW = cdiv(W, 16);
  const long nblk = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(nblk < (1L << 31));
  const size_t lds = (size_t)(V_ELEMS + U_ELEMS) * 4;
  hipLaunchKernelGGL(wino_conv3_kernel, dim3((unsigned)nblk), dim3(256), lds, stream, a);
  return dca_launch_status();
}
