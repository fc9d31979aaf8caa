// Fused training head: trilinear x s up-sampling of a coarse logit volume + softmax over the disparity axis +
// disparity regression, without materialising the (B, s*n, s*hc, s*wc) volume.
//
// Replaces (reference, models/gwcnet_dca_g.py:261-264):
//     out = F.upsample(prob_volume3, scale_factor=(8,8,8), mode='trilinear'); pred = F.softmax(out.squeeze(1), 1)
//     pred_dca3 = disparity_regression(pred, maxdisp)
// which moves 4 x 401 MB per pair at 544x960 / D=192.  Trilinear interpolation is separable: per full-res pixel
// the n coarse planes are interpolated bilinearly once (n registers), then the s*n fine logits are linear
// blends of neighbouring planes, consumed by an online softmax.  Backward: stage 1 recomputes the softmax per
// pixel and reduces the gradient along the disparity axis back to the n coarse planes (B,n,H,W scratch);
// stage 2 gathers the (2s)^2 spatial footprint of every coarse voxel.  No atomics, deterministic.
#include "dca_common.h"
#include "../../include/dca_hip.h"

namespace {
constexpr int NMAX = 64;  // planes; the per-thread columns live in dynamic LDS (n * 256 floats)

__device__ __forceinline__ void lin_src(int o, float rs, int n, int& i0, int& i1, float& l0, float& l1) {
  float src = rs * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

// bilinear interpolation of the n coarse planes at fine pixel (y, x) into this thread's LDS column
template <int S>
__device__ __forceinline__ void planes_at(const float* __restrict__ c, int n, int hc, int wc, int y, int x,
                                          float* col) {
  int h0, h1, w0, w1;
  float lh0, lh1, lw0, lw1;
  lin_src(y, 1.0f / S, hc, h0, h1, lh0, lh1);
  lin_src(x, 1.0f / S, wc, w0, w1, lw0, lw1);
  const long o00 = (long)h0 * wc + w0, o01 = (long)h0 * wc + w1, o10 = (long)h1 * wc + w0, o11 = (long)h1 * wc + w1;
  const long plane = (long)hc * wc;
  for (int k = 0; k < n; ++k) {
    const float* p = c + k * plane;
    col[k * 256] = lh0 * (lw0 * p[o00] + lw1 * p[o01]) + lh1 * (lw0 * p[o10] + lw1 * p[o11]);
  }
}

struct Soft { float m, s, sk; };
__device__ __forceinline__ void soft_push(Soft& st, float u, float k) {
  const float mn = fmaxf(st.m, u);
  const float corr = expf(st.m - mn), e = expf(u - mn);
  st.s = st.s * corr + e;
  st.sk = st.sk * corr + e * k;
  st.m = mn;
}

// online softmax statistics over the S*n fine logits of one pixel: the first S/2 equal plane 0, then S blends per
// coarse interval with weights (j+0.5)/S (exactly ATen's align_corners=False source index), the last S/2 equal
// plane n-1.
template <int S>
__device__ __forceinline__ Soft pixel_stats(const float* col, int n) {
  Soft st{-INFINITY, 0.f, 0.f};
  float a = col[0];
  for (int j = 0; j < S / 2; ++j) soft_push(st, a, (float)j);
  for (int kc = 0; kc < n - 1; ++kc) {
    const float b = col[(kc + 1) * 256];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const float l1 = ((float)j + 0.5f) * (1.0f / S);
      soft_push(st, (1.f - l1) * a + l1 * b, (float)(S * kc + S / 2 + j));
    }
    a = b;
  }
  for (int j = 0; j < S / 2; ++j) soft_push(st, a, (float)(S * (n - 1) + S / 2 + j));
  return st;
}

template <int S>
__global__ __launch_bounds__(256) void up_softargmin_fwd_kernel(const float* __restrict__ logits,
                                                                float* __restrict__ disp, int n, int hc, int wc) {
  extern __shared__ float vs[];
  const int W = S * wc, H = S * hc;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  float* col = vs + threadIdx.x;
  planes_at<S>(logits + (long)b * n * hc * wc, n, hc, wc, y, x, col);
  const Soft st = pixel_stats<S>(col, n);
  disp[((long)b * H + y) * W + x] = st.sk / st.s;
}

// stage 1: g1[b, kc, y, x] = sum_k dU_k * (weight of coarse plane kc in fine logit k), dU_k = p_k (k - disp) g
template <int S>
__global__ __launch_bounds__(256) void up_softargmin_bwd1_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ gdisp,
                                                                 float* __restrict__ g1, int n, int hc, int wc) {
  extern __shared__ float vs[];
  const int W = S * wc, H = S * hc;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  float* col = vs + threadIdx.x;
  planes_at<S>(logits + (long)b * n * hc * wc, n, hc, wc, y, x, col);
  const Soft st = pixel_stats<S>(col, n);
  const float inv = 1.f / st.s, dsp = st.sk * inv, g = gdisp[((long)b * H + y) * W + x], m = st.m;
  const long plane = (long)H * W;
  float* o = g1 + (long)b * n * plane + (long)y * W + x;
  float a = col[0], acc_a = 0.f;
  for (int j = 0; j < S / 2; ++j) acc_a += expf(a - m) * inv * ((float)j - dsp) * g;
  for (int kc = 0; kc < n - 1; ++kc) {
    const float bb = col[(kc + 1) * 256];
    float acc_b = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const float l1 = ((float)j + 0.5f) * (1.0f / S), l0 = 1.f - l1;
      const float gu = expf(l0 * a + l1 * bb - m) * inv * ((float)(S * kc + S / 2 + j) - dsp) * g;
      acc_a += l0 * gu;
      acc_b += l1 * gu;
    }
    o[kc * plane] = acc_a;
    a = bb;
    acc_a = acc_b;
  }
  for (int j = 0; j < S / 2; ++j) acc_a += expf(a - m) * inv * ((float)(S * (n - 1) + S / 2 + j) - dsp) * g;
  o[(n - 1) * plane] = acc_a;
}

// weight with which fine index o contributes to coarse index i along one spatial dim
__device__ __forceinline__ float lin_w(int o, int i, float rs, int n) {
  int i0, i1;
  float l0, l1;
  lin_src(o, rs, n, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// stage 2: glogits[b,kc,yc,xc] = sum_{y,x} wh(y,yc) ww(x,xc) g1[b,kc,y,x]
template <int S>
__global__ void up_softargmin_bwd2_kernel(const float* __restrict__ g1, float* __restrict__ glogits, int B, int n,
                                          int hc, int wc) {
  const int W = S * wc, H = S * hc;
  const long total = (long)B * n * hc * wc;
  const float rs = 1.0f / S;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int xc = idx % wc;
    long t = idx / wc;
    const int yc = t % hc;
    const long bk = t / hc;
    const float* p = g1 + bk * (long)H * W;
    // the footprint of a coarse sample is at most 2S fine positions per axis; the S-dependent weights are evaluated once per
    // axis (2 x 2S lin_w calls) instead of once per footprint element ((2S)^2 + 2S calls: the kernel was bound by them)
    const int yb = S * yc - (S + 1) / 2, xb = S * xc - (S + 1) / 2;
    float wy[2 * S], wx[2 * S];
#pragma unroll
    for (int j = 0; j < 2 * S; ++j) {
      const int y = yb + j, x = xb + j;
      wy[j] = ((unsigned)y < (unsigned)H) ? lin_w(y, yc, rs, hc) : 0.f;
      wx[j] = ((unsigned)x < (unsigned)W) ? lin_w(x, xc, rs, wc) : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int jy = 0; jy < 2 * S; ++jy) {
      if (wy[jy] == 0.f) continue;
      const float* row = p + (long)(yb + jy) * W + xb;
      float r = 0.f;
#pragma unroll
      for (int jx = 0; jx < 2 * S; ++jx)
        if (wx[jx] != 0.f) r += wx[jx] * row[jx];
      acc += wy[jy] * r;
    }
    glogits[idx] = acc;
  }
}
}  // namespace

extern "C" int dca_up_softargmin_fwd(const float* logits, float* disp, int B, int n, int hc, int wc, int scale,
                                     hipStream_t stream) {
  DCA_REQUIRE(logits && disp && B > 0 && n >= 2 && n <= NMAX && hc > 0 && wc > 0);
  DCA_REQUIRE((scale == 2 || scale == 4 || scale == 8) && scale * hc <= 65535 && B <= 65535);
  const dim3 grid(cdiv(scale * wc, 256), scale * hc, B);
  const size_t lds = (size_t)n * 256 * 4;
#define LAUNCH(S) hipLaunchKernelGGL(up_softargmin_fwd_kernel<S>, grid, dim3(256), lds, stream, logits, disp, n, hc, wc)
  if (scale == 8) LAUNCH(8);
  else if (scale == 4) LAUNCH(4);
  else LAUNCH(2);
#undef LAUNCH
  return dca_launch_status();
}

extern "C" int dca_up_softargmin_bwd(const float* logits, const float* gdisp, float* g1, float* glogits, int B, int n,
                                     int hc, int wc, int scale, hipStream_t stream) {
  DCA_REQUIRE(logits && gdisp && g1 && glogits && B > 0 && n >= 2 && n <= NMAX && hc > 0 && wc > 0);
  DCA_REQUIRE((scale == 2 || scale == 4 || scale == 8) && scale * hc <= 65535 && B <= 65535);
  const dim3 grid(cdiv(scale * wc, 256), scale * hc, B);
  const long total = (long)B * n * hc * wc;
  const int g2 = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  const size_t lds = (size_t)n * 256 * 4;
#define LAUNCH(S)                                                                                                   \
  hipLaunchKernelGGL(up_softargmin_bwd1_kernel<S>, grid, dim3(256), lds, stream, logits, gdisp, g1, n, hc, wc);      \
  hipLaunchKernelGGL(up_softargmin_bwd2_kernel<S>, dim3(g2), dim3(256), 0, stream, g1, glogits, B, n, hc, wc)
  if (scale == 8) { LAUNCH(8); }
  else if (scale == 4) { LAUNCH(4); }
  else { LAUNCH(2); }
#undef LAUNCH
  return dca_launch_status();
}
