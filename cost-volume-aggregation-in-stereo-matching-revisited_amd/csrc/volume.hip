// Cost-volume builders and soft-argmin for gfx950.
//
// Replaces (reference): build_gwc_volume / groupwise_correlation (models/submodule.py:148-167),
// build_concat_volume (submodule.py:134-145), F.softmax(dim=1) + disparity_regression
// (submodule.py:127-131; call sites models/gwcnet_dca_g.py:237-239, :248-275).
//
// gwc forward: one workgroup per (batch, row y, group g).  The 2*CPG feature rows are staged in
// LDS once; every thread owns 4 consecutive x and walks the disparities with a sliding register
// window over the right features, so each output costs 2 LDS reads and the volume rows are written
// as 16-byte-per-lane coalesced stores (the volume write is the HBM-bound part: 40*D*H*W floats).
// The zero half-plane x < i is produced by the same kernel (no memset pass).
#include "dca_common.h"
#include "../../include/dca_hip.h"

template <int CPG>
__global__ __launch_bounds__(256) void gwc_fwd_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                      float* __restrict__ vol, int B, int C, int H, int W,
                                                      int D, int G, int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ls = smem;
  float* Rs = smem + CPG * W;
  const int y = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  for (int i = tid; i < CPG * W; i += 256) {
    const int c = i / W, x = i % W;
    const long src = (((long)b * C + g * CPG + c) * H + y) * W + x;
    Ls[i] = L[src];
    Rs[i] = R[src];
  }
  __syncthreads();
  constexpr int SEG = 12;
  const int nxq = (W + 3) / 4, nseg = (D + SEG - 1) / SEG;
  const float inv = 1.0f / (float)CPG;
  for (int item = tid; item < nxq * nseg; item += 256) {
    const int xq = item % nxq, seg = item / nxq, x0 = 4 * xq;
    const int ib = seg * SEG, ie = min(D, ib + SEG);
    float l[CPG][4], win[CPG][4];
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 + j, xr = x - ib;
        l[c][j] = (x < W) ? Ls[c * W + x] : 0.f;
        win[c][j] = (xr >= 0 && xr < W) ? Rs[c * W + xr] : 0.f;
      }
    for (int i = ib; i < ie; ++i) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CPG; ++c) s += l[c][j] * win[c][j];
        o[j] = s * inv;
      }
      float* dst = vol + ((((long)b * G + g) * D + i) * H + y) * W + x0;
      if (vec) {
        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x0 + j < W) dst[j] = o[j];
      }
      const int xn = x0 - i - 1;
#pragma unroll
      for (int c = 0; c < CPG; ++c) {
        win[c][3] = win[c][2];
        win[c][2] = win[c][1];
        win[c][1] = win[c][0];
        win[c][0] = (xn >= 0) ? Rs[c * W + xn] : 0.f;
      }
    }
  }
}

// gwc backward: dL[c,x] = 1/CPG sum_{i<=x} gV[i,x] R[c,x-i];
//                            dR[c,x'] = 1/CPG sum_{i<W-x'} gV[i,x'+i] L[c,x'+i].
template <int CPG>
__global__ __launch_bounds__(256) void gwc_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ L,
                                                      const float* __restrict__ R, float* __restrict__ gL,
                                                      float* __restrict__ gR, int B, int C, int H, int W, int D,
                                                      int G, int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gv = smem;
  float* Ls = smem + D * W;
  float* Rs = Ls + CPG * W;
  const int y = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  if (vec) {   // W % 4 == 0, 16-byte aligned tensors: 16-byte loads, four in flight per thread
    const int WQ = W >> 2;
    const float* gbase = gvol + (((long)b * G + g) * D * H + y) * W;       // + d*H*W + x
    const long dstride = (long)H * W;
#pragma unroll 4
    for (int i = tid; i < D * WQ; i += 256) {
      const int d = i / WQ, q = i - d * WQ;
      *(float4*)(gv + d * W + 4 * q) = *(const float4*)(gbase + d * dstride + 4 * q);
    }
    const long cbase = (((long)b * C + g * CPG) * H + y) * W;
    for (int i = tid; i < CPG * WQ; i += 256) {
      const int c = i / WQ, q = i - c * WQ;
      *(float4*)(Ls + c * W + 4 * q) = *(const float4*)(L + cbase + c * dstride + 4 * q);
      *(float4*)(Rs + c * W + 4 * q) = *(const float4*)(R + cbase + c * dstride + 4 * q);
    }
  } else {
    for (int i = tid; i < D * W; i += 256) {
      const int d = i / W, x = i % W;
      gv[i] = gvol[((((long)b * G + g) * D + d) * H + y) * W + x];
    }
    for (int i = tid; i < CPG * W; i += 256) {
      const int c = i / W, x = i % W;
      const long src = (((long)b * C + g * CPG + c) * H + y) * W + x;
      Ls[i] = L[src];
      Rs[i] = R[src];
    }
  }
  __syncthreads();
  const float inv = 1.0f / (float)CPG;
  for (int item = tid; item < CPG * W; item += 256) {
    const int c = item / W, x = item % W;
    float gl = 0.f, gr = 0.f;
    const int nl = min(D - 1, x), nr = min(D - 1, W - 1 - x);
    for (int i = 0; i <= nl; ++i) gl += gv[i * W + x] * Rs[c * W + x - i];
    for (int i = 0; i <= nr; ++i) gr += gv[i * W + x + i] * Ls[c * W + x + i];
    const long dst = (((long)b * C + g * CPG + c) * H + y) * W + x;
    gL[dst] = gl * inv;
    gR[dst] = gr * inv;
  }
}

__global__ void concat_fwd_kernel(const float* __restrict__ L, const float* __restrict__ R, float* __restrict__ vol,
                                  int B, int C, int H, int W, int D) {
  const long total = (long)B * 2 * C * D * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long t = idx / W;
    const int y = t % H; t /= H;
    const int i = t % D; t /= D;
    const int c2 = t % (2 * C);
    const int b = t / (2 * C);
    float v = 0.f;
    if (x >= i) {
      if (c2 < C) v = L[(((long)b * C + c2) * H + y) * W + x];
      else v = R[(((long)b * C + c2 - C) * H + y) * W + x - i];
    }
    vol[idx] = v;
  }
}

__global__ void concat_bwd_kernel(const float* __restrict__ gvol, float* __restrict__ gL, float* __restrict__ gR,
                                  int B, int C, int H, int W, int D) {
  const long total = (long)B * C * H * W;
  const long HW = (long)H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long t = idx / W;
    const int y = t % H; t /= H;
    const int c = t % C;
    const int b = t / C;
    const float* pl = gvol + ((long)b * 2 * C + c) * D * HW + (long)y * W + x;
    const float* pr = gvol + ((long)b * 2 * C + C + c) * D * HW + (long)y * W + x;
    float gl = 0.f, gr = 0.f;
    const int nl = min(D - 1, x), nr = min(D - 1, W - 1 - x);
    for (int i = 0; i <= nl; ++i) gl += pl[i * HW];
    for (int i = 0; i <= nr; ++i) gr += pr[i * HW + i];
    gL[idx] = gl;
    gR[idx] = gr;
  }
}

// ---------------------------------------------------------------------------------------------
// softmax over dim 1 of (B, K, HW) and soft-argmin; one thread per (b, pixel), coalesced over HW.
// mode 0: p = softmax(x)                      mode 1: disp = sum_k k * softmax(x)_k
// mode 2: disp = sum_k k * x_k  (plain disparity_regression on an arbitrary x)
// ---------------------------------------------------------------------------------------------
__global__ void softargmin_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int K, long HW,
                                      int mode) {
  const long total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx % HW;
    const float* xp = x + b * K * HW + p;
    if (mode == 2) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += xp[k * HW] * (float)k;
      out[idx] = s;
      continue;
    }
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, xp[k * HW]);
    float s = 0.f, sk = 0.f;
    for (int k = 0; k < K; ++k) {
      const float e = expf(xp[k * HW] - m);
      s += e;
      sk += e * (float)k;
    }
    if (mode == 1) {
      out[idx] = sk / s;
    } else {
      const float inv = 1.0f / s;
      float* op = out + b * K * HW + p;
      for (int k = 0; k < K; ++k) op[k * HW] = expf(xp[k * HW] - m) * inv;
    }
  }
}

// mode 0: gx_k = p_k (gp_k - sum_j gp_j p_j)   (aux = p, g = gp (B,K,HW))
// mode 1: gx_k = p_k (k - disp) g              (aux = logits x, g = gdisp (B,HW)); p recomputed
// mode 2: gx_k = k * g
__global__ void softargmin_bwd_kernel(const float* __restrict__ aux, const float* __restrict__ g,
                                      float* __restrict__ gx, int B, int K, long HW, int mode) {
  const long total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx % HW;
    const float* ap = aux + b * K * HW + p;
    float* gxp = gx + b * K * HW + p;
    if (mode == 0) {
      const float* gp = g + b * K * HW + p;
      float dot = 0.f;
      for (int k = 0; k < K; ++k) dot += gp[k * HW] * ap[k * HW];
      for (int k = 0; k < K; ++k) gxp[k * HW] = ap[k * HW] * (gp[k * HW] - dot);
    } else if (mode == 1) {
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, ap[k * HW]);
      float s = 0.f, sk = 0.f;
      for (int k = 0; k < K; ++k) {
        const float e = expf(ap[k * HW] - m);
        s += e;
        sk += e * (float)k;
      }
      const float inv = 1.0f / s, disp = sk * inv, gg = g[idx];
      for (int k = 0; k < K; ++k) gxp[k * HW] = expf(ap[k * HW] - m) * inv * ((float)k - disp) * gg;
    } else {
      const float gg = g[idx];
      for (int k = 0; k < K; ++k) gxp[k * HW] = (float)k * gg;
    }
  }
}

static int ew_grid(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 4096 ? (g > 0 ? g : 1) : 4096);
}

template <int CPG>
static int gwc_fwd_launch(const float* L, const float* R, float* vol, int B, int C, int H, int W, int D, int G,
                          hipStream_t s) {
  const size_t lds = (size_t)2 * CPG * W * 4;
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)gwc_fwd_kernel<CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int vec = (W % 4 == 0) && (((uintptr_t)vol & 15) == 0);
  hipLaunchKernelGGL(gwc_fwd_kernel<CPG>, dim3(H, G, B), dim3(256), lds, s, L, R, vol, B, C, H, W, D, G, vec);
  return dca_launch_status();
}
template <int CPG>
static int gwc_bwd_launch(const float* gvol, const float* L, const float* R, float* gL, float* gR, int B, int C,
                          int H, int W, int D, int G, hipStream_t s) {
  const size_t lds = (size_t)(D + 2 * CPG) * W * 4;
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)gwc_bwd_kernel<CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int vec = (W % 4 == 0) && ((((uintptr_t)gvol | (uintptr_t)L | (uintptr_t)R) & 15) == 0);
  hipLaunchKernelGGL(gwc_bwd_kernel<CPG>, dim3(H, G, B), dim3(256), lds, s, gvol, L, R, gL, gR, B, C, H, W, D, G, vec);
  return dca_launch_status();
}

extern "C" int dca_gwc_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W,
                                  int maxdisp, int num_groups, hipStream_t stream) {
  DCA_REQUIRE(ref && tgt && vol && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0 && num_groups > 0);
  DCA_REQUIRE(C % num_groups == 0 && H <= 65535 && num_groups <= 65535 && B <= 65535);
  DCA_REQUIRE((size_t)2 * (C / num_groups) * W * 4 <= 160 * 1024);
  switch (C / num_groups) {
    case 1: return gwc_fwd_launch<1>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 2: return gwc_fwd_launch<2>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 4: return gwc_fwd_launch<4>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 8: return gwc_fwd_launch<8>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 16: return gwc_fwd_launch<16>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int dca_gwc_volume_bwd(const float* gvol, const float* ref, const float* tgt, float* gref, float* gtgt,
                                  int B, int C, int H, int W, int maxdisp, int num_groups, hipStream_t stream) {
  DCA_REQUIRE(gvol && ref && tgt && gref && gtgt && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0);
  DCA_REQUIRE(num_groups > 0 && C % num_groups == 0 && H <= 65535 && num_groups <= 65535 && B <= 65535);
  DCA_REQUIRE((size_t)(maxdisp + 2 * (C / num_groups)) * W * 4 <= 160 * 1024);
  switch (C / num_groups) {
    case 1: return gwc_bwd_launch<1>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 2: return gwc_bwd_launch<2>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 4: return gwc_bwd_launch<4>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 8: return gwc_bwd_launch<8>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 16: return gwc_bwd_launch<16>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int dca_concat_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W,
                                     int maxdisp, hipStream_t stream) {
  DCA_REQUIRE(ref && tgt && vol && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0);
  const long total = (long)B * 2 * C * maxdisp * H * W;
  hipLaunchKernelGGL(concat_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, ref, tgt, vol, B, C, H, W, maxdisp);
  return dca_launch_status();
}

extern "C" int dca_concat_volume_bwd(const float* gvol, float* gref, float* gtgt, int B, int C, int H, int W,
                                     int maxdisp, hipStream_t stream) {
  DCA_REQUIRE(gvol && gref && gtgt && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0);
  hipLaunchKernelGGL(concat_bwd_kernel, dim3(ew_grid((long)B * C * H * W)), dim3(256), 0, stream, gvol, gref, gtgt, B,
                     C, H, W, maxdisp);
  return dca_launch_status();
}

extern "C" int dca_softargmin_fwd(const float* x, float* out, int B, int K, long HW, int mode, hipStream_t stream) {
  DCA_REQUIRE(x && out && B > 0 && K > 0 && HW > 0 && mode >= 0 && mode <= 2);
  hipLaunchKernelGGL(softargmin_fwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, x, out, B, K, HW, mode);
  return dca_launch_status();
}

extern "C" int dca_softargmin_bwd(const float* aux, const float* g, float* gx, int B, int K, long HW, int mode,
                                  hipStream_t stream) {
  DCA_REQUIRE(aux && g && gx && B > 0 && K > 0 && HW > 0 && mode >= 0 && mode <= 2);
  hipLaunchKernelGGL(softargmin_bwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, aux, g, gx, B, K, HW,
                     mode);
  return dca_launch_status();
}
