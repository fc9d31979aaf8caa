// 3x3x3 convolution with ONE output channel (the logit heads), gfx950.
//
// Replaces nn.Conv3d(32, 1, kernel_size=3, padding=1, bias=False): `classif{0..3}.2`
// (models/gwcnet_dca_g.py:154-168) and `cva.classify.2` (models/augment/cva.py:51-53), forward, backward-data and
// weight gradient.  With a single output channel there is no Cin x Cout contraction for the matrix cores (an MFMA
// tile would waste 31 of its 32 rows), so this is a bandwidth/VALU kernel: input halo tiles in LDS, every thread
// owns 4 consecutive W positions (one 16-byte LDS read + 2 scalars per (c, kd, kh) row feed 12 FMAs), 16-byte
// coalesced global loads and stores.
#include "dca_common.h"
#include "../../include/dca_hip.h"

namespace {
constexpr int TD = 4, TH = 8, TW = 32, CK = 4;
constexpr int ID = TD + 2, IH = TH + 2, IWP = 40;

struct C1Args {
  const float* x;   // (N, C, D, H, W)
  const float* w;   // (1, C, 3, 3, 3)
  const float* dy;  // (N, 1, D, H, W)
  float* out;       // fwd: y (N,1,..); bwd-data: dx (N,C,..); wgrad: part [nblk][C][27]
  int N, C, D, H, W;
  int nTD, nTH, nTW, ntiles;
  int vec;
};

// stage `nch` channel planes (starting at channel c0 of sample n, C channels per sample) of a halo tile into LDS
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, float* lds, int n, int C, int c0, int nch,
                                           int D, int H, int W, int d0, int h0, int w0, int vec, int tid) {
  const int rows = nch * ID * IH;
  if (vec) {
    for (int it = tid; it < rows * 8; it += 256) {
      const int row = it >> 3, q = it & 7;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = c0 + c, d = d0 - 1 + id, h = h0 - 1 + ih, wq = w0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ci < C && (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && wq < W)
        v = *(const float4*)(src + ((((long)n * C + ci) * D + d) * H + h) * W + wq);
      *(float4*)(lds + row * IWP + 4 + 4 * q) = v;
    }
    for (int it = tid; it < rows * 2; it += 256) {
      const int row = it >> 1, j = (it & 1) ? 33 : 0;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = c0 + c, d = d0 - 1 + id, h = h0 - 1 + ih, wi = w0 - 1 + j;
      float v = 0.f;
      if (ci < C && (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)wi < (unsigned)W)
        v = src[((((long)n * C + ci) * D + d) * H + h) * W + wi];
      lds[row * IWP + 3 + j] = v;
    }
  } else {
    for (int it = tid; it < rows * 34; it += 256) {
      const int row = it / 34, j = it % 34;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = c0 + c, d = d0 - 1 + id, h = h0 - 1 + ih, wi = w0 - 1 + j;
      float v = 0.f;
      if (ci < C && (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)wi < (unsigned)W)
        v = src[((((long)n * C + ci) * D + d) * H + h) * W + wi];
      lds[row * IWP + 3 + j] = v;
    }
  }
}

__device__ __forceinline__ void tile_coords(const C1Args& a, int tile, int& n, int& d0, int& h0, int& w0) {
  const int tw = tile % a.nTW; tile /= a.nTW;
  const int th = tile % a.nTH; tile /= a.nTH;
  const int td = tile % a.nTD;
  n = tile / a.nTD;
  d0 = td * TD; h0 = th * TH; w0 = tw * TW;
}

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void c1_fwd_kernel(C1Args a) {
  __shared__ __attribute__((aligned(16))) float in_lds[CK * ID * IH * IWP];
  __shared__ __attribute__((aligned(16))) float w_lds[CK * 9 * 4];
  const int tid = threadIdx.x, wq = tid & 7, row = tid >> 3, dl = row >> 3, hl = row & 7;
  int n, d0, h0, w0;
  tile_coords(a, xcd_remap(blockIdx.x, gridDim.x), n, d0, h0, w0);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < a.C; c0 += CK) {
    __syncthreads();
    stage_tile(a.x, in_lds, n, a.C, c0, CK, a.D, a.H, a.W, d0, h0, w0, a.vec, tid);
    if (tid < CK * 27) {
      const int c = tid / 27, t = tid % 27;
      w_lds[(c * 9 + t / 3) * 4 + t % 3] = (c0 + c < a.C) ? a.w[(c0 + c) * 27 + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CK; ++c)
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) {
        const float* p = in_lds + ((c * ID + dl + r9 / 3) * IH + hl + r9 % 3) * IWP + 4 * wq;
        const float xm = p[3], xp = p[8];
        const float4 xv = *(const float4*)(p + 4);
        const float4 wv = *(const float4*)(w_lds + (c * 9 + r9) * 4);
        acc[0] += wv.x * xm + wv.y * xv.x + wv.z * xv.y;
        acc[1] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z;
        acc[2] += wv.x * xv.y + wv.y * xv.z + wv.z * xv.w;
        acc[3] += wv.x * xv.z + wv.y * xv.w + wv.z * xp;
      }
  }
  const int d = d0 + dl, h = h0 + hl, w = w0 + 4 * wq;
  if (d < a.D && h < a.H && w < a.W) {
    float* o = a.out + (((long)n * a.D + d) * a.H + h) * a.W + w;
    if (a.vec) {
      *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (w + j < a.W) o[j] = acc[j];
    }
  }
}

// ------------------------------------------------------------------------------------------ backward-data
// dx[n,ci,v] = sum_k dy[n,0,v+1-k] w[ci][k] = sum_t dy[v+t-1] w[ci][26-t]
__global__ __launch_bounds__(256) void c1_bwd_data_kernel(C1Args a) {
  __shared__ __attribute__((aligned(16))) float dy_lds[ID * IH * IWP];
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // [C][9][4], flipped
  const int tid = threadIdx.x, wq = tid & 7, row = tid >> 3, dl = row >> 3, hl = row & 7;
  int n, d0, h0, w0;
  tile_coords(a, xcd_remap(blockIdx.x, gridDim.x), n, d0, h0, w0);
  stage_tile(a.dy, dy_lds, n, 1, 0, 1, a.D, a.H, a.W, d0, h0, w0, a.vec, tid);
  for (int i = tid; i < a.C * 27; i += 256) {
    const int c = i / 27, t = i % 27;
    w_lds[(c * 9 + t / 3) * 4 + t % 3] = a.w[c * 27 + 26 - t];
  }
  __syncthreads();
  float g[9][6];
#pragma unroll
  for (int r9 = 0; r9 < 9; ++r9) {
    const float* p = dy_lds + ((dl + r9 / 3) * IH + hl + r9 % 3) * IWP + 4 * wq;
    const float4 v = *(const float4*)(p + 4);
    g[r9][0] = p[3]; g[r9][1] = v.x; g[r9][2] = v.y; g[r9][3] = v.z; g[r9][4] = v.w; g[r9][5] = p[8];
  }
  const int d = d0 + dl, h = h0 + hl, w = w0 + 4 * wq;
  const bool ok = d < a.D && h < a.H && w < a.W;
  for (int c = 0; c < a.C; ++c) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r9 = 0; r9 < 9; ++r9) {
      const float4 wv = *(const float4*)(w_lds + (c * 9 + r9) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += wv.x * g[r9][j] + wv.y * g[r9][j + 1] + wv.z * g[r9][j + 2];
    }
    if (ok) {
      float* o = a.out + ((((long)n * a.C + c) * a.D + d) * a.H + h) * a.W + w;
      if (a.vec) {
        *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (w + j < a.W) o[j] = acc[j];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ weight gradient
// dW[ci][t] = sum_{n,v} dy[v] x[ci][v+t-1]; persistent workgroups, 4 channels at a time in registers (4 x 27
// accumulators per thread), one block-level reduction per (worker, channel chunk); partials summed by a second pass.
__global__ __launch_bounds__(256) void c1_wgrad_kernel(C1Args a) {
  __shared__ __attribute__((aligned(16))) float in_lds[CK * ID * IH * IWP];
  __shared__ float red[4][CK * 27];
  const int tid = threadIdx.x, wq = tid & 7, row = tid >> 3, dl = row >> 3, hl = row & 7;
  for (int c0 = 0; c0 < a.C; c0 += CK) {
    float acc[CK][27];
#pragma unroll
    for (int c = 0; c < CK; ++c)
#pragma unroll
      for (int t = 0; t < 27; ++t) acc[c][t] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
      int n, d0, h0, w0;
      tile_coords(a, tile, n, d0, h0, w0);
      __syncthreads();
      stage_tile(a.x, in_lds, n, a.C, c0, CK, a.D, a.H, a.W, d0, h0, w0, a.vec, tid);
      const int d = d0 + dl, h = h0 + hl, w = w0 + 4 * wq;
      float gv[4] = {0.f, 0.f, 0.f, 0.f};
      if (d < a.D && h < a.H) {
        const float* gp = a.dy + (((long)n * a.D + d) * a.H + h) * a.W + w;
        if (a.vec) {
          if (w < a.W) { const float4 t4 = *(const float4*)gp; gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w; }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (w + j < a.W) gv[j] = gp[j];
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CK; ++c)
#pragma unroll
        for (int r9 = 0; r9 < 9; ++r9) {
          const float* p = in_lds + ((c * ID + dl + r9 / 3) * IH + hl + r9 % 3) * IWP + 4 * wq;
          const float4 v = *(const float4*)(p + 4);
          const float xs[6] = {p[3], v.x, v.y, v.z, v.w, p[8]};
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            acc[c][r9 * 3 + kw] += gv[0] * xs[kw] + gv[1] * xs[kw + 1] + gv[2] * xs[kw + 2] + gv[3] * xs[kw + 3];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CK; ++c)
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const float s = wave_sum(acc[c][t]);
        if ((tid & 63) == 0) red[tid >> 6][c * 27 + t] = s;
      }
    __syncthreads();
    if (tid < CK * 27 && c0 + tid / 27 < a.C)
      a.out[((long)blockIdx.x * a.C + c0 + tid / 27) * 27 + tid % 27] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

__global__ void c1_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblk, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(long)b * n + i];
  dw[i] = s;
}

int fill_args(C1Args& a, int N, int C, int D, int H, int W, const void* p0, const void* p1, const void* p2) {
  a.N = N; a.C = C; a.D = D; a.H = H; a.W = W;
  a.nTD = cdiv(D, TD); a.nTH = cdiv(H, TH); a.nTW = cdiv(W, TW);
  const long nt = (long)N * a.nTD * a.nTH * a.nTW;
  if (nt >= (1L << 31)) return 1;
  a.ntiles = (int)nt;
  a.vec = (W % 4 == 0) && ((((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0);
  return 0;
}
}  // namespace

extern "C" int dca_conv3d_c1_fwd(const float* x, const float* w, float* y, int N, int C, int D, int H, int W,
                                 hipStream_t stream) {
  DCA_REQUIRE(x && w && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0);
  C1Args a;
  a.x = x; a.w = w; a.dy = nullptr; a.out = y;
  DCA_REQUIRE(fill_args(a, N, C, D, H, W, x, y, nullptr) == 0);
  hipLaunchKernelGGL(c1_fwd_kernel, dim3(a.ntiles), dim3(256), 0, stream, a);
  return dca_launch_status();
}

extern "C" int dca_conv3d_c1_bwd_data(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W,
                                      hipStream_t stream) {
  DCA_REQUIRE(dy && w && dx && N > 0 && C > 0 && C <= 256 && D > 0 && H > 0 && W > 0);
  C1Args a;
  a.x = nullptr; a.w = w; a.dy = dy; a.out = dx;
  DCA_REQUIRE(fill_args(a, N, C, D, H, W, dy, dx, nullptr) == 0);
  hipLaunchKernelGGL(c1_bwd_data_kernel, dim3(a.ntiles), dim3(256), (size_t)C * 36 * 4, stream, a);
  return dca_launch_status();
}

extern "C" long dca_conv3d_c1_wgrad_workspace(int N, int C, int D, int H, int W) {
  const long nt = (long)N * cdiv(D, TD) * cdiv(H, TH) * cdiv(W, TW);
  const long nblk = nt < 512 ? nt : 512;
  return nblk * C * 27;
}

extern "C" int dca_conv3d_c1_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int C, int D, int H,
                                   int W, hipStream_t stream) {
  DCA_REQUIRE(x && dy && part && dw && N > 0 && C > 0 && D > 0 && H > 0 && W > 0);
  C1Args a;
  a.x = x; a.w = nullptr; a.dy = dy; a.out = part;
  DCA_REQUIRE(fill_args(a, N, C, D, H, W, x, dy, nullptr) == 0);
  const int nblk = a.ntiles < 512 ? a.ntiles : 512;
  hipLaunchKernelGGL(c1_wgrad_kernel, dim3(nblk), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3(cdiv(C * 27, 256)), dim3(256), 0, stream, part, dw, nblk, C * 27);
  return dca_launch_status();
}
