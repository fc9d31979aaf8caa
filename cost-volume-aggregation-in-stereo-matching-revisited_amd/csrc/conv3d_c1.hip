// 3x3x3 convolution with ONE output channel (the logit heads), gfx950.
//
// Replaces nn.Conv3d(32, 1, kernel_size=3, padding=1, bias=False): `classif{0..3}.2`
// (models/gwcnet_dca_g.py:154-168) and `cva.classify.2` (models/augment/cva.py:51-53), forward, backward-data and
// weight gradient.  With a single output channel a direct MFMA tile would waste 31 of its 32 rows; instead the 27
// taps become the GEMM's output axis (see "tap expansion" below) for forward and weight gradient, and the
// backward-data pass is a VALU kernel (dy halo tile in LDS, 4 consecutive W positions per thread).
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
      for (int j = 0; j < 4; ++j)      // three chained FMAs (the sum-then-add form costs a multiply and an add more per tap row)
        acc[j] = fmaf(wv.z, g[r9][j + 2], fmaf(wv.y, g[r9][j + 1], fmaf(wv.x, g[r9][j], acc[j])));
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

// ------------------------------------------------------------------------------------------ tap expansion
// (A direct fp32 forward kernel -- the mirror image of the backward-data kernel, channels through LDS four at a time -- was
// built and measured in round 3: 0.80 ms per batch-4 head against 0.47 ms for tap GEMM + gather: 27 LDS reads per channel and
// thread; withdrawn.)
// The forward and the weight gradient are expressed as 1x1x1 GEMMs over a 27-channel "tap" axis so they run on
// the matrix-core kernels of conv3d_mfma.hip / conv3d_wgrad.hip:
//   forward : T[t][v'] = sum_ci w[ci][t] x[ci][v']  (conv1_mfma_kernel, 32 -> 27)      y[v] = sum_t T[t][v + t - 1]
//   wgrad   : G[t][v'] = dy[v' - (t - 1)]                                              dW[ci][t] = sum_v' x[ci][v'] G[t][v']
// (t - 1) is the 3D tap offset (kd-1, kh-1, kw-1); out-of-volume positions contribute zero.
// grid (voxel tiles, N): the sample base is wave-uniform, so each of the 27 shifted taps is one hardware-predicated
// buffer load (one descriptor per kd slab of 9 tap planes) -- 27 independent loads in flight, no branches.
__global__ __launch_bounds__(256) void c1_gather_kernel(const float* __restrict__ T, float* __restrict__ y, int D, int H,
                                                        int W) {
  const long DHW = (long)D * H * W;
  const long n = blockIdx.y;
  const float* p = T + n * 27 * DHW;
  const long end = min(DHW, ((long)blockIdx.x + 1) * 1024);
  for (long idx = (long)blockIdx.x * 1024 + threadIdx.x; idx < end; idx += 256) {
    const int w = idx % W;
    const int t = (int)(idx / W);
    const int h = t % H, d = t / H;
    float s = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int dd = d + kd - 1;
      const __amdgpu_buffer_rsrc_t r = dca_rsrc(p + kd * 9 * DHW, 9 * DHW * 4);  // one descriptor per kd slab
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = w + kw - 1;
          const int ok = (int)((unsigned)dd < (unsigned)D) & (int)((unsigned)hh < (unsigned)H) & (int)((unsigned)ww < (unsigned)W);
          s += dca_bload1(r, ((kh * 3 + kw) * (int)DHW + (dd * H + hh) * W + ww) * 4, ok);
        }
      }
    }
    y[n * DHW + idx] = s;
  }
}

// G[n][tap][v] = dy[n][v - offset(tap)] (0 outside): grid (voxel tiles, 27 taps, N), 32-bit index arithmetic only
// (the former flat version spent its time in 64-bit div/mod chains: 135 us for a 170 MB store stream).
__global__ __launch_bounds__(256) void c1_expand_kernel(const float* __restrict__ dy, float* __restrict__ G, int D, int H,
                                                        int W) {
  const int DHW = D * H * W, tap = blockIdx.y;
  const long n = blockIdx.z;
  const int od = tap / 9 - 1, oh = (tap / 3) % 3 - 1, ow = tap % 3 - 1;
  const __amdgpu_buffer_rsrc_t dr = dca_rsrc(dy + n * DHW, (long)DHW * 4);
  float* out = G + (n * 27 + tap) * (long)DHW;
  const int end = min(DHW, ((int)blockIdx.x + 1) * 1024);
  for (int idx = blockIdx.x * 1024 + threadIdx.x; idx < end; idx += 256) {
    const int w = idx % W, t = idx / W, h = t % H, d = t / H;
    const int dd = d - od, hh = h - oh, ww = w - ow;
    const int ok = (int)((unsigned)dd < (unsigned)D) & (int)((unsigned)hh < (unsigned)H) & (int)((unsigned)ww < (unsigned)W);
    out[idx] = dca_bload1(dr, ((dd * H + hh) * W + ww) * 4, ok);
  }
}

// the same with four consecutive w per thread and one 16-byte store (W % 4 == 0, 16-byte aligned G): the scalar form wrote its
// 680 MB (batch 4, 48x136x240) at 2.6 TB/s
__global__ __launch_bounds__(256) void c1_expand4_kernel(const float* __restrict__ dy, float* __restrict__ G, int D, int H,
                                                         int W) {
  const int DHW = D * H * W, Q = DHW >> 2, WQ = W >> 2, tap = blockIdx.y;
  const long n = blockIdx.z;
  const int od = tap / 9 - 1, oh = (tap / 3) % 3 - 1, ow = tap % 3 - 1;
  const __amdgpu_buffer_rsrc_t dr = dca_rsrc(dy + n * DHW, (long)DHW * 4);
  float* out = G + (n * 27 + tap) * (long)DHW;
  const int end = min(Q, ((int)blockIdx.x + 1) * 1024);
  for (int q = blockIdx.x * 1024 + threadIdx.x; q < end; q += 256) {
    const int wq = q % WQ, t = q / WQ, h = t % H, d = t / H;
    const int dd = d - od, hh = h - oh, w0 = 4 * wq - ow;
    const int okr = (int)((unsigned)dd < (unsigned)D) & (int)((unsigned)hh < (unsigned)H);
    const int base = ((dd * H + hh) * W + w0) * 4;
    float v[4];
    if (ow == 0) {                               // wave-uniform branch: aligned row, one 16-byte load
      const float4 r = dca_bload4(dr, base, okr);
      v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = dca_bload1(dr, base + 4 * j, okr & (int)((unsigned)(w0 + j) < (unsigned)W));
    }
    *(float4*)(out + 4 * (long)q) = make_float4(v[0], v[1], v[2], v[3]);
  }
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

extern "C" int dca_conv3d_c1_bwd_data(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W,
                                      hipStream_t stream) {
  DCA_REQUIRE(dy && w && dx && N > 0 && C > 0 && C <= 256 && D > 0 && H > 0 && W > 0);
  C1Args a;
  a.x = nullptr; a.w = w; a.dy = dy; a.out = dx;
  DCA_REQUIRE(fill_args(a, N, C, D, H, W, dy, dx, nullptr) == 0);
  hipLaunchKernelGGL(c1_bwd_data_kernel, dim3(a.ntiles), dim3(256), (size_t)C * 36 * 4, stream, a);
  return dca_launch_status();
}

extern "C" int dca_conv3d_c1_gather(const float* T, float* y, int N, int D, int H, int W, hipStream_t stream) {
  DCA_REQUIRE(T && y && N > 0 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE(9L * D * H * W * 4 < 0x7ffffff0L && N <= 65535);  // 32-bit byte offsets inside one kd slab
  hipLaunchKernelGGL(c1_gather_kernel, dim3(cdiv((long)D * H * W, 1024), N), dim3(256), 0, stream, T, y, D, H, W);
  return dca_launch_status();
}

extern "C" int dca_conv3d_c1_expand(const float* dy, float* G, int N, int D, int H, int W, hipStream_t stream) {
  DCA_REQUIRE(dy && G && N > 0 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE((long)D * H * W * 4 < 0x7ffffff0L && N <= 65535);
  if (W % 4 == 0 && ((((uintptr_t)dy) | ((uintptr_t)G)) & 15) == 0) {
    hipLaunchKernelGGL(c1_expand4_kernel, dim3(cdiv((long)D * H * W / 4, 1024), 27, N), dim3(256), 0, stream, dy, G, D, H, W);
    return dca_launch_status();
  }
  hipLaunchKernelGGL(c1_expand_kernel, dim3(cdiv((long)D * H * W, 1024), 27, N), dim3(256), 0, stream, dy, G, D, H, W);
  return dca_launch_status();
}
