// Weight gradients of the 3D convolutions on the fp32 matrix cores (gfx950).
//
// Replaces the autograd weight-gradient of nn.Conv3d / nn.ConvTranspose3d on the aggregation path
// (reference modules: models/submodule.py:121-124, models/augment/cva.py:13-55).
//
//   dW[cy][cx][k] = sum_{n, o} dy[n][cy][o] * x[n][cx][S*o - 1 + k]        (3x3x3, pad 1, stride S)
//
// As an MFMA contraction the reduction (K) axis is the voxel index: D[cy][cx] += A[cy][v] B[v][cx]
// with v_mfma_f32_32x32x2_f32 (two voxels per instruction).  A workgroup stages a dy tile
// [32 cy][TD*TH*32 voxels] and the matching x halo tile [32 cx][...] in LDS with ODD channel
// pitches (lanes index channels, so an odd pitch makes the operand reads conflict free), the 27
// taps are split over the 4 waves (7 accumulators each), and persistent workgroups keep their
// accumulators over many tiles, writing one partial slab each; a small second kernel sums the
// slabs in a fixed order (deterministic, no float atomics).
//
// The transposed convolution's weight gradient is the same reduction with the roles of the two
// tensors exchanged (dWt[ci][co][k] = sum_m x[ci][m] dy[co][2m-1+k]); see dca_hip.h.
#include "dca_common.h"
#include "../../include/dca_hip.h"

struct WgArgs {
  const float* x;
  const float* dy;
  float* part;
  int N, Cx, Cy, Di, Hi, Wi, Do, Ho, Wo;
  int nTD, nTH, nTW, ntiles, nCxT;
  int vecx, vecy, small_offsets;
};

// Tile = TD x TH x TW output voxels (256 for stride 1, 64 for stride 2); TW = 16 is chosen when it pads W less
// (W = 240: 15 x 16 exactly instead of 8 x 32 = 256).
template <int S, int TW>
__global__ __launch_bounds__(256, 1) void wgrad3_kernel(WgArgs a) {
  static_assert(TW == 32 || (TW == 16 && S == 1), "tile width");
  constexpr int TD = (S == 1) ? 2 : 1, TH = (S == 1) ? (TW == 32 ? 4 : 8) : 2;
  constexpr int NR = TD * TH, NV = NR * TW;
  constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int IWP = ((3 + IW + 3) / 4) * 4;
  constexpr int XP = ID * IH * IWP + 1;  // odd
  constexpr int YP = NV + 1;             // odd
  static_assert((XP & 1) && (YP & 1), "odd pitches");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + 32 * XP;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int cyT = blockIdx.y / a.nCxT, cxT = blockIdx.y % a.nCxT;
  const int cy0 = cyT * 32, cx0 = cxT * 32;

  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int toff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    int tap = wv + 4 * j;
    if (tap > 26) tap = 26;
    toff[j] = ((tap / 9) * IH + (tap / 3) % 3) * IWP + tap % 3;
  }
  const float* xb = xs + l31 * XP + 3 + half * S;
  const float* yb = ys + l31 * YP + half;

  // Software pipeline (aligned case): the NEXT tile's global loads are issued into registers before the MFMA
  // loop of the current tile and written to LDS after it, so HBM/L2 latency hides under the matrix work
  // (one wave per SIMD here: 512 registers are available, the prefetch set is ~130).
  constexpr int ROWS = 32 * ID * IH;
  constexpr int QPR = TW * S / 4, NH = (S == 1) ? 2 : 1, YQ = TW / 4;
  constexpr int KX = (ROWS * QPR + 255) / 256, KH = (ROWS * NH + 255) / 256, KY = (32 * NR * YQ + 255) / 256;
  float4 rx[KX], ry[KY];
  float rh[KH];
  const bool pipelined = a.vecx && a.vecy && a.small_offsets;

  // prefetch loads are hardware-predicated buffer loads (dca_common.h): straight-line code, masked elements -> 0
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x, (long)a.N * a.Cx * a.Di * a.Hi * a.Wi * 4);
  const __amdgpu_buffer_rsrc_t yr_ = dca_rsrc(a.dy, (long)a.N * a.Cy * a.Do * a.Ho * a.Wo * 4);
  auto load_regs = [&](int tile) __attribute__((always_inline)) {
    int t = tile;
    const int tw = t % a.nTW; t /= a.nTW;
    const int th = t % a.nTH; t /= a.nTH;
    const int td = t % a.nTD;
    const int n = t / a.nTD;
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int di0 = d0 * S - 1, hi0 = h0 * S - 1, wi0 = w0 * S - 1;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 256 * k;
      const int row = it / QPR, q = it % QPR;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = cx0 + c, di = di0 + id, hi = hi0 + ih, wi = wi0 + 1 + 4 * q;
      const int ok = (int)(it < ROWS * QPR) & (int)(ci < a.Cx) & (int)((unsigned)di < (unsigned)a.Di) &
                     (int)((unsigned)hi < (unsigned)a.Hi) & (int)(wi < a.Wi);
      rx[k] = dca_bload4(xr, ((((n * a.Cx + ci) * a.Di + di) * a.Hi + hi) * a.Wi + wi) * 4, ok);
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int it = tid + 256 * k;
      const int row = it / NH, j = (it % NH) ? (IW - 1) : 0;
      const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
      const int ci = cx0 + c, di = di0 + id, hi = hi0 + ih, wi = wi0 + j;
      const int ok = (int)(it < ROWS * NH) & (int)(ci < a.Cx) & (int)((unsigned)di < (unsigned)a.Di) &
                     (int)((unsigned)hi < (unsigned)a.Hi) & (int)((unsigned)wi < (unsigned)a.Wi);
      rh[k] = dca_bload1(xr, ((((n * a.Cx + ci) * a.Di + di) * a.Hi + hi) * a.Wi + wi) * 4, ok);
    }
#pragma unroll
    for (int k = 0; k < KY; ++k) {
      const int it = tid + 256 * k;
      const int q = it % YQ, row = (it / YQ) % NR, c = it / (YQ * NR);
      const int co = cy0 + c, d = d0 + row / TH, h = h0 + row % TH, w = w0 + 4 * q;
      const int ok = (int)(it < 32 * NR * YQ) & (int)(co < a.Cy) & (int)(d < a.Do) & (int)(h < a.Ho) & (int)(w < a.Wo);
      ry[k] = dca_bload4(yr_, ((((n * a.Cy + co) * a.Do + d) * a.Ho + h) * a.Wo + w) * 4, ok);
    }
  };
  auto store_regs = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 256 * k;
      if (it < ROWS * QPR) {
        const int row = it / QPR, q = it % QPR;
        const int c = row / (ID * IH), rem = row % (ID * IH);
        float* dst = xs + c * XP + rem * IWP + 4 + 4 * q;
        dst[0] = rx[k].x; dst[1] = rx[k].y; dst[2] = rx[k].z; dst[3] = rx[k].w;
      }
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int it = tid + 256 * k;
      if (it < ROWS * NH) {
        const int row = it / NH, j = (it % NH) ? (IW - 1) : 0;
        const int c = row / (ID * IH), rem = row % (ID * IH);
        xs[c * XP + rem * IWP + 3 + j] = rh[k];
      }
    }
#pragma unroll
    for (int k = 0; k < KY; ++k) {
      const int it = tid + 256 * k;
      if (it < 32 * NR * YQ) {
        const int q = it % YQ, row = (it / YQ) % NR, c = it / (YQ * NR);
        float* dst = ys + c * YP + row * TW + 4 * q;
        dst[0] = ry[k].x; dst[1] = ry[k].y; dst[2] = ry[k].z; dst[3] = ry[k].w;
      }
    }
  };

  if (pipelined) load_regs(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    if (pipelined) {
      store_regs();
    } else {
      int t = tile;
      const int tw = t % a.nTW; t /= a.nTW;
      const int th = t % a.nTH; t /= a.nTH;
      const int td = t % a.nTD;
      const int n = t / a.nTD;
      const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
      const int di0 = d0 * S - 1, hi0 = h0 * S - 1, wi0 = w0 * S - 1;
      for (int it = tid; it < ROWS * IW; it += 256) {
        const int row = it / IW, j = it % IW;
        const int c = row / (ID * IH), rem = row % (ID * IH), id = rem / IH, ih = rem % IH;
        const int ci = cx0 + c, di = di0 + id, hi = hi0 + ih, wi = wi0 + j;
        float v = 0.f;
        if (ci < a.Cx && (unsigned)di < (unsigned)a.Di && (unsigned)hi < (unsigned)a.Hi &&
            (unsigned)wi < (unsigned)a.Wi)
          v = a.x[((((long)n * a.Cx + ci) * a.Di + di) * a.Hi + hi) * a.Wi + wi];
        xs[c * XP + rem * IWP + 3 + j] = v;
      }
      for (int it = tid; it < 32 * NV; it += 256) {
        const int wl = it % TW, row = (it / TW) % NR, c = it / NV;
        const int co = cy0 + c, d = d0 + row / TH, h = h0 + row % TH, w = w0 + wl;
        float v = 0.f;
        if (co < a.Cy && d < a.Do && h < a.Ho && w < a.Wo)
          v = a.dy[((((long)n * a.Cy + co) * a.Do + d) * a.Ho + h) * a.Wo + w];
        ys[c * YP + row * TW + wl] = v;
      }
    }
    __syncthreads();
    if (pipelined && tile + (int)gridDim.x < a.ntiles) load_regs(tile + gridDim.x);
    // ---- contraction over the tile's voxels, two per MFMA
#pragma unroll 1
    for (int row = 0; row < NR; ++row) {
      const int dl = row / TH, hl = row % TH;
      const float* xr = xb + ((dl * S) * IH + hl * S) * IWP;
      const float* yr = yb + row * TW;
#pragma unroll 4
      for (int wp = 0; wp < TW / 2; ++wp) {
        const float av = yr[2 * wp];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const float bv = xr[2 * wp * S + toff[j]];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // ---- partial slab: part[((blk*nChanTiles + chanTile)*27 + tap)*1024 + cy*32 + cx]
  const long slab = ((long)blockIdx.x * gridDim.y + blockIdx.y) * 27;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = wv + 4 * j;
    if (tap > 26) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cy = (r & 3) + 8 * (r >> 2) + 4 * half;
      a.part[(slab + tap) * 1024 + cy * 32 + l31] = acc[j][r];
    }
  }
}

// 1x1x1: dW[cy][cx] = sum_v dy[cy][v] x[cx][v]; tile = 256 voxels, each wave contracts 64 of them.  HBM-bound (reads x
// and dy once): two workgroups per CU, and on aligned inputs the next tile's 2 x 32 KB are fetched into registers by
// predicated buffer loads while the current tile's MFMAs run.
// TAPS (the 32 -> 1 logit heads, conv3d_c1.hip): dy is the ONE-channel gradient (N, 1, D, H, W) and "channel" t < 27 of the dy
// operand is the tap-shifted view G[t][v] = dy[v - offset(t)] (zero outside the volume), built while the tile is fetched --
// the 27-plane tensor dca_conv3d_c1_expand materialises (680 MB written and read back per batch-4 head) never exists; dy
// itself (25 MB) is served by the caches.  Pipelined path only (W % 4 == 0: a quad of voxels lies in one row).
template <bool TAPS>
__global__ __launch_bounds__(256, 2) void wgrad1_kernel(WgArgs a) {
  constexpr int NV = 256, P = NV + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + 32 * P;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int cyT = blockIdx.y / a.nCxT, cxT = blockIdx.y % a.nCxT;
  const int cy0 = cyT * 32, cx0 = cxT * 32;
  const long DHW = (long)a.Do * a.Ho * a.Wo;
  const long tiles_per_n = (DHW + NV - 1) / NV;
  const long ntile = (long)a.N * tiles_per_n;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool pipelined = a.vecx && a.small_offsets;   // 32-bit byte offsets inside the whole tensors
  if (pipelined) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x, (long)a.N * a.Cx * DHW * 4);
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.dy, (long)a.N * (TAPS ? 1 : a.Cy) * DHW * 4);
    // TAPS: a thread's dy items are (row r = (od, oh) of the 3 x 3 tap rows, quad q): the aligned quad M of dy at the shifted
    // row plus its left and right neighbours give the quads of the row's three taps (ow = -1, 0, +1): 3 loads per 3 taps
    // (first version: one item per tap, four 4-byte loads for every ow != 0 -- 24 load instructions per thread and tile
    // against 8 for x, 616 us per batch-4 head where x streams in 170)
    float4 rx[8], ry[TAPS ? 3 : 8];
    float rl[TAPS ? 3 : 1], rr[TAPS ? 3 : 1];
    auto load_regs = [&](long tile) __attribute__((always_inline)) {
      const int n = (int)(tile / tiles_per_n);
      const int v0 = (int)((tile % tiles_per_n) * NV);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = tid + 256 * k, q = it & 63, c = it >> 6;
        const int v = v0 + 4 * q, ok = (int)(v < DHW);
        rx[k] = dca_bload4(xr, (int)((((long)n * a.Cx + cx0 + c) * DHW + v) * 4), ok & (int)(cx0 + c < a.Cx));
        if constexpr (!TAPS)
          ry[k] = dca_bload4(yr, (int)((((long)n * a.Cy + cy0 + c) * DHW + v) * 4), ok & (int)(cy0 + c < a.Cy));
      }
      if constexpr (TAPS) {
        const int vq = v0 + 4 * (tid & 63);                      // the thread's quad of voxels: the same for its 3 items
        const int qw = vq % a.Wo, t = vq / a.Wo, qh = t % a.Ho, qd = t / a.Ho;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int r = (tid >> 6) + 4 * k, od = r / 3 - 1, oh = r % 3 - 1;      // wave-uniform
          const int dd = qd - od, hh = qh - oh;
          const int okr = (int)(vq < DHW) & (int)(r < 9) & (int)((unsigned)dd < (unsigned)a.Do) & (int)((unsigned)hh < (unsigned)a.Ho);
          const int base = (int)(((long)n * DHW + ((long)dd * a.Ho + hh) * a.Wo + qw) * 4);
          ry[k] = dca_bload4(yr, base, okr);
          rl[k] = dca_bload1(yr, base - 4, okr & (int)(qw > 0));
          rr[k] = dca_bload1(yr, base + 16, okr & (int)(qw + 4 < a.Wo));
        }
      }
    };
    if constexpr (TAPS) {      // "channels" 27..31 of the dy operand do not exist: zero once, never written afterwards
      for (int i = tid; i < 5 * P; i += 256) ys[27 * P + i] = 0.f;
    }
    if ((long)blockIdx.x < ntile) load_regs(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = tid + 256 * k, q = it & 63, c = it >> 6;
        float* dx = xs + c * P + 4 * q;
        dx[0] = rx[k].x; dx[1] = rx[k].y; dx[2] = rx[k].z; dx[3] = rx[k].w;
        if constexpr (!TAPS) {
          float* dyp = ys + c * P + 4 * q;
          dyp[0] = ry[k].x; dyp[1] = ry[k].y; dyp[2] = ry[k].z; dyp[3] = ry[k].w;
        }
      }
      if constexpr (TAPS) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int r = (tid >> 6) + 4 * k, q = tid & 63;
          if (r < 9) {      // tap 3r (ow = -1): dy[w + 1 ..]; tap 3r + 1: the quad itself; tap 3r + 2 (ow = +1): dy[w - 1 ..]
            float* e0 = ys + (3 * r) * P + 4 * q;
            e0[0] = ry[k].y; e0[1] = ry[k].z; e0[2] = ry[k].w; e0[3] = rr[k];
            e0[P] = ry[k].x; e0[P + 1] = ry[k].y; e0[P + 2] = ry[k].z; e0[P + 3] = ry[k].w;
            e0[2 * P] = rl[k]; e0[2 * P + 1] = ry[k].x; e0[2 * P + 2] = ry[k].y; e0[2 * P + 3] = ry[k].z;
          }
        }
      }
      __syncthreads();
      if (tile + gridDim.x < ntile) load_regs(tile + gridDim.x);
      const float* xr_ = xs + l31 * P + wv * 64 + half;
      const float* yr_ = ys + l31 * P + wv * 64 + half;
#pragma unroll 8
      for (int p = 0; p < 32; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yr_[2 * p], xr_[2 * p], acc, 0, 0, 0);
    }
  } else {
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
      const int n = (int)(tile / tiles_per_n);
      const long v0 = (tile % tiles_per_n) * NV;
      __syncthreads();
      for (int it = tid; it < 32 * NV; it += 256) {
        const int vl = it & 255, c = it >> 8;
        const long v = v0 + vl;
        xs[c * P + vl] = (v < DHW && cx0 + c < a.Cx) ? a.x[((long)n * a.Cx + cx0 + c) * DHW + v] : 0.f;
        ys[c * P + vl] = (v < DHW && cy0 + c < a.Cy) ? a.dy[((long)n * a.Cy + cy0 + c) * DHW + v] : 0.f;
      }
      __syncthreads();
      const float* xr_ = xs + l31 * P + wv * 64 + half;
      const float* yr_ = ys + l31 * P + wv * 64 + half;
#pragma unroll 8
      for (int p = 0; p < 32; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yr_[2 * p], xr_[2 * p], acc, 0, 0, 0);
    }
  }
  // reduce the 4 waves through LDS, then one slab per workgroup
  __syncthreads();
  float* red = smem;  // [4][1024]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wv * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
  __syncthreads();
  const long slab = (long)blockIdx.x * gridDim.y + blockIdx.y;
  for (int i = tid; i < 1024; i += 256)
    a.part[slab * 1024 + i] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
}

// dw[cy*s_cy + cx*s_cx + tap] = sum_blk part[((blk*nCT + ct)*K + tap)*1024 + (cy%32)*32 + cx%32]
// Threads walk the slab in memory order (coalesced); 16 thread groups split the slabs (8 independent loads in flight
// per thread) and meet in LDS, always summing in the same order (deterministic).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                            int nblk, int nCxT, int nCT, int K, int Cy, int Cx,
                                                            long s_cy, long s_cx) {
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long slab_elems = (long)nCT * K * 1024;
  const long idx = (long)blockIdx.x * 64 + e;   // element inside one worker's slab set
  float s = 0.f;
  if (idx < slab_elems) {
    int b = g;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(long)(b + 16 * j) * slab_elems + idx];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; b < nblk; b += 16) s += part[(long)b * slab_elems + idx];
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && idx < slab_elems) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += red[j][e];
    const int cxl = idx & 31, cyl = (idx >> 5) & 31;
    const long t = idx >> 10;
    const int tap = t % K, ct = t / K;
    const int cy = (ct / nCxT) * 32 + cyl, cx = (ct % nCxT) * 32 + cxl;
    if (cy < Cy && cx < Cx) dw[cy * s_cy + cx * s_cx + tap] = v;
  }
}

// shared with conv3d_wgrad_bf16x3.hip (same slab format)
int dca_internal_wgrad_reduce(const float* part, float* dw, int nblk, int nCxT, int nCT, int K, int Cy, int Cx,
                              long s_cy, long s_cx, hipStream_t stream) {
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)nCT * K * 1024, 64)), dim3(1024), 0, stream, part, dw, nblk,
                     nCxT, nCT, K, Cy, Cx, s_cy, s_cx);
  return dca_launch_status();
}

static int wg_workers(int ntiles, int nCT, int per_cu = 1) {
  int w = 256 * per_cu / nCT;
  if (w < 1) w = 1;
  return ntiles < w ? ntiles : w;
}

static int wg_tile_w(int ksize, int stride, int Wo) {
  if (ksize != 3 || stride != 1) return 32;
  return (cdiv(Wo, 16) * 16 < cdiv(Wo, 32) * 32) ? 16 : 32;
}

static void wg_geometry(int ksize, int stride, int N, int Do, int Ho, int Wo, int* nTD, int* nTH, int* nTW,
                        long* ntiles) {
  if (ksize == 1) {
    const long DHW = (long)Do * Ho * Wo;
    *nTD = *nTH = *nTW = 1;
    *ntiles = (long)N * ((DHW + 255) / 256);
    return;
  }
  const int TW = wg_tile_w(ksize, stride, Wo);
  const int TD = stride == 1 ? 2 : 1, TH = stride == 1 ? (TW == 32 ? 4 : 8) : 2;
  *nTD = cdiv(Do, TD); *nTH = cdiv(Ho, TH); *nTW = cdiv(Wo, TW);
  *ntiles = (long)N * *nTD * *nTH * *nTW;
}

// number of floats of scratch `part` needed by dca_conv3d_wgrad for this problem
extern "C" long dca_conv3d_wgrad_workspace(int N, int Cx, int Cy, int Do, int Ho, int Wo, int ksize, int stride) {
  int nTD, nTH, nTW; long ntiles;
  wg_geometry(ksize, stride, N, Do, Ho, Wo, &nTD, &nTH, &nTW, &ntiles);
  const int nCT = cdiv(Cx, 32) * cdiv(Cy, 32);
  const int K = ksize == 1 ? 1 : 27;
  return (long)wg_workers((int)(ntiles < 65535 ? ntiles : 65535), nCT, ksize == 1 ? 2 : 1) * nCT * K * 1024;
}

extern "C" int dca_conv3d_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int Cx, int Cy, int Di,
                                int Hi, int Wi, int Do, int Ho, int Wo, int ksize, int stride, long s_cy, long s_cx,
                                hipStream_t stream) {
  DCA_REQUIRE(x && dy && part && dw && N > 0 && Cx > 0 && Cy > 0 && (ksize == 1 || ksize == 3));
  DCA_REQUIRE(stride == 1 || (stride == 2 && ksize == 3));
  if (ksize == 1) DCA_REQUIRE(Di == Do && Hi == Ho && Wi == Wo);
  else DCA_REQUIRE(Do == (Di + stride - 1) / stride && Ho == (Hi + stride - 1) / stride &&
                   Wo == (Wi + stride - 1) / stride);
  WgArgs a;
  a.x = x; a.dy = dy; a.part = part; a.N = N; a.Cx = Cx; a.Cy = Cy;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  long ntiles;
  wg_geometry(ksize, stride, N, Do, Ho, Wo, &a.nTD, &a.nTH, &a.nTW, &ntiles);
  DCA_REQUIRE(ntiles < (1L << 31));
  a.ntiles = (int)ntiles;
  a.small_offsets = ((long)N * Cx * Di * Hi * Wi * 4 < 0x7ffffff0L) && ((long)N * Cy * Do * Ho * Wo * 4 < 0x7ffffff0L);
  a.nCxT = cdiv(Cx, 32);
  const int nCT = a.nCxT * cdiv(Cy, 32);
  const int K = ksize == 1 ? 1 : 27;
  const int nblk = wg_workers((int)(ntiles < 65535 ? ntiles : 65535), nCT, ksize == 1 ? 2 : 1);
  const dim3 grid(nblk, nCT);
  if (ksize == 1) {
    const long DHW = (long)Do * Ho * Wo;
    a.vecx = a.vecy = (DHW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);
    const size_t lds = (size_t)2 * 32 * 257 * 4;
    hipError_t e = hipFuncSetAttribute((const void*)wgrad1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(wgrad1_kernel<false>, grid, dim3(256), lds, stream, a);
  } else {
    a.vecx = (Wi % 4 == 0) && (((uintptr_t)x & 15) == 0);
    a.vecy = (Wo % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    if (stride == 1 && wg_tile_w(3, 1, Wo) == 16) {
      const size_t lds = (size_t)(32 * (4 * 10 * 24 + 1) + 32 * 257) * 4;
      hipError_t e = hipFuncSetAttribute((const void*)wgrad3_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((wgrad3_kernel<1, 16>), grid, dim3(256), lds, stream, a);
    } else if (stride == 1) {
      const size_t lds = (size_t)(32 * (4 * 6 * 40 + 1) + 32 * 257) * 4;
      hipError_t e = hipFuncSetAttribute((const void*)wgrad3_kernel<1, 32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((wgrad3_kernel<1, 32>), grid, dim3(256), lds, stream, a);
    } else {
      const size_t lds = (size_t)(32 * (3 * 5 * 68 + 1) + 32 * 65) * 4;
      hipError_t e = hipFuncSetAttribute((const void*)wgrad3_kernel<2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((wgrad3_kernel<2, 32>), grid, dim3(256), lds, stream, a);
    }
  }
  int st = dca_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)nCT * K * 1024, 64)), dim3(1024), 0, stream, part, dw, nblk,
                     a.nCxT, nCT, K, Cy, Cx, s_cy, s_cx);
  return dca_launch_status();
}

// Weight gradient of the 32 -> 1 logit heads (nn.Conv3d(C, 1, 3, padding=1, bias=False): `classif*.2`
// models/gwcnet_dca_g.py:154-168, `classify.2` models/augment/cva.py:51-53) without the tap-expanded gradient tensor:
// dw[ci * 27 + tap] = sum_{n, v} x[n][ci][v] dy[n][0][v - offset(tap)].  x (N, C, D, H, W), dy (N, 1, D, H, W), 16-byte aligned,
// W % 4 == 0, N*C*D*H*W*4 < 2^31 (hipErrorInvalidValue otherwise: callers use dca_conv3d_c1_expand + dca_conv3d_wgrad);
// part = dca_conv3d_wgrad_workspace(N, C, 27, D, H, W, 1, 1) floats.
extern "C" int dca_conv3d_c1_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int C, int D, int H,
                                   int W, hipStream_t stream) {
  DCA_REQUIRE(x && dy && part && dw && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && W % 4 == 0);
  DCA_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && (long)N * C * D * H * W * 4 < 0x7ffffff0L);
  WgArgs a;
  a.x = x; a.dy = dy; a.part = part; a.N = N; a.Cx = C; a.Cy = 27;
  a.Di = a.Do = D; a.Hi = a.Ho = H; a.Wi = a.Wo = W;
  long ntiles;
  wg_geometry(1, 1, N, D, H, W, &a.nTD, &a.nTH, &a.nTW, &ntiles);
  DCA_REQUIRE(ntiles < (1L << 31));
  a.ntiles = (int)ntiles;
  a.small_offsets = 1; a.vecx = a.vecy = 1;
  a.nCxT = cdiv(C, 32);
  const int nCT = a.nCxT;
  const int nblk = wg_workers((int)(ntiles < 65535 ? ntiles : 65535), nCT, 2);
  const size_t lds = (size_t)2 * 32 * 257 * 4;
  hipError_t e = hipFuncSetAttribute((const void*)wgrad1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(wgrad1_kernel<true>, dim3(nblk, nCT), dim3(256), lds, stream, a);
  int st = dca_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)nCT * 1024, 64)), dim3(1024), 0, stream, part, dw, nblk, a.nCxT,
                     nCT, 1, 27, C, 1L, 27L);
  return dca_launch_status();
}
