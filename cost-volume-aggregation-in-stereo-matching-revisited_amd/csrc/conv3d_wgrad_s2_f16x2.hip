// Weight gradient of the 3x3x3 STRIDE-2 convolution (and of the transposed convolution, whose weight gradient is the same
// sum with the roles of the two tensors exchanged) on the f16 matrix pipe with fp32-grade accuracy: the "f16x2" split of
// conv3d_f16x2.hip / conv3d_wgrad_f16x2.hip (operands scaled by a power of two from their tensor's max |.|, two f16 terms,
// three MFMA products, fp32 accumulation, the slab scaled back when it is written).
//
//   dW[cy][cx][kd,kh,kw] = sum_{n, o} c[n][cy][o] * f[n][cx][2 o + k - 1]      f: fine (N,Cx,D,H,W),  c: coarse (N,Cy,D/2,H/2,W/2)
//
// Per tap a 32 x 32 (cy x cx) matrix contracted over the COARSE voxels: one v_mfma_f32_32x32x16_f16 takes 16 coarse voxels
// of a W row as K.  Along W the fine voxels of those 16 coarse positions are f[2 wo + kw - 1]: kw = 1 reads the EVEN fine
// voxels, kw = 2 the ODD ones, kw = 0 the odd ones shifted by one position.  So a fine row is staged de-interleaved into an
// even and an odd f16 image [term][fine row][parity][k half][cx][8 voxels] (a lane's fragment = one ds_read_b128) plus the
// one odd voxel left of the row; the kw = 0 fragment is built in registers from the odd fragment, the partner lane's last
// dword (v_permlane32_swap) or that edge voxel, and four v_alignbit_b32 -- the machinery of conv3d_wgrad_f16x2.hip, which
// shifts its one image by +-1.
//
// Reference operators served: the weight gradients autograd computes for `cost_agg.conv1` = nn.Conv3d(32, 64, 3, stride 2,
// padding 1) and `cost_agg.conv3` = nn.ConvTranspose3d(64, 32, 3, stride 2, padding 1, output_padding 1)
// (models/augment/cva.py:16-29).
//
// Work decomposition: persistent workgroups of 8 waves (one per CU).  A tile is 1 x 4 x 16 coarse voxels = 4 K-steps and
// its 3 x 9 fine halo rows of 32 (+1) voxels; the 27 taps are split 3/3/4/3/3/4/3/4 over the eight waves (4 x 16 accumulator
// registers per lane), every wave runs all four K-steps.  The next tile's rows are fetched into registers during the MFMA
// phase, split, de-interleaved and written to LDS between two barriers.  At the end every wave writes its taps of the
// workgroup's slab of partial sums; wgrad_reduce_kernel (conv3d_wgrad.hip) adds the slabs in a fixed order: bitwise
// reproducible, no atomics.
#include "dca_common.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

int dca_internal_wgrad_reduce(const float* part, float* dw, int nblk, int nCxT, int nCT, int K, int Cy, int Cx,
                              long s_cy, long s_cx, hipStream_t stream);  // conv3d_wgrad.hip

// WX2_STAMP (debug build, tools/wx3_stamps.py): `part` is followed by an unsigned long long stamp buffer (the tool
// allocates it) that receives s_memtime stamps of the first 64 tiles of workgroup 0, waves 0 and 3
#ifndef WX2_STAMP
#define WX2_STAMP 0
#endif
// WS2_SPLIT_LOADS: the next tile's loads in two portions, behind the first and the second K-step (batch 4, 32->64 at
// 48x136x240: 842 -> 785 us; all of them in front of the MFMA phase: 908 us)
#ifndef WS2_SPLIT_LOADS
#define WS2_SPLIT_LOADS 1
#endif
#ifndef WX2_LOADS_IN
#define WX2_LOADS_IN 1
#endif
#if WX2_STAMP
#define WX2_MARK(i) do { if (stamp_on && stamp_k < 64) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[((wq == 3) * 64 + stamp_k) * 8 + (i)] = t_; } } while (0)
#else
#define WX2_MARK(i) do { } while (0)
#endif

namespace {

constexpr int NT = 2;                                 // terms per operand
constexpr int TH = 4, TW = 16;                        // coarse tile 1 x 4 x 16
constexpr int NROW = TH;                              // 4 K-steps (coarse rows) per tile
constexpr int FD = 3, FH = 2 * TH + 1, NFROW = FD * FH;   // 27 fine halo rows
constexpr int X_TERM = NFROW * 2 * 2 * 32 * 16;       // bytes of one term image of f: [frow][parity][k half][cx][8 f16]
constexpr int XE_TERM = NFROW * 32 * 4;               // the odd voxel left of the row, one dword per [frow][cx] (HIGH half)
constexpr int Y_TERM = NROW * 2 * 32 * 16;
constexpr int X_OFF = 0, XE_OFF = NT * X_TERM, Y_OFF = XE_OFF + NT * XE_TERM;
constexpr int LDS_BYTES = Y_OFF + NT * Y_TERM;        // 110592 + 6912 + 8192 = 125696
constexpr int NX_ITEMS = NFROW * 2 * 32, KX = (NX_ITEMS + 511) / 512;   // 1728 items of 16 fine voxels -> 4 per thread
constexpr int NE_ITEMS = NFROW * 32, KE = (NE_ITEMS + 511) / 512;       // 864 -> 2
constexpr int NY_ITEMS = NROW * 2 * 32;                                  // 256: threads 0-255
static_assert(NY_ITEMS <= 512, "one coarse item per thread");

struct WS2Args {
  const float* x;
  const float* dy;
  float* part;
  int N, Cx, Cy, D, H, W;        // fine dims
  int Do, Ho, Wo;                // coarse dims
  int nTD, nTH, nTW, nCxT;
  const int* xexps;         // per-channel scale exponents of the fine / coarse operand (Cx / Cy ints, dca_common.h)
  const int* yexps;
};

__device__ __forceinline__ void split2(float v, int s, _Float16& h, _Float16& l) {
  const float u = ldexpf(v, s);   // exact; scaled maximum < 2^15
  h = (_Float16)u;
  l = (_Float16)(u - (float)h);   // the residual is exact in fp32
}

__global__ __launch_bounds__(512) void wgrad3s2_f16x2_kernel(WS2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int wq = wv;     // tap group of the wave: taps 27*wq/8 .. 27*(wq+1)/8 - 1 (3 or 4), all four K-steps of a tile
  const int ct = blockIdx.y, cy0 = (ct / a.nCxT) * 32, cx0 = (ct % a.nCxT) * 32;

  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int cstride = a.D * a.H * a.W, ystride = a.Do * a.Ho * a.Wo;
  const long xsample = (long)a.Cx * cstride, ysample = (long)a.Cy * ystride;
  // a thread's staging items all belong to channel (tid & 31) of the block: one exponent per operand
  const int xs = (cx0 + (tid & 31) < a.Cx) ? dca_coherent_loadi(a.xexps + cx0 + (tid & 31)) : 0;
  const int ys = (cy0 + (tid & 31) < a.Cy) ? dca_coherent_loadi(a.yexps + cy0 + (tid & 31)) : 0;

  // staging items: channel fastest (conflict-free LDS writes), then k half, then fine row
  float4 rx[KX][4], ry[2];
  float re[KE];
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    const int tw = tile % a.nTW; tile /= a.nTW;
    const int th = tile % a.nTH; tile /= a.nTH;
    const int td = tile % a.nTD;
    n = tile / a.nTD;
    d0 = td; h0 = th * TH; w0 = tw * TW;            // coarse coordinates
  };
  // part 0: the first two rounds of fine items, part 1: the rest (fine items, edge voxels, coarse rows), part -1: all
  auto load_tile = [&](int n, int d0, int h0, int w0, int part = -1) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * xsample, xsample * 4);
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.dy + (long)n * ysample, ysample * 4);
#pragma unroll
    for (int k = 0; k < KX; ++k) {   // 16 consecutive fine voxels from 2 w0 + 16 hf: the 8 even and 8 odd ones of a k half
      if ((part == 0 && k >= 2) || (part == 1 && k < 2)) continue;
      const int it = tid + 512 * k, c = it & 31, hf = (it >> 5) & 1, frow = it >> 6;
      const int d = 2 * d0 - 1 + frow / FH, h = 2 * h0 - 1 + frow % FH, w = 2 * w0 + 16 * hf;
      const int ok = (int)(it < NX_ITEMS) & (int)(cx0 + c < a.Cx) & (int)((unsigned)d < (unsigned)a.D) &
                     (int)((unsigned)h < (unsigned)a.H);
      const int off = ((cx0 + c) * cstride + (d * a.H + h) * a.W + w) * 4;
      if (512 * k + (tid & ~63) < NX_ITEMS) {   // wave-uniform: the last item round has work for three waves only
#pragma unroll
        for (int q = 0; q < 4; ++q) rx[k][q] = dca_bload4(xr, off + 16 * q, ok & (int)(w + 4 * q + 3 < a.W));   // W % 4 == 0
      }
    }
    if (part != 0) {
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int it = tid + 512 * k, c = it & 31, frow = it >> 5;
      const int d = 2 * d0 - 1 + frow / FH, h = 2 * h0 - 1 + frow % FH, w = 2 * w0 - 1;
      const int ok = (int)(it < NE_ITEMS) & (int)(cx0 + c < a.Cx) & (int)((unsigned)d < (unsigned)a.D) &
                     (int)((unsigned)h < (unsigned)a.H) & (int)((unsigned)w < (unsigned)a.W);
      if (512 * k + (tid & ~63) < NE_ITEMS) re[k] = dca_bload1(xr, ((cx0 + c) * cstride + (d * a.H + h) * a.W + w) * 4, ok);
    }
    if ((tid & ~63) < NY_ITEMS) {
      const int c = tid & 31, hf = (tid >> 5) & 1, row = tid >> 6;
      const int h = h0 + row, w = w0 + 8 * hf;
      const int ok = (int)(tid < NY_ITEMS) & (int)(cy0 + c < a.Cy) & (int)(h < a.Ho);
      const int off = ((cy0 + c) * ystride + (d0 * a.Ho + h) * a.Wo + w) * 4;
      ry[0] = dca_bload4(yr, off, ok & (int)(w + 3 < a.Wo));        // Wo % 4 == 0
      ry[1] = dca_bload4(yr, off + 16, ok & (int)(w + 7 < a.Wo));
    }
    }
  };
  auto split_store8 = [&](const float (&v)[8], int sc, char* base, int term_stride, int off) __attribute__((always_inline)) {
    f16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 h, l;
      split2(v[j], sc, h, l);
      hv[j] = h; lv[j] = l;
    }
    *(f16x8*)(base + off) = hv;
    *(f16x8*)(base + term_stride + off) = lv;
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const int it = tid + 512 * k, c = it & 31, hf = (it >> 5) & 1, frow = it >> 6;
      if (it < NX_ITEMS) {
        const float ev[8] = {rx[k][0].x, rx[k][0].z, rx[k][1].x, rx[k][1].z, rx[k][2].x, rx[k][2].z, rx[k][3].x, rx[k][3].z};
        const float od[8] = {rx[k][0].y, rx[k][0].w, rx[k][1].y, rx[k][1].w, rx[k][2].y, rx[k][2].w, rx[k][3].y, rx[k][3].w};
        split_store8(ev, xs, smem + X_OFF, X_TERM, (((frow * 2 + 0) * 2 + hf) * 32 + c) * 16);
        split_store8(od, xs, smem + X_OFF, X_TERM, (((frow * 2 + 1) * 2 + hf) * 32 + c) * 16);
      }
    }
    if (tid < NY_ITEMS) {
      const float v[8] = {ry[0].x, ry[0].y, ry[0].z, ry[0].w, ry[1].x, ry[1].y, ry[1].z, ry[1].w};
      split_store8(v, ys, smem + Y_OFF, Y_TERM, tid * 16);
    }
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int it = tid + 512 * k;
      if (it < NE_ITEMS) {
        _Float16 h, l;
        split2(re[k], xs, h, l);
        // the odd voxel left of the row sits in the HIGH half of its dword (see the shift below)
        *(unsigned*)(smem + XE_OFF + it * 4) = (unsigned)__builtin_bit_cast(unsigned short, h) << 16;
        *(unsigned*)(smem + XE_OFF + XE_TERM + it * 4) = (unsigned)__builtin_bit_cast(unsigned short, l) << 16;
      }
    }
  };

  // The MFMA phase of one tile for tap group WQ.  The next tile's global loads are issued BEHIND the first K-step's MFMAs
  // (as in conv3d_wgrad_f16x2.hip).  Eight tap groups of 3-4 taps (4 x 16 accumulator registers) rather than two K-step
  // groups x four tap groups of 7: the 16-voxel staging items of a stride-2 row need 64 registers per thread, which do not
  // fit beside 112 accumulator registers.
  auto mfma_tile = [&](auto WQC, bool more, int next_tile) __attribute__((always_inline)) {
    constexpr int WQ = decltype(WQC)::value;
    constexpr int TAP0 = 27 * WQ / 8, TAP1 = 27 * (WQ + 1) / 8;
    constexpr int R0 = TAP0 / 3, R1 = (TAP1 - 1) / 3;  // (kd, kh) rows this wave touches
#pragma unroll 1
    for (int i = 0; i < NROW; ++i) {
#if WS2_SPLIT_LOADS
      if (WX2_LOADS_IN && (i == 1 || i == 2) && more) {
        int nn, nd0, nh0, nw0;
        decode(next_tile, nn, nd0, nh0, nw0);
        load_tile(nn, nd0, nh0, nw0, i - 1);
      }
#else
      if (WX2_LOADS_IN && i == 1 && more) {
        int nn, nd0, nh0, nw0;
        decode(next_tile, nn, nd0, nh0, nw0);
        load_tile(nn, nd0, nh0, nw0);
      }
#endif
      const int row = i;                             // coarse h row of the tile (one coarse d plane per tile)
      f16x8 ay[NT];
#pragma unroll
      for (int term = 0; term < NT; ++term)
        ay[term] = *(const f16x8*)(smem + Y_OFF + term * Y_TERM + ((row * 2 + half) * 32 + l31) * 16);
#pragma unroll
      for (int rr = R0; rr <= R1; ++rr) {
        const int kd = rr / 3, kh = rr % 3;
        const int frow = kd * FH + 2 * row + kh;
        u32x4v ge[NT], go[NT];
        unsigned e[NT];
#pragma unroll
        for (int term = 0; term < NT; ++term) {
          ge[term] = *(const u32x4v*)(smem + X_OFF + term * X_TERM + (((frow * 2 + 0) * 2 + half) * 32 + l31) * 16);
          go[term] = *(const u32x4v*)(smem + X_OFF + term * X_TERM + (((frow * 2 + 1) * 2 + half) * 32 + l31) * 16);
          e[term] = *(const unsigned*)(smem + XE_OFF + term * XE_TERM + (frow * 32 + l31) * 4);
        }
        // kw = 1: the even voxels, kw = 2: the odd ones, kw = 0: the odd ones one position to the left
        u32x4v fm[NT];
#pragma unroll
        for (int term = 0; term < NT; ++term) {
          const auto sw = __builtin_amdgcn_permlane32_swap(go[term][0], go[term][3], false, false);
          const unsigned ld = half ? sw[0] : e[term];   // dword whose HIGH half is the odd voxel left of go[0]
          const unsigned s0 = __builtin_amdgcn_alignbit(go[term][0], ld, 16);
          const unsigned s1 = __builtin_amdgcn_alignbit(go[term][1], go[term][0], 16);
          const unsigned s2 = __builtin_amdgcn_alignbit(go[term][2], go[term][1], 16);
          const unsigned s3 = __builtin_amdgcn_alignbit(go[term][3], go[term][2], 16);
          fm[term] = (u32x4v){s0, s1, s2, s3};
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = rr * 3 + kw;
          if (tap < TAP0 || tap >= TAP1) continue;
          const int j = tap - TAP0;
          f16x8 bx[NT];
#pragma unroll
          for (int term = 0; term < NT; ++term)
            bx[term] = __builtin_bit_cast(f16x8, kw == 0 ? fm[term] : (kw == 1 ? ge[term] : go[term]));
          // smallest terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[0], bx[1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[1], bx[0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[0], bx[0], acc[j], 0, 0, 0);
        }
      }
    }
  };

#if WX2_STAMP
  unsigned long long* stamps = (unsigned long long*)(a.part + (long)gridDim.x * gridDim.y * 27 * 1024);
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && (wq == 0 || wq == 3);
  int stamp_k = 0;
#endif
  if (t_begin < t_end) {
    int n, d0, h0, w0;
    decode(t_begin, n, d0, h0, w0);
    load_tile(n, d0, h0, w0);
    store_tile();
    __syncthreads();
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += t_step) {
      const bool more = tile + t_step < t_end;
      WX2_MARK(0);
      if (more && !WX2_LOADS_IN) {
        decode(tile + t_step, n, d0, h0, w0);
        load_tile(n, d0, h0, w0);
      }
      WX2_MARK(1);
      switch (wq) {
        case 0: mfma_tile(std::integral_constant<int, 0>{}, more, tile + t_step); break;
        case 1: mfma_tile(std::integral_constant<int, 1>{}, more, tile + t_step); break;
        case 2: mfma_tile(std::integral_constant<int, 2>{}, more, tile + t_step); break;
        case 3: mfma_tile(std::integral_constant<int, 3>{}, more, tile + t_step); break;
        case 4: mfma_tile(std::integral_constant<int, 4>{}, more, tile + t_step); break;
        case 5: mfma_tile(std::integral_constant<int, 5>{}, more, tile + t_step); break;
        case 6: mfma_tile(std::integral_constant<int, 6>{}, more, tile + t_step); break;
        default: mfma_tile(std::integral_constant<int, 7>{}, more, tile + t_step); break;
      }
      WX2_MARK(2);
      __syncthreads();  // every wave is done reading this tile
      WX2_MARK(3);
      if (more) store_tile();
      WX2_MARK(4);
      __syncthreads();
      WX2_MARK(5);
#if WX2_STAMP
      ++stamp_k;
#endif
    }
  }

  // every wave writes the slab entries of its own taps: part[((blk*nCT + ct)*27 + tap)*1024 + co*32 + ci]
  // scale-back: entry (co, ci) by 2^-(yexps[co] + xexps[ci])
  __syncthreads();
  int* ey_lds = (int*)smem;
  if (tid < 32) ey_lds[tid] = (cy0 + tid < a.Cy) ? dca_coherent_loadi(a.yexps + cy0 + tid) : 0;
  __syncthreads();
  const int xe_l = (cx0 + l31 < a.Cx) ? dca_coherent_loadi(a.xexps + cx0 + l31) : 0;
  int ninv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ninv[r] = -(ey_lds[(r & 3) + 8 * (r >> 2) + 4 * half] + xe_l);
  {
    float* slab = a.part + ((long)blockIdx.x * gridDim.y + ct) * 27 * 1024;
    const int tap0 = 27 * wq / 8, ntap = 27 * (wq + 1) / 8 - tap0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < ntap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[(tap0 + j) * 1024 + co * 32 + l31] = ldexpf(acc[j][r], ninv[r]);
        }
      }
    }
  }
}

int workers(long ntiles, int nCT) {
  int ncu = 256;
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      v > 0)
    ncu = v;
  long w = ncu / nCT;
  if (w < 1) w = 1;
  return (int)(ntiles < w ? ntiles : w);
}

}  // namespace

// floats of scratch `part` dca_conv3d_wgrad_s2_x2 needs; D, H, W = FINE dims
extern "C" long dca_conv3d_wgrad_s2_x2_workspace(int N, int Cx, int Cy, int D, int H, int W) {
  if (N <= 0 || Cx <= 0 || Cy <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long ntiles = (long)N * Do * cdiv(Ho, TH) * cdiv(Wo, TW);
  const int nCT = cdiv(Cx, 32) * cdiv(Cy, 32);
  return (long)workers(ntiles, nCT) * nCT * 27 * 1024;
}

// dw[cy*s_cy + cx*s_cx + tap] = sum_{n, o} c[n][cy][o] * f[n][cx][2 o + tap - 1] (3x3x3, stride 2, pad 1); f (N,Cx,D,H,W),
// c (N,Cy,(D+1)/2,(H+1)/2,(W+1)/2); f_exps / c_exps = the operands' per-channel scale exponents (Cx / Cy ints: dca_cmax_exps).  Requires
// W % 4 == 0, (W+1)/2 % 4 == 0 and 16-byte aligned f / c (callers fall back to dca_conv3d_wgrad otherwise).
extern "C" int dca_conv3d_wgrad_s2_x2(const float* f, const int* f_exps, const float* c, const int* c_exps,
                                      float* part, float* dw, int N, int Cx, int Cy, int D, int H, int W, long s_cy,
                                      long s_cx, hipStream_t stream) {
  DCA_REQUIRE(f && c && part && dw && f_exps && c_exps && N > 0 && Cx > 0 && Cy > 0 && D > 0 && H > 0 && W > 0);
  WS2Args a;
  a.x = f; a.dy = c; a.part = part; a.xexps = f_exps; a.yexps = c_exps;
  a.N = N; a.Cx = Cx; a.Cy = Cy; a.D = D; a.H = H; a.W = W;
  a.Do = (D + 1) / 2; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;
  DCA_REQUIRE(W % 4 == 0 && a.Wo % 4 == 0 && ((((uintptr_t)f | (uintptr_t)c) & 15) == 0));
  DCA_REQUIRE((long)Cx * D * H * W * 4 < 0x7ffffff0L && (long)Cy * a.Do * a.Ho * a.Wo * 4 < 0x7ffffff0L);
  a.nTD = a.Do; a.nTH = cdiv(a.Ho, TH); a.nTW = cdiv(a.Wo, TW); a.nCxT = cdiv(Cx, 32);
  const long ntiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(ntiles < 0x7fffffffL);
  const int nCT = a.nCxT * cdiv(Cy, 32);
  DCA_REQUIRE(nCT <= 65535);
  const int nblk = workers(ntiles, nCT);
  hipError_t e = hipFuncSetAttribute((const void*)wgrad3s2_f16x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(wgrad3s2_f16x2_kernel, dim3(nblk, nCT), dim3(512), LDS_BYTES, stream, a);
  int st = dca_launch_status();
  if (st) return st;
  return dca_internal_wgrad_reduce(part, dw, nblk, a.nCxT, nCT, 27, Cy, Cx, s_cy, s_cx, stream);
}
