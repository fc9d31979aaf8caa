// Weight gradient of the 3x3x3 STRIDE-2 convolution (and of the transposed convolution, whose weight gradient is the same
// sum with the roles of the two tensors exchanged) on the f16 matrix pipe with fp32-grade accuracy: the "f16x2" split of
// conv3d_f16x2.hip / conv3d_wgrad_f16x2.hip (every operand channel scaled by its own power of two, two f16 terms, three
// MFMA products, fp32 accumulation, the slab scaled back when it is written).
//
//   dW[cy][cx][kd,kh,kw] = sum_{n, o} c[n][cy][o] * f[n][cx][2 o + k - 1]      f: fine (N,Cx,D,H,W),  c: coarse (N,Cy,D/2,H/2,W/2)
//
// Per tap a 32 x 32 (cy x cx) matrix contracted over the COARSE voxels: one v_mfma_f32_32x32x16_f16 takes 16 coarse voxels
// of a W row as K; along W their fine partners are f[2 wo + kw - 1].  Round 3: both LDS images are [voxel][32 channels] f16
// (64 bytes per voxel) and the MFMA fragments (8 voxels of one channel per lane) are read with gfx950's transposing
// ds_read_b64_tr_b16, which takes the ADDRESS OF EVERY VOXEL ROW from a lane: the stride-2 walk along W and the kw shift
// are address arithmetic.  Round 2 staged a fine row de-interleaved into an even and an odd [channel][8 voxels] image and
// rebuilt the kw = 0 fragments in registers (12.7 vector instructions per MFMA, matrix pipe busy 0.14), and it staged the
// fine tile once per 32-channel block of the coarse tensor (HBM traffic 1.6 x algorithmic): here ONE staged fine tile
// serves both 32-channel blocks of a 64-channel coarse tensor (waves 0-3 / 4-7).
//
// Reference operators served: the weight gradients autograd computes for `cost_agg.conv1` = nn.Conv3d(32, 64, 3, stride 2,
// padding 1) and `cost_agg.conv3` = nn.ConvTranspose3d(64, 32, 3, stride 2, padding 1, output_padding 1)
// (models/augment/cva.py:16-29).
//
// Work decomposition: persistent workgroups of 8 waves (one per CU).  A tile is 1 x 4 x 16 coarse voxels = 4 K-steps and
// its 3 x 9 fine halo rows of 33 voxels; wave (b, q) = (coarse channel block b of the pair, tap group q: taps 7q .. 7q+6) runs
// all four K-steps (7 x 16 accumulator registers per lane).  The next tile's data are fetched into registers during the MFMA
// phase and scaled, split and written to LDS between two barriers.  At the end every wave writes its taps of its block's
// slab of partial sums; wgrad_reduce_kernel (conv3d_wgrad.hip) adds the slabs in a fixed order: bitwise reproducible, no
// atomics.
#include "dca_common.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

int dca_internal_wgrad_reduce(const float* part, float* dw, int nblk, int nCxT, int nCT, int K, int Cy, int Cx,
                              long s_cy, long s_cx, hipStream_t stream);  // conv3d_wgrad.hip

namespace {

constexpr int NT = 2;                                 // terms per operand
constexpr int TH = 4, TW = 16;                        // coarse tile 1 x 4 x 16
constexpr int NROW = TH;                              // 4 K-steps (coarse rows) per tile
constexpr int FD = 3, FH = 2 * TH + 1, NFROW = FD * FH;   // 27 fine halo rows
constexpr int FV = 2 * TW + 1;                        // 33 fine voxels per halo row (index 0 = 2 w0 - 1)
constexpr int VB = 64;                                // bytes per voxel of an image: 32 channels x f16
constexpr int X_TERM = NFROW * FV * VB;               // 57024: [frow][voxel][32 cx]
constexpr int Y_BLK = NROW * TW * VB;                 // 4096:  [row][voxel][32 cy] of one coarse channel block
constexpr int Y_TERM = 2 * Y_BLK;                     // two blocks
constexpr int X_OFF = 0, Y_OFF = NT * X_TERM;
constexpr int LDS_BYTES = Y_OFF + NT * Y_TERM;        // 114048 + 16384 = 130432
constexpr int NXQ = NFROW * 8 * 4;                    // 864 fine quad items (frow, quad, channel group): 4 voxels x 8 channels
constexpr int KXQ = (NXQ + 511) / 512;                // 2 rounds; round 1 has 352 items
constexpr int NXE = NFROW * 4;                        // 108 edge voxels (frow, channel group), fine index 0
constexpr int NYQ = 2 * NROW * 4 * 4;                 // 128 coarse quad items (block, row, quad, channel group)
static_assert(NXQ - 512 + NYQ <= 512, "coarse items fit behind the second round of fine items");

struct WS2Args {
  const float* x;
  const float* dy;
  float* part;
  int N, Cx, Cy, D, H, W;        // fine dims
  int Do, Ho, Wo;                // coarse dims
  int nTD, nTH, nTW, nCxT, nCyP; // nCyP: pairs of 32-channel blocks of the coarse tensor
  const int* xexps;         // per-channel scale exponents of the fine / coarse operand (Cx / Cy ints, dca_common.h)
  const int* yexps;
};

__device__ __forceinline__ void split2(float v, int s, _Float16& h, _Float16& l) {
  const float u = ldexpf(v, s);   // exact; scaled maximum < 2^15
  h = (_Float16)u;
  l = (_Float16)(u - (float)h);   // the residual is exact in fp32
}

// 8 voxels x 1 channel MFMA fragment from a [voxel][32 channels] image: two transposing reads of 4 voxel rows each; the
// lane supplies the address of ITS row (dca_wgrad: consecutive voxels; here every second fine voxel), `step` = bytes between
// the two blocks of four
__device__ __forceinline__ f16x8 tr_frag(const char* p, int step) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + step));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, c);
}

__global__ __launch_bounds__(512) void wgrad3s2_f16x2_kernel(WS2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  // (per-lane selectors on purpose: see conv3d_wgrad_f16x2.hip -- scalar branches give each tap group its own accumulator
  // registers and the copies spill)
  const int blk = wv >> 2, wq = wv & 3;
  const int ct = blockIdx.y, cyp = ct / a.nCxT, cx0 = (ct % a.nCxT) * 32, cy0 = cyp * 64;
  const bool blk_on = cy0 + 32 * blk < a.Cy;          // a coarse tensor with one 32-channel block: waves 4-7 only stage

  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;

  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int cstride = a.D * a.H * a.W, ystride = a.Do * a.Ho * a.Wo;
  const long xsample = (long)a.Cx * cstride, ysample = (long)a.Cy * ystride;

  // transposing read: lane 4q+p of a 16-lane group supplies the address of voxel row q, channels 4p .. 4p+3 of the group's
  // 4-row x 16-channel block; group g = (lane >> 4) & 1 takes channels 16g ..; the wave half takes K elements 8 half ..
  const int q4 = (lane & 15) >> 2, chb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int lane_off_x = (2 * (8 * half + q4)) * VB + chb;      // fine voxel 2 k (+ kw): every second row
  const int lane_off_y = (8 * half + q4) * VB + chb;

  // ---- staging ------------------------------------------------------------------------------------------------------
  // fine quad items (channel group fastest, then quad, then fine row): two rounds of 512 / 352; edge voxels: threads 0-107;
  // coarse quad items: threads 352-479 of the second round.  All of a thread's fine items use channel group tid & 3.
  float4 rq[KXQ][8];
  float re[8];
  // the exponents of the block's 32 fine and 64 coarse channels, in LDS behind the images (read when a tile is split)
  int* ex_lds = (int*)(smem + LDS_BYTES);
  if (tid < 32) ex_lds[tid] = (cx0 + tid < a.Cx) ? dca_coherent_loadi(a.xexps + cx0 + tid) : 0;
  else if (tid < 96) ex_lds[tid] = (cy0 + tid - 32 < a.Cy) ? dca_coherent_loadi(a.yexps + cy0 + tid - 32) : 0;
  __syncthreads();
  const int yu = tid - (NXQ - 512);               // coarse item of this thread (second round), valid for 0 <= yu < NYQ
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    const int tw = tile % a.nTW; tile /= a.nTW;
    const int th = tile % a.nTH; tile /= a.nTH;
    const int td = tile % a.nTD;
    n = tile / a.nTD;
    d0 = td; h0 = th * TH; w0 = tw * TW;            // coarse coordinates
  };
  // part 0: the first round of fine items, part 1: the rest (second round, edge voxels, coarse items), part -1: all
  auto load_tile = [&](int n, int d0, int h0, int w0, int part = -1) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * xsample, xsample * 4);
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.dy + (long)n * ysample, ysample * 4);
#pragma unroll
    for (int k = 0; k < KXQ; ++k) {
      if ((part == 0 && k >= 1) || (part == 1 && k < 1)) continue;
      const int it = tid + 512 * k, cg = it & 3, quad = (it >> 2) & 7, frow = it >> 5;
      const int d = 2 * d0 - 1 + frow / FH, h = 2 * h0 - 1 + frow % FH, w = 2 * w0 + 4 * quad, c0 = cx0 + cg * 8;
      const int ok = (int)(it < NXQ) & (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) & (int)(w + 3 < a.W);
      const int off = (c0 * cstride + (d * a.H + h) * a.W + w) * 4;
      if (512 * k + (tid & ~63) < NXQ) {     // wave-uniform: the second round has work for 5.5 waves
#pragma unroll
        for (int j = 0; j < 8; ++j) rq[k][j] = dca_bload4(xr, off + j * cstride * 4, ok & (int)(c0 + j < a.Cx));
      }
    }
    if (part == 0) return;
    if ((tid & ~63) < NXE + 63) {
      const int cg = tid & 3, frow = tid >> 2, c0 = cx0 + cg * 8;
      const int d = 2 * d0 - 1 + frow / FH, h = 2 * h0 - 1 + frow % FH, w = 2 * w0 - 1;
      const int ok = (int)(tid < NXE) & (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) &
                     (int)((unsigned)w < (unsigned)a.W);
      const int off = (c0 * cstride + (d * a.H + h) * a.W + w) * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) re[j] = dca_bload1(xr, off + j * cstride * 4, ok & (int)(c0 + j < a.Cx));
    }
    if ((unsigned)yu < (unsigned)NYQ) {     // (these threads have no second-round fine item: rq[1] is free)
      const int cg = yu & 3, quad = (yu >> 2) & 3, row = (yu >> 4) & 3, by = yu >> 6;
      const int h = h0 + row, w = w0 + 4 * quad, c0 = cy0 + 32 * by + cg * 8;
      const int ok = (int)(h < a.Ho) & (int)(w + 3 < a.Wo);
      const int off = (c0 * ystride + (d0 * a.Ho + h) * a.Wo + w) * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) rq[1][j] = dca_bload4(yr, off + j * ystride * 4, ok & (int)(c0 + j < a.Cy));
    }
  };
  auto split_word = [&](const float (&v)[8], const int* exp8, char* dst, int term_stride) __attribute__((always_inline)) {
    const int4 e0 = *(const int4*)exp8, e1 = *(const int4*)(exp8 + 4);
    const int ex[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    f16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 h, l;
      split2(v[j], ex[j], h, l);
      hv[j] = h; lv[j] = l;
    }
    *(f16x8*)dst = hv;
    *(f16x8*)(dst + term_stride) = lv;
  };
  auto quad_words = [&](const float4 (&q)[8], const int* ex, char* dst, int term_stride) __attribute__((always_inline)) {
    const float* q0 = (const float*)&q[0];
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) {
      const float v[8] = {q0[vv], q0[4 + vv], q0[8 + vv], q0[12 + vv], q0[16 + vv], q0[20 + vv], q0[24 + vv], q0[28 + vv]};
      split_word(v, ex, dst + vv * VB, term_stride);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KXQ; ++k) {
      const int it = tid + 512 * k, cg = it & 3, quad = (it >> 2) & 7, frow = it >> 5;
      if (it < NXQ) quad_words(rq[k], ex_lds + cg * 8, smem + X_OFF + (frow * FV + 1 + 4 * quad) * VB + cg * 16, X_TERM);
    }
    if (tid < NXE) {
      const int cg = tid & 3, frow = tid >> 2;
      const float v[8] = {re[0], re[1], re[2], re[3], re[4], re[5], re[6], re[7]};
      split_word(v, ex_lds + cg * 8, smem + X_OFF + (frow * FV) * VB + cg * 16, X_TERM);
    }
    if ((unsigned)yu < (unsigned)NYQ) {
      const int cg = yu & 3, quad = (yu >> 2) & 3, row = (yu >> 4) & 3, by = yu >> 6;
      quad_words(rq[1], ex_lds + 32 + 32 * by + cg * 8, smem + Y_OFF + by * Y_BLK + (row * TW + 4 * quad) * VB + cg * 16, Y_TERM);
    }
  };

  // The MFMA phase of one tile for tap group WQ (taps 7*WQ .. 7*WQ+6, < 27), coarse channel block `blk`.
  auto mfma_tile = [&](auto WQC, bool more, int next_tile) __attribute__((always_inline)) {
    constexpr int WQ = decltype(WQC)::value;
    constexpr int TAP0 = 7 * WQ, TAP1 = (TAP0 + 7 < 27) ? TAP0 + 7 : 27, NTAP = TAP1 - TAP0;
#pragma unroll 1
    for (int i = 0; i < NROW; ++i) {
      if ((i == 1 || i == 2) && more) {     // the next tile's loads in two portions behind the first two K-steps
        int nn, nd0, nh0, nw0;
        decode(next_tile, nn, nd0, nh0, nw0);
        load_tile(nn, nd0, nh0, nw0, i - 1);
      }
      if (blk_on) {
        // K-step = coarse row i: coarse fragment from block `blk`; fine rows (kd, 2 i + kh), fine voxels 2 k + kw
        const char* yb = smem + Y_OFF + blk * Y_BLK + (i * TW) * VB + lane_off_y;
        const char* xb = smem + X_OFF + ((2 * i) * FV) * VB + lane_off_x;
        f16x8 ay[NT];
#pragma unroll
        for (int term = 0; term < NT; ++term) ay[term] = tr_frag(yb + term * Y_TERM, 4 * VB);
        f16x8 bx[2][NT];
        auto load_b = [&](int tap, int slot) __attribute__((always_inline)) {
          const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
          for (int term = 0; term < NT; ++term)
            bx[slot][term] = tr_frag(xb + term * X_TERM + ((kd * FH + kh) * FV + kw) * VB, 8 * VB);
        };
        load_b(TAP0, 0);
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
          const int cur = j & 1;
          if (j + 1 < NTAP) load_b(TAP0 + j + 1, cur ^ 1);
          // smallest terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[0], bx[cur][1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[1], bx[cur][0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[0], bx[cur][0], acc[j], 0, 0, 0);
          if (j + 1 < NTAP) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  if (t_begin < t_end) {
    int n, d0, h0, w0;
    decode(t_begin, n, d0, h0, w0);
    load_tile(n, d0, h0, w0);
    store_tile();
    __syncthreads();
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += t_step) {
      const bool more = tile + t_step < t_end;
      switch (wq) {
        case 0: mfma_tile(std::integral_constant<int, 0>{}, more, tile + t_step); break;
        case 1: mfma_tile(std::integral_constant<int, 1>{}, more, tile + t_step); break;
        case 2: mfma_tile(std::integral_constant<int, 2>{}, more, tile + t_step); break;
        default: mfma_tile(std::integral_constant<int, 3>{}, more, tile + t_step); break;
      }
      __syncthreads();  // every wave is done reading this tile
      if (more) store_tile();
      __syncthreads();
    }
  }

  // every wave writes the slab entries of its own taps and block: part[((wg*nCT + ct32)*27 + tap)*1024 + co*32 + ci], scaled
  // back by 2^-(yexps[co] + xexps[ci])
  if (blk_on) {
    const int xe_l = ex_lds[l31];
    int ninv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ninv[r] = -(ex_lds[32 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * half] + xe_l);
    const int nCT32 = a.nCxT * ((a.Cy + 31) / 32), ct32 = (2 * cyp + blk) * a.nCxT + (ct % a.nCxT);
    float* slab = a.part + ((long)blockIdx.x * nCT32 + ct32) * 27 * 1024;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int tap = 7 * wq + j;
      if (tap < 27) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[tap * 1024 + co * 32 + l31] = ldexpf(acc[j][r], ninv[r]);
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
  const int nWG = cdiv(Cx, 32) * cdiv(Cy, 64);
  return (long)workers(ntiles, nWG) * cdiv(Cx, 32) * cdiv(Cy, 32) * 27 * 1024;
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
  a.nTD = a.Do; a.nTH = cdiv(a.Ho, TH); a.nTW = cdiv(a.Wo, TW); a.nCxT = cdiv(Cx, 32); a.nCyP = cdiv(Cy, 64);
  const long ntiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(ntiles < 0x7fffffffL);
  const int nWG = a.nCxT * a.nCyP, nCT = a.nCxT * cdiv(Cy, 32);
  DCA_REQUIRE(nWG <= 65535);
  const int nblk = workers(ntiles, nWG);
  const int lds = LDS_BYTES + 96 * 4;      // + the exponents of the block's channels
  hipError_t e = hipFuncSetAttribute((const void*)wgrad3s2_f16x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(wgrad3s2_f16x2_kernel, dim3(nblk, nWG), dim3(512), lds, stream, a);
  int st = dca_launch_status();
  if (st) return st;
  return dca_internal_wgrad_reduce(part, dw, nblk, a.nCxT, nCT, 27, Cy, Cx, s_cy, s_cx, stream);
}
