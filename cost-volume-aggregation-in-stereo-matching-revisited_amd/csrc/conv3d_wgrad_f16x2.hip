// Weight gradient of the 3x3x3 stride-1 convolution on the f16 matrix pipe with fp32-grade accuracy (the "f16x2" split of
// conv3d_f16x2.hip: every CHANNEL of both operands scaled by its own power of two 2^exps[c] -- the channels are the M and N
// dimensions of this product, so the scales factor out exactly --, split into two f16 terms, the three partial products
// >= 2^-11 accumulated in fp32; entry (co, ci) of the slab written at the end is scaled back by 2^-(yexps[co] + xexps[ci])).
// Either operand may arrive in the packed px2 format (dca_common.h: scaled and split by the BatchNorm kernel that wrote
// it, [voxel][8 channels] words): a thread then loads the 8 words of 8 consecutive voxels and transposes the 8 x 8 block of
// f16 in registers (32 v_perm_b32) into the [channel][8 voxels] LDS words the fragments are read from.
//
//   dW[co][ci][kd,kh,kw] = sum_{n,d,h,w} dy[n][co][d][h][w] * x[n][ci][d+kd-1][h+kh-1][w+kw-1]
//
// is, per tap, a 32 x 32 (co x ci) matrix contracted over the voxels: one v_mfma_f32_32x32x16_f16 takes 16 voxels
// (one W row segment of the tile) as K.  A lane's operand fragment is 8 consecutive voxels of one channel, which is
// contiguous in NCDHW -- so the LDS images are [term][row][k half][channel][8 voxels] and every fragment is one
// ds_read_b128.  The kw = -1 / +1 taps need the same 8 voxels shifted by one element: rather than unaligned LDS reads
// the lane takes its aligned group G, one neighbour dword from its partner lane (v_permlane32_swap: the two k halves
// of a row sit in lanes l and l+32) or from a 2-element edge image, and builds both shifted fragments with five
// v_alignbit_b32 per term.  (Structure of conv3d_wgrad_bf16x3.hip with two terms instead of three.)
//
// Reference operator served: the weight gradient autograd computes for nn.Conv3d(k=3, s=1, p=1) of convbn_3d
// (models/submodule.py:121-124) and the cva blocks (models/augment/cva.py:13-55).
//
// Work decomposition: persistent workgroups of 8 waves (one per CU).  A tile is 2 x 4 x 16 output voxels = 8 K-steps;
// the 8 waves are two groups of four, each group takes 4 K-steps, and inside a group the 27 taps are split 7/7/7/6 over
// the waves (7 x 16 accumulator registers per lane).  The next tile's x halo rows and dy rows are fetched into registers
// during the MFMA phase, split and written to LDS between two barriers.  At the end the two groups are summed through
// LDS and the workgroup writes one slab of partial sums; wgrad_reduce_kernel (conv3d_wgrad.hip) adds the slabs in a
// fixed order: bitwise reproducible, no atomics.
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
// WX2_G8 (A/B): eight tap groups of 3-4 taps, every wave runs all K-steps of the tile (64 accumulator registers, no
// cross-group reduction) instead of two K-step groups x four tap groups of 7 (112); with it the 2 x 8 x 16 tile fits
// without spills (216 registers) -- measured slower: 1308 us (2 x 8 x 16) / 1344 us (2 x 4 x 16) against 1246 us at 32->32,
// 48x136x240, batch 4: every wave then builds the shifted fragments of its tap rows for ALL K-steps
#ifndef WX2_G8
#define WX2_G8 0
#endif
#ifndef WX2_TH
#define WX2_TH (WX2_G8 ? 8 : 4)
#endif
constexpr int TD = 2, TH = WX2_TH, TW = 16;
constexpr int NROW = TD * TH;                       // 8 K-steps (output rows) per tile
constexpr int HD = TD + 2, HH = TH + 2, NHROW = HD * HH;  // 24 halo rows
constexpr int X_TERM = NHROW * 2 * 32 * 16;         // bytes of one term image of x: [hrow][k half][ci][8 f16]
constexpr int XE_TERM = NHROW * 2 * 32 * 4;         // edge dwords: [hrow][side][ci]
constexpr int Y_TERM = NROW * 2 * 32 * 16;
constexpr int X_OFF = 0, XE_OFF = NT * X_TERM, Y_OFF = XE_OFF + NT * XE_TERM;
constexpr int LDS_BYTES_T = Y_OFF + NT * Y_TERM;    // 49152 + 12288 + 16384 = 77824
constexpr int LDS_BYTES = 2 * LDS_BYTES_T;           // two tile images (155648); the end-of-kernel reduction reuses them (114688)
static_assert(LDS_BYTES >= 4 * 7 * 4096 && LDS_BYTES + 128 <= 163840, "LDS budget");
constexpr int NX_ITEMS = NHROW * 2 * 32, KX = NX_ITEMS / 512;    // 1536 -> 3 per thread (8 floats each)
constexpr int NE_ITEMS = NHROW * 2 * 32, KE = NE_ITEMS / 512;    // 1536 -> 3 scalars per thread
constexpr int NY_ITEMS = NROW * 2 * 32, KY = NY_ITEMS / 512;     // 512  -> 1 per thread
static_assert(NX_ITEMS % 512 == 0 && NY_ITEMS % 512 == 0, "staging items");

struct WX2Args {
  const float* x;           // fp32 (N,Cx,D,H,W) or its px2 image
  const float* dy;
  float* part;
  int N, Cx, Cy, D, H, W;
  int nTD, nTH, nTW, nCxT;
  const int* xexps;         // per-channel scale exponents of x / dy (Cx / Cy ints, dca_common.h)
  const int* yexps;
};

__device__ __forceinline__ void split2(float v, int e, _Float16& h, _Float16& l) {
  const float u = ldexpf(v, e);   // exact; scaled maximum < 2^15
  h = (_Float16)u;
  l = (_Float16)(u - (float)h);   // the residual is exact in fp32
}

// 8 x 8 transpose of f16: in[v] = the 8 channels of voxel v (one px2 word), out[c] = the 8 voxels of channel c
__device__ __forceinline__ void transpose8x8(const u32x4v (&in)[8], u32x4v (&out)[8]) {
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[2 * d][i] = __builtin_amdgcn_perm(in[2 * i + 1][d], in[2 * i][d], 0x05040100u);       // low halves: channel 2d
      out[2 * d + 1][i] = __builtin_amdgcn_perm(in[2 * i + 1][d], in[2 * i][d], 0x07060302u);   // high halves: channel 2d+1
    }
}

constexpr int NXU = NT * NHROW * 2 * 4;   // packed x: (term, halo row, k half, channel group) units of 8 words = 384 (threads 0-383)
constexpr int NYU = NT * NROW * 2 * 4;    // packed dy: 128 units (threads 384-511)
static_assert(NXU + NYU == 512, "one packed unit per thread");

template <bool XP, bool YP>
__global__ __launch_bounds__(512) void wgrad3_f16x2_kernel(WX2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
#if WX2_G8
  const int grp = 0, wq = wv;
  constexpr int NACC = 4, NGRP = 1;
#else
  const int grp = wv >> 2, wq = wv & 3;
  constexpr int NACC = 7, NGRP = 2;
#endif
  const int ct = blockIdx.y, cy0 = (ct / a.nCxT) * 32, cx0 = (ct % a.nCxT) * 32;

  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;

  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int cstride = a.D * a.H * a.W;
  const long xsample = (long)a.Cx * cstride, ysample = (long)a.Cy * cstride;
  // a thread's fp32 staging items all belong to channel (tid & 31) of the block: one exponent per operand
  const int xe_t = (cx0 + (tid & 31) < a.Cx) ? dca_coherent_loadi(a.xexps + cx0 + (tid & 31)) : 0;
  const int ye_t = (cy0 + (tid & 31) < a.Cy) ? dca_coherent_loadi(a.yexps + cy0 + (tid & 31)) : 0;

  // staging items: channel fastest (conflict-free LDS writes), then k half / side, then row
  float4 rx[XP ? 1 : KX][2], ry[YP ? 1 : KY][2];
  float re[XP ? 1 : KE];
  u32x4v pu[(XP || YP) ? 8 : 1], pe;      // packed operand: the 8 words of this thread's unit, the word of its edge voxel
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    const int tw = tile % a.nTW; tile /= a.nTW;
    const int th = tile % a.nTH; tile /= a.nTH;
    const int td = tile % a.nTD;
    n = tile / a.nTD;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
  };
  // Staging of a tile, in steps that ride between the MFMAs of the PREVIOUS tile (round 3: the loads issued as one burst
  // and the split / transpose + LDS stores between two barriers left the matrix pipe idle for a third of a tile):
  //   load step  s = 0 .. NLOAD-1:  one global load (packed: word s of the thread's 8-voxel unit, then its edge word;
  //                                 fp32: x quads, x edge voxels, dy quads)
  //   store step q = 0 .. NSTORE-1: [split /] transpose and LDS stores of a quarter unit (packed) or of one item (fp32)
  // into the tile image that is not being read: ONE barrier per tile.
  constexpr int NLOAD = 11, NSTORE = 7;
  auto load_step = [&](int s, __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t yr, int d0, int h0, int w0, int on) __attribute__((always_inline)) {
    if constexpr (XP) {
      if (s < 9 && tid < NXU) {    // unit (term, hrow, k half, group): words of voxels w0 + 8 hf + 0..7; + the edge word of (term, hrow, side, group)
        const int g = tid & 3, hf = (tid >> 2) & 1, hrow = (tid >> 3) % NHROW, term = tid / (8 * NHROW);
        const int d = d0 - 1 + hrow / HH, h = h0 - 1 + hrow % HH, w = w0 + 8 * hf, cg = (cx0 >> 3) + g;
        const int ok = (int)(cg * 8 < a.Cx) & (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) & on;
        const int off = term * (a.Cx * cstride * 2) + (cg * cstride + (d * a.H + h) * a.W + w) * 16;
        if (s < 8) {
          pu[s] = __builtin_bit_cast(u32x4v, dca_bload4(xr, off + 16 * s, ok & (int)(w + s < a.W)));
        } else {
          const int we = hf ? w0 + TW : w0 - 1;     // side = hf
          pe = __builtin_bit_cast(u32x4v, dca_bload4(xr, off + (we - w) * 16, ok & (int)((unsigned)we < (unsigned)a.W)));
        }
      }
    } else {
      if (s < 6) {
        const int k = s >> 1, it = tid + 512 * k, c = it & 31, hf = (it >> 5) & 1, hrow = it >> 6;
        const int d = d0 - 1 + hrow / HH, h = h0 - 1 + hrow % HH, w = w0 + 8 * hf + 4 * (s & 1);
        const int ok = (int)(cx0 + c < a.Cx) & (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) & on;
        rx[k][s & 1] = dca_bload4(xr, ((cx0 + c) * cstride + (d * a.H + h) * a.W + w) * 4, ok & (int)(w + 3 < a.W));   // W % 4 == 0
      } else if (s < 9) {
        const int k = s - 6, it = tid + 512 * k, c = it & 31, side = (it >> 5) & 1, hrow = it >> 6;
        const int d = d0 - 1 + hrow / HH, h = h0 - 1 + hrow % HH, w = side ? w0 + TW : w0 - 1;
        const int ok = (int)(cx0 + c < a.Cx) & (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) &
                       (int)((unsigned)w < (unsigned)a.W) & on;
        re[k] = dca_bload1(xr, ((cx0 + c) * cstride + (d * a.H + h) * a.W + w) * 4, ok);
      }
    }
    if constexpr (YP) {
      if (s < 8 && tid >= NXU) {
        const int u = tid - NXU, g = u & 3, hf = (u >> 2) & 1, row = (u >> 3) % NROW, term = u / (8 * NROW);
        const int d = d0 + row / TH, h = h0 + row % TH, w = w0 + 8 * hf, cg = (cy0 >> 3) + g;
        const int ok = (int)(cg * 8 < a.Cy) & (int)(d < a.D) & (int)(h < a.H) & on;
        const int off = term * (a.Cy * cstride * 2) + (cg * cstride + (d * a.H + h) * a.W + w) * 16;
        pu[s] = __builtin_bit_cast(u32x4v, dca_bload4(yr, off + 16 * s, ok & (int)(w + s < a.W)));
      }
    } else {
      if (s >= 9) {
        const int it = tid, c = it & 31, hf = (it >> 5) & 1, row = it >> 6;
        const int d = d0 + row / TH, h = h0 + row % TH, w = w0 + 8 * hf + 4 * (s - 9);
        const int ok = (int)(cy0 + c < a.Cy) & (int)(d < a.D) & (int)(h < a.H) & on;
        ry[0][s - 9] = dca_bload4(yr, ((cy0 + c) * cstride + (d * a.H + h) * a.W + w) * 4, ok & (int)(w + 3 < a.W));
      }
    }
  };
  auto split_store8 = [&](const float4& p, const float4& q, int sc, char* base, int term_stride, int off) __attribute__((always_inline)) {
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
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
  // channels 2q, 2q+1 of the unit in pu[]: the 8 voxels of each as one LDS word
  auto transpose_store = [&](int q, char* dst) __attribute__((always_inline)) {
    u32x4v lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[i] = __builtin_amdgcn_perm(pu[2 * i + 1][q], pu[2 * i][q], 0x05040100u);       // low halves: channel 2q
      hi[i] = __builtin_amdgcn_perm(pu[2 * i + 1][q], pu[2 * i][q], 0x07060302u);       // high halves: channel 2q+1
    }
    *(u32x4v*)(dst + (2 * q) * 16) = lo;
    *(u32x4v*)(dst + (2 * q + 1) * 16) = hi;
  };
  auto store_step = [&](int q, char* img) __attribute__((always_inline)) {
    if constexpr (XP) {
      if (tid < NXU) {
        const int g = tid & 3, hf = (tid >> 2) & 1, hrow = (tid >> 3) % NHROW, term = tid / (8 * NHROW);
        if (q < 4) {
          transpose_store(q, img + X_OFF + term * X_TERM + ((hrow * 2 + hf) * 32 + g * 8) * 16);      // [hrow][k half][ci][8 voxels]
        } else if (q == 4) {
          // edge image [hrow][side][ci] dwords: left edge (side 0) in the HIGH half, right edge (side 1) in the LOW half
          u32x4v e0, e1;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned lo = pe[d] & 0xffffu, hi = pe[d] >> 16;      // channels 2d, 2d+1 of the group
            const unsigned a0 = hf ? lo : lo << 16, a1 = hf ? hi : hi << 16;
            if (d < 2) { e0[2 * d] = a0; e0[2 * d + 1] = a1; } else { e1[2 * d - 4] = a0; e1[2 * d - 3] = a1; }
          }
          char* de = img + XE_OFF + term * XE_TERM + ((hrow * 2 + hf) * 32 + g * 8) * 4;
          *(u32x4v*)de = e0;
          *(u32x4v*)(de + 16) = e1;
        }
      }
    } else {
      if (q < 3) {
        split_store8(rx[q][0], rx[q][1], xe_t, img + X_OFF, X_TERM, (tid + 512 * q) * 16);
      } else if (q < 6) {
        const int k = q - 3, it = tid + 512 * k, side = (it >> 5) & 1;
        _Float16 h, l;
        split2(re[k], xe_t, h, l);
        // left edge (side 0) sits in the HIGH half of its dword, right edge (side 1) in the LOW half (see shifts below)
        const unsigned sh = side ? 0 : 16;
        *(unsigned*)(img + XE_OFF + it * 4) = (unsigned)__builtin_bit_cast(unsigned short, h) << sh;
        *(unsigned*)(img + XE_OFF + XE_TERM + it * 4) = (unsigned)__builtin_bit_cast(unsigned short, l) << sh;
      }
    }
    if constexpr (YP) {
      if (q < 4 && tid >= NXU) {
        const int u = tid - NXU, g = u & 3, hf = (u >> 2) & 1, row = (u >> 3) % NROW, term = u / (8 * NROW);
        transpose_store(q, img + Y_OFF + term * Y_TERM + ((row * 2 + hf) * 32 + g * 8) * 16);
      }
    } else {
      if (q == 6) split_store8(ry[0][0], ry[0][1], ye_t, img + Y_OFF, Y_TERM, tid * 16);
    }
  };

  // The MFMA phase of one tile for tap group WQ (taps 7*WQ .. 7*WQ+6, < 27), reading tile image `img`; the next tile's
  // staging steps follow the MFMAs of a tap at compile-time positions: loads behind taps 0 .. 10 of the wave's 24-28 (tap,
  // K-step) slots, stores into `nimg` behind slots 16 .. 22.
  auto mfma_tile = [&](auto WQC, const char* img, char* nimg, int on, __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t yr,
                       int nd0, int nh0, int nw0) __attribute__((always_inline)) {
    constexpr int WQ = decltype(WQC)::value;
    constexpr int TAP0 = 7 * WQ, TAP1 = (TAP0 + 7 < 27) ? TAP0 + 7 : 27;
    constexpr int R0 = TAP0 / 3, R1 = (TAP1 - 1) / 3;  // (kd, kh) rows this wave touches
    int slot = 0;    // a compile-time constant after unrolling
#pragma unroll
    for (int i = 0; i < NROW / NGRP; ++i) {
      const int row = grp * (NROW / NGRP) + i, dl = row / TH, hl = row % TH;
      f16x8 ay[NT];
#pragma unroll
      for (int term = 0; term < NT; ++term)
        ay[term] = *(const f16x8*)(img + Y_OFF + term * Y_TERM + ((row * 2 + half) * 32 + l31) * 16);
#pragma unroll
      for (int rr = R0; rr <= R1; ++rr) {
        const int kd = rr / 3, kh = rr % 3;
        const int hrow = (dl + kd) * HH + hl + kh;
        u32x4v g[NT];
        unsigned e[NT];
#pragma unroll
        for (int term = 0; term < NT; ++term) {
          g[term] = *(const u32x4v*)(img + X_OFF + term * X_TERM + ((hrow * 2 + half) * 32 + l31) * 16);
          e[term] = *(const unsigned*)(img + XE_OFF + term * XE_TERM + ((hrow * 2 + half) * 32 + l31) * 4);
        }
        // fragments for kw = 0 (g itself), kw = -1 (s[0..3]) and kw = +1 (s[1..4])
        u32x4v fm[NT], fp[NT];
#pragma unroll
        for (int term = 0; term < NT; ++term) {
          const auto sw = __builtin_amdgcn_permlane32_swap(g[term][0], g[term][3], false, false);
          const unsigned ld = half ? sw[0] : e[term];   // dword whose HIGH half is the voxel left of g[0]
          const unsigned rd = half ? e[term] : sw[1];   // dword whose LOW half is the voxel right of g[7]
          const unsigned s0 = __builtin_amdgcn_alignbit(g[term][0], ld, 16);
          const unsigned s1 = __builtin_amdgcn_alignbit(g[term][1], g[term][0], 16);
          const unsigned s2 = __builtin_amdgcn_alignbit(g[term][2], g[term][1], 16);
          const unsigned s3 = __builtin_amdgcn_alignbit(g[term][3], g[term][2], 16);
          const unsigned s4 = __builtin_amdgcn_alignbit(rd, g[term][3], 16);
          fm[term] = (u32x4v){s0, s1, s2, s3};
          fp[term] = (u32x4v){s1, s2, s3, s4};
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = rr * 3 + kw;
          if (tap < TAP0 || tap >= TAP1) continue;
          const int j = tap - TAP0;
          f16x8 bx[NT];
#pragma unroll
          for (int term = 0; term < NT; ++term)
            bx[term] = __builtin_bit_cast(f16x8, kw == 0 ? fm[term] : (kw == 1 ? g[term] : fp[term]));
          // smallest terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[0], bx[1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[1], bx[0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay[0], bx[0], acc[j], 0, 0, 0);
          if (slot < NLOAD) load_step(slot, xr, yr, nd0, nh0, nw0, on);
          if (slot >= 16 && slot - 16 < NSTORE) store_step(slot - 16, nimg);
          ++slot;
        }
        __builtin_amdgcn_sched_barrier(0);     // the staging steps stay behind their taps; no fragment reads hoisted across rows
      }
    }
  };

  if (t_begin < t_end) {
    int n, d0, h0, w0;
    decode(t_begin, n, d0, h0, w0);
    {
      const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * xsample, xsample * 4);
      const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.dy + (long)n * ysample, ysample * 4);
#pragma unroll
      for (int s2 = 0; s2 < NLOAD; ++s2) load_step(s2, xr, yr, d0, h0, w0, 1);
#pragma unroll
      for (int q = 0; q < NSTORE; ++q) store_step(q, smem);
    }
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += t_step, buf ^= 1) {
      const bool more = tile + t_step < t_end;
      int nn = n, nd0 = d0, nh0 = h0, nw0 = w0;
      if (more) decode(tile + t_step, nn, nd0, nh0, nw0);
      const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)nn * xsample, xsample * 4);
      const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.dy + (long)nn * ysample, ysample * 4);
      const char* img = smem + buf * LDS_BYTES_T;
      char* nimg = smem + (buf ^ 1) * LDS_BYTES_T;
      const int on = more ? 1 : 0;
      switch (wq) {
        case 0: mfma_tile(std::integral_constant<int, 0>{}, img, nimg, on, xr, yr, nd0, nh0, nw0); break;
        case 1: mfma_tile(std::integral_constant<int, 1>{}, img, nimg, on, xr, yr, nd0, nh0, nw0); break;
        case 2: mfma_tile(std::integral_constant<int, 2>{}, img, nimg, on, xr, yr, nd0, nh0, nw0); break;
        default: mfma_tile(std::integral_constant<int, 3>{}, img, nimg, on, xr, yr, nd0, nh0, nw0); break;
      }
      __syncthreads();  // every wave is done reading this tile's image and writing the next one's
    }
  }


  // scale-back: entry (co, ci) by 2^-(yexps[co] + xexps[ci]); ci = this lane's column, co = 16 rows per lane
  __syncthreads();
  int* ey_lds = (int*)(smem + LDS_BYTES);
  if (tid < 32) ey_lds[tid] = (cy0 + tid < a.Cy) ? dca_coherent_loadi(a.yexps + cy0 + tid) : 0;
  __syncthreads();
  const int xe_l = (cx0 + l31 < a.Cx) ? dca_coherent_loadi(a.xexps + cx0 + l31) : 0;
  int ninv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ninv[r] = -(ey_lds[(r & 3) + 8 * (r >> 2) + 4 * half] + xe_l);
#if WX2_G8
  {  // every wave writes the slab entries of its own taps: part[((blk*nCT + ct)*27 + tap)*1024 + co*32 + ci]
    float* slab = a.part + ((long)blockIdx.x * gridDim.y + ct) * 27 * 1024;
    const int tap0 = 27 * wq / 8, ntap = 27 * (wq + 1) / 8 - tap0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      if (j < ntap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[(tap0 + j) * 1024 + co * 32 + l31] = ldexpf(acc[j][r], ninv[r]);
        }
      }
    }
  }
#else
  // group 1 -> LDS, group 0 adds and writes the slab: part[((blk*nCT + ct)*27 + tap)*1024 + co*32 + ci]
  float* red = (float*)smem;  // [wq 4][j 7][co 32][ci 32]
  if (grp == 1) {
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[((wq * 7 + j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[j][r];
  }
  __syncthreads();
  if (grp == 0) {
    float* slab = a.part + ((long)blockIdx.x * gridDim.y + ct) * 27 * 1024;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int tap = 7 * wq + j;
      if (tap < 27) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[tap * 1024 + co * 32 + l31] = ldexpf(acc[j][r] + red[((wq * 7 + j) * 32 + co) * 32 + l31], ninv[r]);
        }
      }
    }
  }
#endif
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

// floats of scratch `part` dca_conv3d_wgrad_x3 needs
extern "C" long dca_conv3d_wgrad_x2_workspace(int N, int Cx, int Cy, int D, int H, int W) {
  if (N <= 0 || Cx <= 0 || Cy <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const long ntiles = (long)N * cdiv(D, TD) * cdiv(H, TH) * cdiv(W, TW);
  const int nCT = cdiv(Cx, 32) * cdiv(Cy, 32);
  return (long)workers(ntiles, nCT) * nCT * 27 * 1024;
}

// dw[cy*s_cy + cx*s_cx + tap] = sum dy[cy] * x[cx] shifted by the tap (3x3x3, stride 1, pad 1); x (N,Cx,D,H,W),
// dy (N,Cy,D,H,W), each fp32 or (x_packed / dy_packed != 0) its px2 image; xexps / yexps = the per-channel scale exponents of
// x / dy (dca_common.h: from dca_conv3d_x2_prep_weight, dca_cmax_exps or the kernel that wrote the packed image).
// fp32 operands need W % 4 == 0 and 16-byte alignment (callers fall back to dca_conv3d_wgrad otherwise), packed ones C % 8 == 0.
extern "C" int dca_conv3d_wgrad_x2(const void* x, int x_packed, const int* xexps, const void* dy, int dy_packed,
                                   const int* yexps, float* part, float* dw, int N, int Cx, int Cy, int D, int H, int W,
                                   long s_cy, long s_cx, hipStream_t stream) {
  DCA_REQUIRE(x && dy && part && dw && xexps && yexps && N > 0 && Cx > 0 && Cy > 0 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE(((((uintptr_t)x | (uintptr_t)dy) & 15) == 0));
  DCA_REQUIRE((x_packed ? Cx % 8 == 0 : W % 4 == 0) && (dy_packed ? Cy % 8 == 0 : W % 4 == 0));
  DCA_REQUIRE((long)Cx * D * H * W * 4 < 0x7ffffff0L && (long)Cy * D * H * W * 4 < 0x7ffffff0L);
  WX2Args a;
  a.x = (const float*)x; a.dy = (const float*)dy; a.part = part; a.xexps = xexps; a.yexps = yexps;
  a.N = N; a.Cx = Cx; a.Cy = Cy; a.D = D; a.H = H; a.W = W;
  a.nTD = cdiv(D, TD); a.nTH = cdiv(H, TH); a.nTW = cdiv(W, TW); a.nCxT = cdiv(Cx, 32);
  const long ntiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(ntiles < 0x7fffffffL);
  const int nCT = a.nCxT * cdiv(Cy, 32);
  DCA_REQUIRE(nCT <= 65535);
  const int nblk = workers(ntiles, nCT);
  const int lds = LDS_BYTES + 128;     // + the 32 exponents of the block's dy channels (end of kernel)
  auto kern = x_packed ? (dy_packed ? wgrad3_f16x2_kernel<true, true> : wgrad3_f16x2_kernel<true, false>)
                       : (dy_packed ? wgrad3_f16x2_kernel<false, true> : wgrad3_f16x2_kernel<false, false>);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(nblk, nCT), dim3(512), lds, stream, a);
  int st = dca_launch_status();
  if (st) return st;
  return dca_internal_wgrad_reduce(part, dw, nblk, a.nCxT, nCT, 27, Cy, Cx, s_cy, s_cx, stream);
}
