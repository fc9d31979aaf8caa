// Weight gradient of the 3x3x3 stride-1 convolution on the f16 matrix pipe with fp32-grade accuracy (the "f16x2" split of
// conv3d_f16x2.hip: every CHANNEL of both operands scaled by its own power of two 2^exps[c] -- the channels are the M and N
// dimensions of this product, so the scales factor out exactly --, split into two f16 terms, the three partial products
// >= 2^-11 accumulated in fp32; entry (co, ci) of the slab written at the end is scaled back by 2^-(yexps[co] + xexps[ci])).
//
//   dW[co][ci][kd,kh,kw] = sum_{n,d,h,w} dy[n][co][d][h][w] * x[n][ci][d+kd-1][h+kh-1][w+kw-1]
//
// is, per tap, a 32 x 32 (co x ci) matrix contracted over the voxels: one v_mfma_f32_32x32x16_f16 takes 16 voxels (one W
// row segment of the tile) as K.  Round 3: the LDS images are [voxel][32 channels] f16 (64 bytes per voxel) -- the layout
// of the packed px2 operand format (dca_common.h), so staging a packed operand is a copy of 16-byte words -- and the MFMA
// fragments (8 consecutive voxels of one channel per lane) are read with gfx950's transposing ds_read_b64_tr_b16: a tap's
// kw shift is then one voxel = 64 bytes of ADDRESS, where the [channel][8 voxels] images of round 2 needed the shifted
// fragments rebuilt in registers for every (K-step, kd, kh) (v_permlane32_swap + 5 v_alignbit per term, an edge image, ...:
// 4.3 of that kernel's 6.9 vector instructions per MFMA, which -- not the matrix pipe -- bounded it).  A half-wave's two
// transposed reads of a 4-voxel block cover 256 contiguous bytes: conflict free for every tap.
// An fp32 operand is scaled, split and written as the same words while it is staged (a thread takes 4 voxels x 8 channels).
//
// Reference operator served: the weight gradient autograd computes for nn.Conv3d(k=3, s=1, p=1) of convbn_3d
// (models/submodule.py:121-124) and the cva blocks (models/augment/cva.py:13-55).
//
// Work decomposition: persistent workgroups of 8 waves (one per CU).  A tile is 2 x 4 x 16 output voxels = 8 K-steps;
// the 8 waves are two groups of four, each group takes 4 K-steps, and inside a group the 27 taps are split 7/7/7/6 over
// the waves (7 x 16 accumulator registers per lane).  The next tile's words are fetched into registers during the MFMA
// phase and written to LDS between two barriers.  At the end the two groups are summed through LDS and the workgroup writes
// one slab of partial sums; wgrad_reduce_kernel (conv3d_wgrad.hip) adds the slabs in a fixed order: bitwise
// reproducible, no atomics.
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
constexpr int TD = 2, TH = 4, TW = 16;
constexpr int NROW = TD * TH;                         // 8 K-steps (output rows) per tile
constexpr int NGRP = 2, NACC = 7;
static_assert(NROW / NGRP == TH, "a K-step group is one d plane of the tile");
constexpr int HD = TD + 2, HH = TH + 2, NHROW = HD * HH;  // 24 halo rows
constexpr int XV = TW + 2;                            // 18 voxels per halo row (index 0 = w0 - 1)
constexpr int VB = 64;                                // bytes per voxel of an image: 32 channels x f16
constexpr int X_TERM = NHROW * XV * VB;               // 27648: [hrow][voxel][32 ci]
constexpr int Y_TERM = NROW * TW * VB;                // 8192:  [row][voxel][32 co]
constexpr int X_OFF = 0, Y_OFF = NT * X_TERM;
constexpr int IMG_BYTES = Y_OFF + NT * Y_TERM;        // 71680
// both operands packed: TWO image sets, filled by LDS-DMA (buffer_load ... lds) while the other one is multiplied: 143360;
// otherwise one set (71680); the end-of-kernel reduction reuses the LDS (114688)
constexpr int LDS_BYTES = 2 * IMG_BYTES;
static_assert(LDS_BYTES >= 4 * 7 * 4096 && LDS_BYTES + 128 <= 160 * 1024, "LDS");
constexpr int NXW = NT * NHROW * XV * 4, KXW = (NXW + 511) / 512;     // packed x: 3456 words -> 7 per thread
constexpr int NYW = NT * NROW * TW * 4, KYW = NYW / 512;              // packed dy: 1024 words -> 2 per thread
constexpr int NXQ = NHROW * 4 * 4, NXE = NHROW * 2 * 4, NYQ = NROW * 4 * 4;   // fp32: 384 x quads, 192 x edge voxels, 128 dy quads
static_assert(NXQ + NYQ == 512 && NYW % 512 == 0, "staging items");

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

// the 8 voxels x 1 channel MFMA fragment of this lane from a [voxel][32 channels] image: two transposing reads of 4 voxels
__device__ __forceinline__ f16x8 tr_frag(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * VB));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, c);
}

template <bool XP, bool YP>
__global__ __launch_bounds__(512) void wgrad3_f16x2_kernel(WX2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  // (wq stays a per-lane value on purpose: as a scalar the switch below becomes four real branches, and the compiler gives
  // the 112 accumulator registers a different home in each of them -- 56 v_mov_b64 each way per tile and ~500 bytes of
  // scratch; with a per-lane selector the four tap groups are predicated regions over one register assignment)
  const int grp = wv >> 2, wq = wv & 3;
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

  // transposing read (ds_read_b64_tr_b16, cdna_hip_programming.md T10): lane 4q+p of a 16-lane group supplies the address of
  // voxel q, channels 4p .. 4p+3 of the group's 4-voxel x 16-channel block and receives channel (lane & 15) of the 4 voxels;
  // group g = (lane >> 4) & 1 takes channels 16g .., the wave half the voxels 8 half .. 8 half + 7 of the K-step
  const int lane_off = (8 * half + ((lane & 15) >> 2)) * VB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  // ---- staging: global -> registers (load_tile) -> LDS words (store_tile) ----------------------------------------------
  // packed operand: word it of the image = LDS byte 16 it (it = ((term * rows + row) * voxels + voxel) * 4 + channel group)
  // fp32 operand: threads 0-383 one x quad item (hrow, quad, channel group: 4 voxels x 8 channels, 8 x b128), threads 0-191
  // also one x edge voxel (8 x b32), threads 384-511 one dy quad item; exponents of the thread's 8 channels in registers
  float4 pw[(XP ? KXW : 0) + (YP ? KYW : 0) + ((XP || YP) ? 0 : 1)];
  float4 rq[(XP && YP) ? 1 : 8];
  float re[XP ? 1 : 8];
  int ex8[(XP && YP) ? 1 : 8];
  if constexpr (!XP || !YP) {
    // x quads / edges of threads 0-383 use channel group (tid & 3) of x; dy quads (threads 384-511, or 0-127 when x is
    // packed) group (tid & 3) of dy
    const bool isy = !YP && (XP || tid >= NXQ);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (isy ? cy0 : cx0) + (tid & 3) * 8 + j;
      ex8[j] = (c < (isy ? a.Cy : a.Cx)) ? dca_coherent_loadi((isy ? a.yexps : a.xexps) + c) : 0;
    }
  }
  // packed operands: the (row, voxel, term, channel group) of each of the thread's words, packed into one register per word
  // (unpacked again at every use behind an opaque asm: left to itself the compiler hoists ~5 derived values per word out of
  // the tile loop and spills them -- and the staged words -- to scratch)
  int xcrd[XP ? KXW : 1], ycrd[YP ? KYW : 1];
  if constexpr (XP) {
#pragma unroll
    for (int k = 0; k < KXW; ++k) {
      const int it = tid + 512 * k, cg = it & 3, vv = it >> 2, v = vv % XV, r = vv / XV, hrow = r % NHROW, term = r / NHROW;
      xcrd[k] = (it < NXW) ? ((hrow / HH) | ((hrow % HH) << 4) | (v << 8) | (term << 16) | (cg << 20)) : -1;
    }
  }
  if constexpr (YP) {
#pragma unroll
    for (int k = 0; k < KYW; ++k) {
      const int it = tid + 512 * k, cg = it & 3, vv = it >> 2, v = vv % TW, r = vv / TW, row = r % NROW, term = r / NROW;
      ycrd[k] = (row / TH) | ((row % TH) << 4) | (v << 8) | (term << 16) | (cg << 20);
    }
  }
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    const int tw = tile % a.nTW; tile /= a.nTW;
    const int th = tile % a.nTH; tile /= a.nTH;
    const int td = tile % a.nTD;
    n = tile / a.nTD;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
  };
  // part 0: the first half of the x loads (packed: words 0-3; fp32: the quad), part 1: the rest of x and dy; -1: everything
  auto load_tile = [&](int n, int d0, int h0, int w0, int part = -1) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * xsample, xsample * 4);
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.dy + (long)n * ysample, ysample * 4);
    if constexpr (XP) {
#pragma unroll
      for (int k = 0; k < KXW; ++k) {
        if ((part == 0 && k >= 4) || (part == 1 && k < 4)) continue;
        int crd = xcrd[k];
        asm volatile("" : "+v"(crd));
        const int d = d0 - 1 + (crd & 15), h = h0 - 1 + ((crd >> 4) & 15), w = w0 - 1 + ((crd >> 8) & 255);
        const int term = (crd >> 16) & 1, g = (cx0 >> 3) + ((crd >> 20) & 3);
        const int ok = (int)(crd >= 0) & (int)(g * 8 < a.Cx) & (int)((unsigned)d < (unsigned)a.D) &
                       (int)((unsigned)h < (unsigned)a.H) & (int)((unsigned)w < (unsigned)a.W);
        pw[k] = dca_bload4(xr, term * (a.Cx * cstride * 2) + (g * cstride + (d * a.H + h) * a.W + w) * 16, ok);
      }
    } else {
      if (part != 1 && tid < NXQ) {
        const int cg = tid & 3, quad = (tid >> 2) & 3, hrow = tid >> 4;
        const int d = d0 - 1 + hrow / HH, h = h0 - 1 + hrow % HH, w = w0 + 4 * quad, c0 = cx0 + cg * 8;
        const int ok = (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) & (int)(w + 3 < a.W);   // W % 4 == 0
        const int off = (c0 * cstride + (d * a.H + h) * a.W + w) * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) rq[j] = dca_bload4(xr, off + j * cstride * 4, ok & (int)(c0 + j < a.Cx));
      }
      if (part != 0 && tid < NXE) {
        const int cg = tid & 3, side = (tid >> 2) & 1, hrow = tid >> 3;
        const int d = d0 - 1 + hrow / HH, h = h0 - 1 + hrow % HH, w = side ? w0 + TW : w0 - 1, c0 = cx0 + cg * 8;
        const int ok = (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) & (int)((unsigned)w < (unsigned)a.W);
        const int off = (c0 * cstride + (d * a.H + h) * a.W + w) * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) re[j] = dca_bload1(xr, off + j * cstride * 4, ok & (int)(c0 + j < a.Cx));
      }
    }
    if (part == 0) return;
    if constexpr (YP) {
#pragma unroll
      for (int k = 0; k < KYW; ++k) {
        int crd = ycrd[k];
        asm volatile("" : "+v"(crd));
        const int d = d0 + (crd & 15), h = h0 + ((crd >> 4) & 15), w = w0 + ((crd >> 8) & 255);
        const int term = (crd >> 16) & 1, g = (cy0 >> 3) + ((crd >> 20) & 3);
        const int ok = (int)(g * 8 < a.Cy) & (int)(d < a.D) & (int)(h < a.H) & (int)(w < a.W);
        pw[(XP ? KXW : 0) + k] = dca_bload4(yr, term * (a.Cy * cstride * 2) + (g * cstride + (d * a.H + h) * a.W + w) * 16, ok);
      }
    } else {
      if (XP || tid >= NXQ) {     // (with a packed x every thread is free for a dy quad: threads 0-127 take them)
        const int u = XP ? tid : tid - NXQ;
        if (u < NYQ) {
          const int cg = u & 3, quad = (u >> 2) & 3, row = u >> 4;
          const int d = d0 + row / TH, h = h0 + row % TH, w = w0 + 4 * quad, c0 = cy0 + cg * 8;
          const int ok = (int)(d < a.D) & (int)(h < a.H) & (int)(w + 3 < a.W);
          const int off = (c0 * cstride + (d * a.H + h) * a.W + w) * 4;
#pragma unroll
          for (int j = 0; j < 8; ++j) rq[j] = dca_bload4(yr, off + j * cstride * 4, ok & (int)(c0 + j < a.Cy));
        }
      }
    }
  };
  // Both operands packed: a tile's words go global -> LDS directly (word it of an image = LDS byte 16 it = wave-uniform base +
  // 16 * lane: exactly what buffer_load ... lds writes; the per-lane SOURCE address carries the halo geometry, an out-of-range
  // offset writes zeros) into the image set that is not being multiplied: no staging registers, no store phase between two
  // barriers -- ONE barrier per tile.  (With register staging a second image set spilled: DESIGN.md section 5.)
  // The copies are issued through inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and put `s_waitcnt
  // vmcnt(0)` in front of the next K-step's transposing reads (it cannot see that they touch the other image set), which
  // exposed the DMA latency twice per tile.  An asm load is outside its bookkeeping (cdna_hip_programming.md, "What hipcc does
  // not do"), so the kernel waits itself: vmcnt(0) in front of the tile's barrier; this variant has no other vector-memory
  // load in flight inside the tile loop.
  typedef __attribute__((address_space(3))) char lds_char;
  const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar copy for the LDS bases only (see the note on wq above)
  const unsigned lds0 = (unsigned)(unsigned long)(lds_char*)smem;
  auto desc = [&](const void* base, long bytes) __attribute__((always_inline)) {     // the words dca_rsrc builds, as SGPRs for the asm
    const unsigned long long b = (unsigned long long)base;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;
    r.z = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r.w = 0x00020000u;
    return r;
  };
  auto glds16 = [&](u32x4 rs, int voff, unsigned lds_byte) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rs) : "memory");
  };
  auto glds_tile = [&](int n, int d0, int h0, int w0, int part, int buf) __attribute__((always_inline)) {
    if constexpr (XP && YP) {
      const u32x4 xr = desc(a.x + (long)n * xsample, xsample * 4), yr = desc(a.dy + (long)n * ysample, ysample * 4);
      const unsigned img = lds0 + buf * IMG_BYTES;
#pragma unroll
      for (int k = 0; k < KXW; ++k) {
        if ((part == 0 && k >= 4) || (part == 1 && k < 4)) continue;
        if (512 * k + 64 * wvu < NXW) {        // NXW is a multiple of 64: whole waves
          int crd = xcrd[k];
          asm volatile("" : "+v"(crd));
          const int d = d0 - 1 + (crd & 15), h = h0 - 1 + ((crd >> 4) & 15), w = w0 - 1 + ((crd >> 8) & 255);
          const int term = (crd >> 16) & 1, g = (cx0 >> 3) + ((crd >> 20) & 3);
          const int ok = (int)(g * 8 < a.Cx) & (int)((unsigned)d < (unsigned)a.D) & (int)((unsigned)h < (unsigned)a.H) &
                         (int)((unsigned)w < (unsigned)a.W);
          glds16(xr, dca_pred_off(term * (a.Cx * cstride * 2) + (g * cstride + (d * a.H + h) * a.W + w) * 16, ok),
                 img + X_OFF + (512 * k + 64 * wvu) * 16);
        }
      }
      if (part == 0) return;
#pragma unroll
      for (int k = 0; k < KYW; ++k) {
        int crd = ycrd[k];
        asm volatile("" : "+v"(crd));
        const int d = d0 + (crd & 15), h = h0 + ((crd >> 4) & 15), w = w0 + ((crd >> 8) & 255);
        const int term = (crd >> 16) & 1, g = (cy0 >> 3) + ((crd >> 20) & 3);
        const int ok = (int)(g * 8 < a.Cy) & (int)(d < a.D) & (int)(h < a.H) & (int)(w < a.W);
        glds16(yr, dca_pred_off(term * (a.Cy * cstride * 2) + (g * cstride + (d * a.H + h) * a.W + w) * 16, ok),
               img + Y_OFF + (512 * k + 64 * wvu) * 16);
      }
    }
  };

  // the 8 channels of one voxel -> the two f16 words of its channel group
  auto split_word = [&](const float (&v)[8], char* dst, int term_stride) __attribute__((always_inline)) {
    f16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 h, l;
      split2(v[j], ex8[j], h, l);
      hv[j] = h; lv[j] = l;
    }
    *(f16x8*)dst = hv;
    *(f16x8*)(dst + term_stride) = lv;
  };
  auto quad_words = [&](char* dst, int term_stride) __attribute__((always_inline)) {   // rq[8 channels] -> 4 voxels' words
    const float* q0 = (const float*)&rq[0];
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) {
      const float v[8] = {q0[vv], q0[4 + vv], q0[8 + vv], q0[12 + vv], q0[16 + vv], q0[20 + vv], q0[24 + vv], q0[28 + vv]};
      split_word(v, dst + vv * VB, term_stride);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
    if constexpr (XP) {
#pragma unroll
      for (int k = 0; k < KXW; ++k) {
        const int it = tid + 512 * k;
        if (it < NXW) *(float4*)(smem + X_OFF + it * 16) = pw[k];
      }
    } else {
      if (tid < NXQ) {
        const int cg = tid & 3, quad = (tid >> 2) & 3, hrow = tid >> 4;
        quad_words(smem + X_OFF + (hrow * XV + 1 + 4 * quad) * VB + cg * 16, X_TERM);
      }
      if (tid < NXE) {
        const int cg = tid & 3, side = (tid >> 2) & 1, hrow = tid >> 3;
        const float v[8] = {re[0], re[1], re[2], re[3], re[4], re[5], re[6], re[7]};
        split_word(v, smem + X_OFF + (hrow * XV + (side ? XV - 1 : 0)) * VB + cg * 16, X_TERM);
      }
    }
    if constexpr (YP) {
#pragma unroll
      for (int k = 0; k < KYW; ++k) *(float4*)(smem + Y_OFF + (tid + 512 * k) * 16) = pw[(XP ? KXW : 0) + k];
    } else {
      if (XP || tid >= NXQ) {
        const int u = XP ? tid : tid - NXQ;
        if (u < NYQ) {
          const int cg = u & 3, quad = (u >> 2) & 3, row = u >> 4;
          quad_words(smem + Y_OFF + (row * TW + 4 * quad) * VB + cg * 16, Y_TERM);
        }
      }
    }
  };

  // The MFMA phase of one tile for tap group WQ (taps 7*WQ .. 7*WQ+6, < 27).  Every fragment is two transposing reads at a
  // compile-time offset from one per-K-step base; the next tap's fragments are requested before this tap's MFMAs.  The
  // next tile's global loads are issued BEHIND the first K-step's MFMAs (in front of the phase their address arithmetic
  // and the memory pipe's back-pressure kept all eight waves -- and the matrix pipe -- busy).
  auto mfma_tile = [&](auto WQC, bool more, int next_tile, int buf) __attribute__((always_inline)) {
    constexpr int WQ = decltype(WQC)::value;
    constexpr int TAP0 = 7 * WQ, TAP1 = (TAP0 + 7 < 27) ? TAP0 + 7 : 27, NTAP = TAP1 - TAP0;
#pragma unroll 1
    for (int i = 0; i < TH; ++i) {
      if ((i == 1 || i == 2) && more) {     // in two portions: one burst of up to 18 loads stalls the memory pipe -- and the MFMAs
        int nn, nd0, nh0, nw0;
        decode(next_tile, nn, nd0, nh0, nw0);
        if constexpr (XP && YP) glds_tile(nn, nd0, nh0, nw0, i - 1, buf ^ 1);
        else load_tile(nn, nd0, nh0, nw0, i - 1);
      }
      // K-step (d = d0 + grp, h = h0 + i): dy row grp*TH + i; x halo rows (grp + kd, i + kh)
      const char* yb = smem + buf * IMG_BYTES + Y_OFF + ((grp * TH + i) * TW) * VB + lane_off;
      const char* xb = smem + buf * IMG_BYTES + X_OFF + ((grp * HH + i) * XV) * VB + lane_off;
      f16x8 ay[NT];
#pragma unroll
      for (int term = 0; term < NT; ++term) ay[term] = tr_frag(yb + term * Y_TERM);
      f16x8 bx[2][NT];
      auto load_b = [&](int tap, int slot) __attribute__((always_inline)) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
        for (int term = 0; term < NT; ++term) bx[slot][term] = tr_frag(xb + term * X_TERM + ((kd * HH + kh) * XV + kw) * VB);
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
        // the next tap's four reads between this tap's three MFMAs, and nothing hoisted further ahead (left alone the
        // scheduler requests the fragments of all seven taps up front: 56 registers, and the staged words go to scratch)
        if (j + 1 < NTAP) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (t_begin < t_end) {
    int n, d0, h0, w0;
    decode(t_begin, n, d0, h0, w0);
    if constexpr (XP && YP) {
      glds_tile(n, d0, h0, w0, -1, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the asm copies
    } else {
      load_tile(n, d0, h0, w0);
      store_tile();
    }
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += t_step) {
      const bool more = tile + t_step < t_end;
      switch (wq) {
        case 0: mfma_tile(std::integral_constant<int, 0>{}, more, tile + t_step, buf); break;
        case 1: mfma_tile(std::integral_constant<int, 1>{}, more, tile + t_step, buf); break;
        case 2: mfma_tile(std::integral_constant<int, 2>{}, more, tile + t_step, buf); break;
        default: mfma_tile(std::integral_constant<int, 3>{}, more, tile + t_step, buf); break;
      }
      if constexpr (XP && YP) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's copies of the next tile have landed
      __syncthreads();  // every wave is done reading this tile [and all copies of the next one are in LDS]
      if constexpr (XP && YP) {
        buf ^= 1;
      } else {
        if (more) store_tile();
        __syncthreads();
      }
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

// floats of scratch `part` dca_conv3d_wgrad_x2 needs
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
