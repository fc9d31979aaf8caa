// Reduced-precision 3x3x3 stride-1 convolution (BASELINE configs 2 "bf16" and 5 "fp16"): bf16 / fp16 operands, ONE native
// matrix-core product per multiply (v_mfma_f32_32x32x16_{bf16,f16}), fp32 accumulation, fp32 folded-BatchNorm affine +
// activation + residual epilogue.  Activations are STORED in the 2-byte type (or fp32 at the boundaries of the
// reduced-precision region), so this kernel is bandwidth bound where its fp32-grade sibling conv3d_bf16x3.hip (six
// products per multiply) is matrix-pipe bound.  Inference only (no backward): the training path stays fp32.
//
// Reference operators served: nn.Conv3d(k=3, s=1, p=1) + BatchNorm3d (eval) + ReLU of convbn_3d
// (models/submodule.py:121-124) in dres0/dres1/classif* (models/gwcnet_dca_g.py:141-168) and the cva blocks
// (models/augment/cva.py:39-53, 13-31).
//
// Work decomposition (same tile geometry as conv3d_bf16x3.hip): persistent 8-wave workgroups, one per CU, XCD-aware
// contiguous tile ranges; a 4 x 8 x 16 output tile = 16 MFMA column tiles (two per wave) x 32 output channels.  Input
// channels go through LDS in chunks of 16 (one MFMA K): the chunk's 6 x 10 x 18 halo tile as [k half][voxel][8 x 2 B]
// (a lane's B fragment is one ds_read_b128) and all 27 taps of pre-swizzled weight fragments, so a chunk is one phase of
// 54 MFMAs per wave behind a single barrier while the next chunk's halo tile is fetched into registers by
// hardware-predicated buffer loads.  LDS modes (160 KB per CU):
//   MODE 1  Cin <= 48: the weight fragments of ALL chunks stay resident in LDS for the whole kernel (<= 81 KB, loaded
//           once per workgroup) and the halo image is double buffered (68 KB);
//   MODE 2  Cin <= 64: resident weights (108 KB), single halo image (two barriers per chunk);
//   MODE 0  wider inputs: weights stream through a double-buffered 27-tap slab next to the double-buffered halo image.
#include "dca_common.h"
#include "../../include/dca_hip.h"
#include <type_traits>

typedef __bf16 lp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lp_f16x8 __attribute__((ext_vector_type(8)));

// compile-time ablation switches for tools/lp_ablate.sh (never set in the shipped library):
// 1 no halo loads, 2 no MFMAs, 4 no epilogue stores, 8 no LDS fragment reads, 16 no LDS halo writes
#ifndef LP_ABL
#define LP_ABL 0
#endif
#ifndef LP_LOOK    // LDS fragment reads run this many taps ahead of their MFMAs (measured 1..4: 3 is best by ~2 %)
#define LP_LOOK 3
#endif
#ifndef LP_SGB
#define LP_SGB 1
#endif
#ifndef LP_NT      // non-temporal output stores
#define LP_NT 0
#endif
#ifndef LP_DEFER   // experiment: epilogue deferred into the next tile's first step (slower: the VALU work does not hide)
#define LP_DEFER 0
#endif

namespace {

constexpr int TD = 4, TH = 8, TW = 16;
constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
constexpr int NVOX = ID * IH * IW;                  // 1080 halo voxels
constexpr int B_IMG = 2 * NVOX * 16;                // 34560 B: (k half, voxel) x 8 two-byte channels
constexpr int A_SLAB = 27 * 1024;                   // 27 taps x (64 lanes x 16 B)
constexpr int lds_bytes(int mode, int nch) {   // dynamic part (+ 256 B static for the epilogue affine)
  return mode == 0 ? 2 * B_IMG + 2 * A_SLAB : (mode == 1 ? 2 : 1) * B_IMG + nch * A_SLAB;
}
constexpr int NB_ITEMS = 2 * NVOX;                  // unaligned path: (k half, voxel) items of 8 channels
constexpr int KB = (NB_ITEMS + 511) / 512;          // 5
constexpr int NROWS = 2 * ID * IH;                  // 120 halo rows of (k half, d, h)
constexpr int NQ = NROWS * 6;                       // aligned path: a row = the 6 aligned quads w0-4 .. w0+19 (24 voxels,
static_assert(NQ <= 1024, "two quad items per thread");  // 18 used): ONE contiguous 48 / 96-byte request per channel row
constexpr int NA_ITEMS = A_SLAB / 16;               // 1728 b128 per slab
constexpr int KA = (NA_ITEMS + 511) / 512;          // 4

struct LpArgs {
  const void* x;
  const unsigned short* wx;
  void* y;
  const float* scale;
  const float* shift;
  const void* res_pre;
  const void* res_post;
  float slope;
  int N, Cin, Cout, NCH;
  int D, H, W;
  int nTD, nTH, nTW;
};

// 2-byte matrix types: conversion from fp32 (round to nearest even), raw 16-bit pattern, MFMA
template <typename MT> struct Lp;
template <> struct Lp<__bf16> {
  typedef lp_bf16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Lp<_Float16> {
  typedef lp_f16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <typename MT> __device__ __forceinline__ unsigned short lp_bits(float v) {
  const MT m = (MT)v;
  return __builtin_bit_cast(unsigned short, m);
}
template <typename MT> __device__ __forceinline__ float lp_float(unsigned short b) {
  return (float)__builtin_bit_cast(MT, b);
}
// two fp32 values -> one dword of two 2-byte values (lo = a, hi = b), round to nearest even
template <typename MT> __device__ __forceinline__ unsigned lp_pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef MT mtx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mtx2));
}
// value of lane ^ 1 (DPP quad_perm [1,0,3,2]: no LDS traffic, unlike __shfl_xor)
__device__ __forceinline__ unsigned lp_swap1(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned short dca_bload_u16(__amdgpu_buffer_rsrc_t r, int byte_off, int ok) {
  return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, dca_pred_off(byte_off, ok), 0, 0);
}
__device__ __forceinline__ void dca_bstore_u16(__amdgpu_buffer_rsrc_t r, unsigned short v, int byte_off, int ok) {
  __builtin_amdgcn_raw_buffer_store_b16((short)v, r, dca_pred_off(byte_off, ok), 0, 0);
}

// MT: matrix type; IN32 / OUT32: the input / output (and residual) tensors are fp32 instead of MT;
// VEC: W % 4 == 0 and an aligned base, so a halo row is fetched as 6 aligned quads.
template <typename MT, bool IN32, bool OUT32, bool VEC, int MODE>
__global__ __launch_bounds__(512) void conv3_lp_kernel(LpArgs a) {
  typedef typename Lp<MT>::vec8 vec8;
  constexpr int ISZ = IN32 ? 4 : 2, OSZ = OUT32 ? 4 : 2;
  constexpr bool RES = MODE != 0, DBUF = MODE != 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;                           // halo image(s)
  char* a_lds = smem + (DBUF ? 2 : 1) * B_IMG;  // weight slabs: all chunks (RES) or two streaming buffers
  __shared__ float aff_lds[64];                 // this channel block's folded-BN scale | shift (read once per kernel)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int cblk = blockIdx.y;
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  if (t_begin >= t_end) return;

  // ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...:
  // MI355X_MICROARCH.md, LDS).  A column tile is two h-rows of 16 voxels; the second row starts IW*16 = 288 B = 32 B
  // (mod 256) after the first, which put lanes 20-27 on the banks of lanes 12-15 (2-way conflict in every group: 39 % of
  // this kernel's LDS cycles were SQ_LDS_BANK_CONFLICT).  Rotating the second row's w by -2 voxels makes the bank pattern of
  // lanes 16-31 equal to that of a contiguous 1 KiB read: conflict free.  The epilogue uses the same lane -> w map.
  const int wlane = (l31 & 16) ? (((l31 & 15) - 2) & 15) : (l31 & 15);
  int boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (wv * 2 + t) * 2 + (l31 >> 4), dl = r >> 3, hl = r & 7;
    boff[t] = (half * NVOX + (dl * IH + hl) * IW + wlane) * 16;
  }

  const int cstride = a.D * a.H * a.W;
  const long sample = (long)a.Cin * cstride;
  const long wbytes = (long)a.NCH * A_SLAB;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc((const char*)a.wx + (long)cblk * wbytes, wbytes);
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  if (tid < 64) {
    const int co = min(cblk * 32 + (tid & 31), a.Cout - 1);
    aff_lds[tid] = has_aff ? (tid < 32 ? a.scale[co] : a.shift[co]) : (tid < 32 ? 1.f : 0.f);
  }

  float4 ra[KA];
  auto load_A = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      ra[k] = dca_bload4(wr, chunk * A_SLAB + it * 16, (int)(it < NA_ITEMS));
    }
  };
  auto store_A = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      if (it < NA_ITEMS) *(float4*)(a_lds + buf * A_SLAB + it * 16) = ra[k];
    }
  };

  // staging registers: raw 32-bit words of the loads (fp32 values, or pairs of 2-byte values)
  unsigned rq[VEC ? 2 : 1][VEC ? 8 : 1][IN32 ? 4 : 2];
  unsigned rb[VEC ? 1 : KB][8];
  int item_crd[VEC ? 2 : KB];   // packed halo coordinates (d | h << 8 | w or quad << 16 | k half << 24), fixed per thread
  if constexpr (VEC) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = tid + 512 * k, row = it / 6, q = it - row * 6;
      const int kh = row / (ID * IH), rem = row - kh * (ID * IH), id = rem / IH, ih = rem - id * IH;
      item_crd[k] = (it < NQ) ? (id | (ih << 8) | (q << 16) | (kh << 24)) : -1;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int it = tid + 512 * k;
      const int kh = it / NVOX, v = it - kh * NVOX;
      const int id = v / (IH * IW), rem = v - id * (IH * IW), ih = rem / IW, iw = rem - ih * IW;
      item_crd[k] = (it < NB_ITEMS) ? (id | (ih << 8) | (iw << 16) | (kh << 24)) : -1;
    }
  }
  auto item_off = [&](int crd, int d0, int h0, int w0, int chunk, int& c0, int& okv) __attribute__((always_inline)) {
    const int di = d0 - 1 + (crd & 255), hi = h0 - 1 + ((crd >> 8) & 255);
    const int wi = VEC ? w0 - 4 + 4 * ((crd >> 16) & 255) : w0 - 1 + ((crd >> 16) & 255);   // VEC: first voxel of the quad
    c0 = chunk * 16 + ((crd >> 24) & 1) * 8;
    okv = (int)(crd >= 0) & (int)((unsigned)di < (unsigned)a.D) & (int)((unsigned)hi < (unsigned)a.H) &
          (int)((unsigned)wi < (unsigned)a.W);
    return (c0 * cstride + (di * a.H + hi) * a.W + wi) * ISZ;
  };
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc((const char*)a.x + (long)n * sample * ISZ, sample * ISZ);
    if constexpr (VEC) {
      // one predicated byte offset per item; the 8 channels add a uniform stride, and a channel >= Cin lands beyond the
      // descriptor's range (Cin * cstride elements), so the hardware range check supplies the zero padding of the chunk
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int c0, okv;
        const int base = dca_pred_off(item_off(item_crd[k], d0, h0, w0, chunk, c0, okv), okv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (IN32) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, base + j * cstride * 4, 0, 0);
            rq[k][j][0] = v.x; rq[k][j][1] = v.y; rq[k][j][2] = v.z; rq[k][j][3] = v.w;
          } else {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xr, base + j * cstride * 2, 0, 0);
            rq[k][j][0] = v.x; rq[k][j][1] = v.y;
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        int c0, okv;
        const int off = item_off(item_crd[k], d0, h0, w0, chunk, c0, okv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ok = okv & (int)(c0 + j < a.Cin);
          if constexpr (IN32) rb[k][j] = __builtin_amdgcn_raw_buffer_load_b32(xr, dca_pred_off(off + j * cstride * 4, ok), 0, 0);
          else rb[k][j] = dca_bload_u16(xr, off + j * cstride * 2, ok);
        }
      }
    }
  };
  // eight channel values of one voxel (16-bit patterns in the low halves) -> one 16-byte LDS element
  auto put_voxel = [&](char* img, int vox_off, const unsigned (&v)[8]) __attribute__((always_inline)) {
    const u32x4 o = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    *(u32x4*)(img + vox_off) = o;
  };
  auto crd_lds = [&](int crd) __attribute__((always_inline)) {   // unaligned path: the item's voxel
    return ((((crd >> 24) & 1) * ID + (crd & 255)) * IH + ((crd >> 8) & 255)) * IW * 16 + ((crd >> 16) & 255) * 16;
  };
  auto store_B = [&](int buf) __attribute__((always_inline)) {
    char* img = b_lds + buf * B_IMG;
    if constexpr (VEC) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int crd = item_crd[k];
        if (crd >= 0) {
          const int q = (crd >> 16) & 255;
          // LDS offset of halo voxel iw = 4q - 3 + i of this (k half, d, h) row; iw in [0, 17] exists
          const int o_base = ((((crd >> 24) & 1) * ID + (crd & 255)) * IH + ((crd >> 8) & 255)) * IW * 16 + (4 * q - 3) * 16;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if ((q == 0 && i < 3) || (q == 5 && i > 0)) continue;
            u32x4 o;
            if constexpr (IN32) {
              o.x = lp_pack2<MT>(__uint_as_float(rq[k][0][i]), __uint_as_float(rq[k][1][i]));
              o.y = lp_pack2<MT>(__uint_as_float(rq[k][2][i]), __uint_as_float(rq[k][3][i]));
              o.z = lp_pack2<MT>(__uint_as_float(rq[k][4][i]), __uint_as_float(rq[k][5][i]));
              o.w = lp_pack2<MT>(__uint_as_float(rq[k][6][i]), __uint_as_float(rq[k][7][i]));
            } else {   // voxel i of the quad = half (i & 1) of word i >> 1 of every channel: one v_perm per channel pair
              constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;
              const unsigned sel = (i & 1) ? HI : LO;
              o.x = __builtin_amdgcn_perm(rq[k][1][i >> 1], rq[k][0][i >> 1], sel);
              o.y = __builtin_amdgcn_perm(rq[k][3][i >> 1], rq[k][2][i >> 1], sel);
              o.z = __builtin_amdgcn_perm(rq[k][5][i >> 1], rq[k][4][i >> 1], sel);
              o.w = __builtin_amdgcn_perm(rq[k][7][i >> 1], rq[k][6][i >> 1], sel);
            }
            *(u32x4*)(img + o_base + 16 * i) = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (item_crd[k] >= 0) {
          unsigned v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = IN32 ? (unsigned)lp_bits<MT>(__uint_as_float(rb[k][j])) : (rb[k][j] & 0xffffu);
          put_voxel(img, crd_lds(item_crd[k]), v);
        }
    }
  };
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    const int tw = tile % a.nTW; tile /= a.nTW;
    const int th = tile % a.nTH; tile /= a.nTH;
    const int td = tile % a.nTD;
    n = tile / a.nTD;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
  };

  // MODE 1 runs a deeper software pipeline: the halo tile of step s+2 (a step = one chunk of one tile) is requested in
  // the MIDDLE of step s, right after the registers that held step s+1's tile were written to the other LDS image --
  // a full step of latency tolerance, and the pack + ds_write work overlaps the second half of the step's MFMAs.
  constexpr bool PIPE = MODE == 1;
  constexpr int SPLIT = 13, ESPLIT = 3;
  struct Cursor { int tile, chunk, n, d0, h0, w0; };
  auto advance = [&](Cursor& c) __attribute__((always_inline)) {   // next step; c.tile >= t_end marks "none"
    if (c.chunk + 1 < a.NCH) { ++c.chunk; return; }
    c.chunk = 0;
    c.tile += t_step;
    if (c.tile < t_end) decode(c.tile, c.n, c.d0, c.h0, c.w0);
  };

  int n, d0, h0, w0;
  decode(t_begin, n, d0, h0, w0);
  load_B(n, d0, h0, w0, 0);
  load_A(0);
  store_B(0);
  store_A(0);
  if constexpr (RES) {
#pragma unroll 1
    for (int c = 1; c < a.NCH; ++c) {
      load_A(c);
      store_A(c);
    }
  }
  Cursor c1{t_begin, 0, n, d0, h0, w0}, c2;
  if constexpr (PIPE) {
    advance(c1);
    c2 = c1;
    if (c1.tile < t_end) {
      load_B(c1.n, c1.d0, c1.h0, c1.w0, c1.chunk);   // stays in registers until the middle of step 0
      advance(c2);
    }
  }
  __syncthreads();

  // epilogue: y = act(acc * scale + shift + res_pre) + res_post, fp32 arithmetic, stored as OUT.  `live` = 0 turns every
  // access into an out-of-range one (the deferred epilogue slot of a step that has nothing to write).
  auto epilogue = [&](const f32x16 (&eacc)[2], int n, int d0, int h0, int w0, int live) __attribute__((always_inline)) {
    const long osample = (long)a.Cout * cstride;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc((char*)a.y + (long)n * osample * OSZ, osample * OSZ);
    const __amdgpu_buffer_rsrc_t pr = dca_rsrc((const char*)(has_pre ? a.res_pre : a.y) + (long)n * osample * OSZ, osample * OSZ);
    const __amdgpu_buffer_rsrc_t qr = dca_rsrc((const char*)(has_post ? a.res_post : a.y) + (long)n * osample * OSZ, osample * OSZ);
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;
      sc[r] = aff_lds[cl];
      sh[r] = aff_lds[32 + cl];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r0 = (wv * 2 + t) * 2 + (l31 >> 4), d = d0 + (r0 >> 3), h = h0 + (r0 & 7), w = w0 + wlane;
      const int ok = live & (int)(d < a.D) & (int)(h < a.H) & (int)(w < a.W) & (int)!((LP_ABL & 4) && eacc[t][1] != 12345.f);
      // Byte offsets: ONE hardware-predicated base per lane; register r adds the uniform cu(r) * cstride.  An output
      // channel >= Cout lands beyond the descriptor's range (Cout * cstride elements) and is dropped / read as zero by
      // the range check, so partial channel blocks need no per-register predicate.
      float rp[16], rq_[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rp[r] = rq_[r] = 0.f;
      if constexpr (OUT32 || !VEC) {
        const int base = dca_pred_off(((d * a.H + h) * a.W + w + (cblk * 32 + 4 * half) * cstride) * OSZ, ok);
        if (has_pre) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int off = base + ((r & 3) + 8 * (r >> 2)) * cstride * OSZ;
            rp[r] = OUT32 ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pr, off, 0, 0))
                          : lp_float<MT>((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(pr, off, 0, 0));
          }
        }
        if (has_post) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int off = base + ((r & 3) + 8 * (r >> 2)) * cstride * OSZ;
            rq_[r] = OUT32 ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(qr, off, 0, 0))
                           : lp_float<MT>((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(qr, off, 0, 0));
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = base + ((r & 3) + 8 * (r >> 2)) * cstride * OSZ;
          const float v = act_apply(eacc[t][r] * sc[r] + sh[r] + rp[r], a.slope) + rq_[r];
          if constexpr (OUT32) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, off, 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b16((short)lp_bits<MT>(v), yr, off, 0, 0);
        }
      } else {
        // 2-byte output, W even: the voxels of lanes (2i, 2i+1) form a 4-byte aligned pair.  Registers go in pairs
        // (r, r+1) = channels (c, c+1): the even lane stores / loads the pair of voxels for channel c, the odd lane for
        // channel c+1 -- 8 dword accesses per lane instead of 16 short ones; the exchange is one DPP move + one v_perm.
        const int odd = lane & 1;
        const unsigned sel = odd ? 0x07060302u : 0x01000504u;
        const int base = dca_pred_off(((d * a.H + h) * a.W + (w - odd) + (cblk * 32 + 4 * half + odd) * cstride) * 2, ok);
        auto unpack = [&](const __amdgpu_buffer_rsrc_t& rr, float (&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const unsigned L = __builtin_amdgcn_raw_buffer_load_b32(rr, base + ((r & 3) + 8 * (r >> 2)) * cstride * 2, 0, 0);
            const unsigned M = lp_swap1(L);   // the partner's pair: the other channel, same two voxels
            dst[r] = lp_float<MT>((unsigned short)(odd ? (M >> 16) : (L & 0xffffu)));
            dst[r + 1] = lp_float<MT>((unsigned short)(odd ? (L >> 16) : (M & 0xffffu)));
          }
        };
        if (has_pre) unpack(pr, rp);
        if (has_post) unpack(qr, rq_);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float v0 = act_apply(eacc[t][r] * sc[r] + sh[r] + rp[r], a.slope) + rq_[r];
          const float v1 = act_apply(eacc[t][r + 1] * sc[r + 1] + sh[r + 1] + rp[r + 1], a.slope) + rq_[r + 1];
          const unsigned P = lp_pack2<MT>(v0, v1), Q = lp_swap1(P);
          // even lane: [my v0 | partner's v0]; odd lane: [partner's v1 | my v1]
          const unsigned word = __builtin_amdgcn_perm(P, Q, sel);
          __builtin_amdgcn_raw_buffer_store_b32(word, yr, base + ((r & 3) + 8 * (r >> 2)) * cstride * 2, 0, LP_NT ? 2 : 0);
        }
      }
    }
  };
  f32x16 pacc[2];
  int pn = 0, pd0 = 0, ph0 = 0, pw0 = 0, pvalid = 0;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[t][r] = 0.f;

  int buf = 0;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool more_tiles = tile + t_step < t_end;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int nn = n, nd0 = d0, nh0 = h0, nw0 = w0;
    if (more_tiles) decode(tile + t_step, nn, nd0, nh0, nw0);

#pragma unroll 1
    for (int chunk = 0; chunk < a.NCH; ++chunk, buf ^= (DBUF ? 1 : 0)) {
      const bool next_chunk = chunk + 1 < a.NCH;
      const bool stage = next_chunk || more_tiles;
      const char* ab = a_lds + (RES ? chunk : buf) * A_SLAB + lane * 16;
      const char* bb = b_lds + buf * B_IMG;
      if (!PIPE && stage && !(LP_ABL & 1)) {
        if (next_chunk) load_B(n, d0, h0, w0, chunk + 1); else load_B(nn, nd0, nh0, nw0, 0);
        if constexpr (!RES) load_A(next_chunk ? chunk + 1 : 0);
      }
      // LDS fragment reads run LP_LOOK taps ahead of the MFMAs that consume them (register ring of LP_LOOK + 1 slots);
      // sched_group_barrier pins "one MFMA, then the reads" -- left alone hipcc issues each read right before its MFMA
      // and waits for the full LDS round trip.
      constexpr int LOOK = LP_LOOK, NS = LOOK + 1;
      vec8 fa[NS], fb[NS][2];
      auto load_frag = [&](int tap, int slot) __attribute__((always_inline)) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        fa[slot] = *(const vec8*)(ab + tap * 1024);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[slot][t] = *(const vec8*)(bb + boff[t] + ((kd * IH + kh) * IW + kw) * 16);
      };
#pragma unroll
      for (int t0 = 0; t0 < LOOK; ++t0) load_frag(t0, t0);
      if (LP_ABL & 8) load_frag(LOOK, LOOK);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int cur = tap % NS;
        if (PIPE && LP_DEFER && tap == ESPLIT) {   // the previous tile's epilogue, overlapped with this tile's first MFMAs
          epilogue(pacc, pn, pd0, ph0, pw0, pvalid & (int)(chunk == 0));
          if (chunk == 0) pvalid = 0;
        }
        if (PIPE && tap == SPLIT) {
          if (c1.tile < t_end && !(LP_ABL & 16)) store_B(buf ^ 1);                                  // step s+1 -> other image
          if (c2.tile < t_end && !(LP_ABL & 1)) load_B(c2.n, c2.d0, c2.h0, c2.w0, c2.chunk);        // request step s+2
          c1 = c2;
          advance(c2);
        }
        if (tap + LOOK < 27 && !(LP_ABL & 8)) load_frag(tap + LOOK, (tap + LOOK) % NS);
        if (!(LP_ABL & 2)) {
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t] = Lp<MT>::mfma(fa[cur], fb[cur][t], acc[t]);
        } else {
          acc[0][0] += (float)fa[cur][0] + (float)fb[cur][0][0] + (float)fb[cur][1][0];
        }
        if (LP_SGB && tap + LOOK < 27 && tap != SPLIT) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      if constexpr (PIPE) {
        __syncthreads();
      } else if constexpr (DBUF) {
        if (stage && !(LP_ABL & 16)) {
          store_B(buf ^ 1);
          if constexpr (!RES) store_A(buf ^ 1);
        }
        __syncthreads();
      } else {
        __syncthreads();   // every wave is done reading the single halo image
        if (stage) store_B(0);
        __syncthreads();
      }
    }

    if constexpr (PIPE && LP_DEFER) {   // deferred: runs inside the first step of the next tile (or after the loop)
#pragma unroll
      for (int t = 0; t < 2; ++t) pacc[t] = acc[t];
      pn = n; pd0 = d0; ph0 = h0; pw0 = w0;
      pvalid = 1;
    } else {
      epilogue(acc, n, d0, h0, w0, 1);
    }
    n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
  }
  if constexpr (PIPE && LP_DEFER) epilogue(pacc, pn, pd0, ph0, pw0, pvalid);
}

// wx[cblk][chunk][tap][lane][j] (2-byte): lane (r = lane & 31, h = lane >> 5) holds A[row = output channel
// cblk*32 + r][k = input channel chunk*16 + 8h + j] of the tap, zero padded.  Source indexing as
// dca_conv3d_prep_weight: src_ab ? src[a][b][27] : src[b][a][27]; flip reverses the tap order.
template <typename MT>
__global__ void lp_prep_weight_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int A, int Bn,
                                      int NCH, int src_ab, int flip, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    long t = idx >> 9;
    const int tap = t % 27; t /= 27;
    const int chunk = t % NCH;
    const int cblk = (int)(t / NCH);
    const int bi = cblk * 32 + (lane & 31), ai = chunk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (ai < A && bi < Bn) {
      const int st = flip ? 26 - tap : tap;
      v = src_ab ? src[((long)ai * Bn + bi) * 27 + st] : src[((long)bi * A + ai) * 27 + st];
    }
    dst[idx] = lp_bits<MT>(v);
  }
}

template <typename MT, bool IN32, bool OUT32, bool VEC, int MODE>
int launch_lp2(const LpArgs& a, int gx, int cblks, hipStream_t stream) {
  auto kern = conv3_lp_kernel<MT, IN32, OUT32, VEC, MODE>;
  const int lds = lds_bytes(MODE, a.NCH);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(gx, cblks), dim3(512), lds, stream, a);
  return dca_launch_status();
}

template <typename MT, bool IN32, bool OUT32>
int launch_lp(const LpArgs& a, bool vec, int gx, int cblks, hipStream_t stream) {
  const int mode = a.NCH <= 3 ? 1 : (a.NCH == 4 ? 2 : 0);
  if (vec) {
    if (mode == 1) return launch_lp2<MT, IN32, OUT32, true, 1>(a, gx, cblks, stream);
    if (mode == 2) return launch_lp2<MT, IN32, OUT32, true, 2>(a, gx, cblks, stream);
    return launch_lp2<MT, IN32, OUT32, true, 0>(a, gx, cblks, stream);
  }
  if (mode == 1) return launch_lp2<MT, IN32, OUT32, false, 1>(a, gx, cblks, stream);
  if (mode == 2) return launch_lp2<MT, IN32, OUT32, false, 2>(a, gx, cblks, stream);
  return launch_lp2<MT, IN32, OUT32, false, 0>(a, gx, cblks, stream);
}

}  // namespace

extern "C" long dca_conv3d_lp_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (long)((Cout + 31) / 32) * ((Cin + 15) / 16) * A_SLAB;
}

extern "C" int dca_conv3d_lp_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip, int dtype,
                                         hipStream_t stream) {
  DCA_REQUIRE(w && wx && A > 0 && B > 0 && (dtype == DCA_BF16 || dtype == DCA_FP16));
  const int NCH = (A + 15) / 16;
  const long total = dca_conv3d_lp_weight_bytes(A, B) / 2;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == DCA_BF16)
    hipLaunchKernelGGL(lp_prep_weight_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, w, (unsigned short*)wx, A, B,
                       NCH, src_ab, flip, total);
  else
    hipLaunchKernelGGL(lp_prep_weight_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, w, (unsigned short*)wx, A, B,
                       NCH, src_ab, flip, total);
  return dca_launch_status();
}

extern "C" int dca_conv3d_lp_forward(const void* x, const void* wx, void* y, const float* scale, const float* shift,
                                     const void* res_pre, const void* res_post, float slope, int N, int Cin, int Cout,
                                     int D, int H, int W, int dtype, int in_f32, int out_f32, hipStream_t stream) {
  DCA_REQUIRE(x && wx && y && N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE(dtype == DCA_BF16 || dtype == DCA_FP16);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  // 32-bit byte offsets inside one sample, with room for the out-of-range marker + 7 channel strides
  DCA_REQUIRE((long)(Cin > 8 ? Cin : 8) * D * H * W * 4 < 0x7ffffff0L && (long)(Cout > 32 ? Cout : 32) * D * H * W * 4 < 0x7ffffff0L);
  DCA_REQUIRE((((uintptr_t)wx) & 15) == 0);
  LpArgs a;
  a.x = x; a.wx = (const unsigned short*)wx; a.y = y;
  a.scale = scale; a.shift = shift; a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NCH = (Cin + 15) / 16;
  a.D = D; a.H = H; a.W = W;
  a.nTD = cdiv(D, TD); a.nTH = cdiv(H, TH); a.nTW = cdiv(W, TW);
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL && (Cout + 31) / 32 <= 65535);
  const bool vec = (W % 4 == 0) && ((((uintptr_t)x) & (in_f32 ? 15 : 7)) == 0);
  const int cblks = (Cout + 31) / 32;
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      ncu = v;
  }
  int gx = ncu / cblks > 0 ? ncu / cblks : 1;
  if (gx > tiles) gx = (int)tiles;
  if (dtype == DCA_BF16) {
    if (in_f32) return out_f32 ? launch_lp<__bf16, true, true>(a, vec, gx, cblks, stream)
                               : launch_lp<__bf16, true, false>(a, vec, gx, cblks, stream);
    return out_f32 ? launch_lp<__bf16, false, true>(a, vec, gx, cblks, stream)
                   : launch_lp<__bf16, false, false>(a, vec, gx, cblks, stream);
  }
  if (in_f32) return out_f32 ? launch_lp<_Float16, true, true>(a, vec, gx, cblks, stream)
                             : launch_lp<_Float16, true, false>(a, vec, gx, cblks, stream);
  return out_f32 ? launch_lp<_Float16, false, true>(a, vec, gx, cblks, stream)
                 : launch_lp<_Float16, false, false>(a, vec, gx, cblks, stream);
}
