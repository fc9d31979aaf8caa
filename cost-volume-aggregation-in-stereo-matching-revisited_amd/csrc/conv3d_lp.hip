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
// (a lane's B fragment is one ds_read_b128) and all 27 taps of pre-swizzled weight fragments.  Both are DOUBLE buffered
// (124 KB of the CU's 160 KB), so a chunk is one phase of 54 MFMAs per wave behind a single barrier while the next
// chunk's halo tile and weights are fetched into registers by hardware-predicated buffer loads.
#include "dca_common.h"
#include "../../include/dca_hip.h"
#include <type_traits>

typedef __bf16 lp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lp_f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TD = 4, TH = 8, TW = 16;
constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
constexpr int NVOX = ID * IH * IW;                  // 1080 halo voxels
constexpr int B_IMG = 2 * NVOX * 16;                // 34560 B: (k half, voxel) x 8 two-byte channels
constexpr int A_SLAB = 27 * 1024;                   // 27 taps x (64 lanes x 16 B)
constexpr int LDS_BYTES = 2 * B_IMG + 2 * A_SLAB;   // 124416
constexpr int NB_ITEMS = 2 * NVOX;                  // unaligned path: (k half, voxel) items of 8 channels
constexpr int KB = (NB_ITEMS + 511) / 512;          // 5
constexpr int NROWS = 2 * ID * IH;                  // 120 halo rows: 4 aligned quads + 2 edge voxels each
constexpr int NQUAD = NROWS * 4, NEDGE = NROWS * 2;
static_assert(NQUAD <= 512 && NEDGE <= 512, "one quad / edge item per thread");
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
__device__ __forceinline__ unsigned short dca_bload_u16(__amdgpu_buffer_rsrc_t r, int byte_off, int ok) {
  return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, dca_pred_off(byte_off, ok), 0, 0);
}
__device__ __forceinline__ void dca_bstore_u16(__amdgpu_buffer_rsrc_t r, unsigned short v, int byte_off, int ok) {
  __builtin_amdgcn_raw_buffer_store_b16((short)v, r, dca_pred_off(byte_off, ok), 0, 0);
}

// MT: matrix type; IN32 / OUT32: the input / output (and residual) tensors are fp32 instead of MT;
// VEC: W % 4 == 0 and an aligned base, so a halo row is 4 aligned quads + 2 edge voxels.
template <typename MT, bool IN32, bool OUT32, bool VEC>
__global__ __launch_bounds__(512) void conv3_lp_kernel(LpArgs a) {
  typedef typename Lp<MT>::vec8 vec8;
  constexpr int ISZ = IN32 ? 4 : 2, OSZ = OUT32 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;                 // two halo images
  char* a_lds = smem + 2 * B_IMG;     // two weight slabs

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int cblk = blockIdx.y;
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  if (t_begin >= t_end) return;

  int boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (wv * 2 + t) * 2 + (l31 >> 4), dl = r >> 3, hl = r & 7;
    boff[t] = (half * NVOX + (dl * IH + hl) * IW + (l31 & 15)) * 16;
  }

  const int cstride = a.D * a.H * a.W;
  const long sample = (long)a.Cin * cstride;
  const long wbytes = (long)a.NCH * A_SLAB;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc((const char*)a.wx + (long)cblk * wbytes, wbytes);
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;

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
  unsigned rq[VEC ? 8 : 1][IN32 ? 4 : 2];
  unsigned re[VEC ? 8 : 1];
  unsigned rb[VEC ? 1 : KB][8];
  int item_crd[VEC ? 2 : KB];
  if constexpr (VEC) {
    {
      const int row = tid >> 2, q = tid & 3, kh = row / (ID * IH), rem = row - kh * (ID * IH), id = rem / IH, ih = rem - id * IH;
      item_crd[0] = (tid < NQUAD) ? (id | (ih << 8) | ((1 + 4 * q) << 16) | (kh << 24)) : -1;
    }
    {
      const int row = tid >> 1, side = tid & 1, kh = row / (ID * IH), rem = row - kh * (ID * IH), id = rem / IH, ih = rem - id * IH;
      item_crd[1] = (tid < NEDGE) ? (id | (ih << 8) | ((side ? IW - 1 : 0) << 16) | (kh << 24)) : -1;
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
    const int di = d0 - 1 + (crd & 255), hi = h0 - 1 + ((crd >> 8) & 255), wi = w0 - 1 + ((crd >> 16) & 255);
    c0 = chunk * 16 + ((crd >> 24) & 1) * 8;
    okv = (int)(crd >= 0) & (int)((unsigned)di < (unsigned)a.D) & (int)((unsigned)hi < (unsigned)a.H) &
          (int)((unsigned)wi < (unsigned)a.W);
    return (c0 * cstride + (di * a.H + hi) * a.W + wi) * ISZ;
  };
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc((const char*)a.x + (long)n * sample * ISZ, sample * ISZ);
    if constexpr (VEC) {
      int c0, okv;
      const int offq = item_off(item_crd[0], d0, h0, w0, chunk, c0, okv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ok = okv & (int)(c0 + j < a.Cin);
        if constexpr (IN32) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, dca_pred_off(offq + j * cstride * 4, ok), 0, 0);
          rq[j][0] = v.x; rq[j][1] = v.y; rq[j][2] = v.z; rq[j][3] = v.w;
        } else {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xr, dca_pred_off(offq + j * cstride * 2, ok), 0, 0);
          rq[j][0] = v.x; rq[j][1] = v.y;
        }
      }
      const int offe = item_off(item_crd[1], d0, h0, w0, chunk, c0, okv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ok = okv & (int)(c0 + j < a.Cin);
        if constexpr (IN32) re[j] = __builtin_amdgcn_raw_buffer_load_b32(xr, dca_pred_off(offe + j * cstride * 4, ok), 0, 0);
        else re[j] = dca_bload_u16(xr, offe + j * cstride * 2, ok);
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
  auto crd_lds = [&](int crd) __attribute__((always_inline)) {
    return ((((crd >> 24) & 1) * ID + (crd & 255)) * IH + ((crd >> 8) & 255)) * IW * 16 + ((crd >> 16) & 255) * 16;
  };
  auto store_B = [&](int buf) __attribute__((always_inline)) {
    char* img = b_lds + buf * B_IMG;
    if constexpr (VEC) {
      if (item_crd[0] >= 0) {
        const int o = crd_lds(item_crd[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if constexpr (IN32) v[j] = lp_bits<MT>(__uint_as_float(rq[j][i]));
            else v[j] = (i & 1) ? (rq[j][i >> 1] >> 16) : (rq[j][i >> 1] & 0xffffu);
          }
          put_voxel(img, o + 16 * i, v);
        }
      }
      if (item_crd[1] >= 0) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = IN32 ? (unsigned)lp_bits<MT>(__uint_as_float(re[j])) : (re[j] & 0xffffu);
        put_voxel(img, crd_lds(item_crd[1]), v);
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

  int n, d0, h0, w0;
  decode(t_begin, n, d0, h0, w0);
  load_B(n, d0, h0, w0, 0);
  load_A(0);
  store_B(0);
  store_A(0);
  __syncthreads();

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
    for (int chunk = 0; chunk < a.NCH; ++chunk, buf ^= 1) {
      const bool next_chunk = chunk + 1 < a.NCH;
      const bool stage = next_chunk || more_tiles;
      if (stage) {
        if (next_chunk) load_B(n, d0, h0, w0, chunk + 1); else load_B(nn, nd0, nh0, nw0, 0);
        load_A(next_chunk ? chunk + 1 : 0);
      }
      const char* ab = a_lds + buf * A_SLAB + lane * 16;
      const char* bb = b_lds + buf * B_IMG;
      vec8 fa[2], fb[2][2];
      auto load_frag = [&](int tap, int slot) __attribute__((always_inline)) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        fa[slot] = *(const vec8*)(ab + tap * 1024);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[slot][t] = *(const vec8*)(bb + boff[t] + ((kd * IH + kh) * IW + kw) * 16);
      };
      load_frag(0, 0);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int cur = tap & 1;
        if (tap < 26) load_frag(tap + 1, cur ^ 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = Lp<MT>::mfma(fa[cur], fb[cur][t], acc[t]);
      }
      if (stage) {
        store_B(buf ^ 1);
        store_A(buf ^ 1);
      }
      __syncthreads();
    }

    // epilogue: y = act(acc * scale + shift + res_pre) + res_post, fp32 arithmetic, stored as OUT
    const long osample = (long)a.Cout * cstride;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc((char*)a.y + (long)n * osample * OSZ, osample * OSZ);
    const __amdgpu_buffer_rsrc_t pr = dca_rsrc((const char*)(has_pre ? a.res_pre : a.y) + (long)n * osample * OSZ, osample * OSZ);
    const __amdgpu_buffer_rsrc_t qr = dca_rsrc((const char*)(has_post ? a.res_post : a.y) + (long)n * osample * OSZ, osample * OSZ);
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = min(cblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
      sc[r] = has_aff ? a.scale[co] : 1.f;
      sh[r] = has_aff ? a.shift[co] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r0 = (wv * 2 + t) * 2 + (l31 >> 4), d = d0 + (r0 >> 3), h = h0 + (r0 & 7), w = w0 + (l31 & 15);
      const int ok = (int)(d < a.D) & (int)(h < a.H) & (int)(w < a.W);
      const int voff = ((d * a.H + h) * a.W + w + (cblk * 32 + 4 * half) * cstride) * OSZ;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = (r & 3) + 8 * (r >> 2);
        const int okc = ok & (int)(cblk * 32 + cu + 4 * half < a.Cout);
        const int off = voff + cu * cstride * OSZ;
        float v = acc[t][r] * sc[r] + sh[r];
        if (has_pre) v += OUT32 ? dca_bload1(pr, off, okc) : lp_float<MT>(dca_bload_u16(pr, off, okc));
        v = act_apply(v, a.slope);
        if (has_post) v += OUT32 ? dca_bload1(qr, off, okc) : lp_float<MT>(dca_bload_u16(qr, off, okc));
        if constexpr (OUT32) dca_bstore1(yr, v, off, okc);
        else dca_bstore_u16(yr, lp_bits<MT>(v), off, okc);
      }
    }
    n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
  }
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

template <typename MT, bool IN32, bool OUT32>
int launch_lp(const LpArgs& a, bool vec, int gx, int cblks, hipStream_t stream) {
  auto kern = vec ? conv3_lp_kernel<MT, IN32, OUT32, true> : conv3_lp_kernel<MT, IN32, OUT32, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(gx, cblks), dim3(512), LDS_BYTES, stream, a);
  return dca_launch_status();
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
  DCA_REQUIRE((long)Cin * D * H * W * 4 < 0x7ffffff0L && (long)Cout * D * H * W * 4 < 0x7ffffff0L);
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
