// Reduced-precision transposed 3x3x3 convolution, stride 2, padding 1, output_padding 1 (out = 2 * in), of the inference
// path: fp32 coarse input (the 1/8-resolution interior of a DCA block stays fp32), operands rounded to bf16 / fp16, ONE
// native MFMA product per multiply, fp32 accumulation, fp32 epilogue (folded BatchNorm affine, residuals, activation),
// 2-byte fine output and residuals.
//
// Reference operator served (eval mode): `cost_agg.conv3` = ConvTranspose3d(64, 32, 3, padding=1, output_padding=1,
// stride=2) + BatchNorm3d, then ReLU(conv3 + redir(x)) and the caller's outer residual
// (models/augment/cva.py:21-29, models/gwcnet_dca_g.py:229).
//
// out[o] += x[i] w[k] with o = 2i - 1 + k.  Per dimension: k = 1 feeds the even outputs o = 2m from x[m]; k = 0 / k = 2
// feed the odd outputs o = 2m + 1 from x[m+1] / x[m].  So every one of the 8 output-parity classes of a coarse position m
// is a small convolution over the 2x2x2 coarse neighbourhood x[m + delta], delta in {0,1}^3: 27 (tap -> class, delta)
// pairs in total, no multiply by zero anywhere.  A wave owns one MFMA column tile of 32 coarse positions and keeps the 8
// parity classes in 8 accumulators; per 16-channel chunk it reads its 8 neighbourhood fragments once and issues the 27
// MFMAs.  The weight fragments of ALL chunks stay resident in LDS (<= 4 chunks = 108 KB); the coarse 3 x 9 x 17 halo image
// of a chunk is 14 KB, double buffered, staged through registers with the next step's loads issued in the middle of the
// current step (same pipeline as conv3d_lp.hip).  The two w-parities of a coarse position are adjacent fine voxels, so the
// epilogue packs them into one dword per (d-parity, h-parity, channel): 64 coalesced dword stores per lane, no lane
// exchange; residuals come in the same way.
#include "dca_common.h"
#include "../../include/dca_hip.h"

typedef __bf16 dl_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dl_f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TD = 2, TH = 8, TW = 16;              // coarse tile: 256 positions = 8 column tiles, one per wave
constexpr int ID = TD + 1, IH = TH + 1, IW = TW + 1;
constexpr int NVOX = ID * IH * IW;                  // 459 coarse halo voxels
constexpr int B_IMG = 2 * NVOX * 16;                // 14688 B: (k half, voxel) x 8 two-byte channels
constexpr int A_SLAB = 27 * 1024;                   // 27 taps x (64 lanes x 16 B) per 16-channel chunk
constexpr int MAX_NCH = 4;
constexpr int NROWS = 2 * ID * IH;                  // 54 (k half, d, h) rows of IW = 17 voxels: 5 aligned quads (20 values)
constexpr int NQ = NROWS * 5;                       // 270 quad items (8 x b128 loads each): one per thread
constexpr int NB_ITEMS = 2 * NVOX;                  // unaligned path: 918 (k half, voxel) items of 8 x b32
constexpr int KB = (NB_ITEMS + 511) / 512;          // 2
static_assert(NQ <= 512, "one quad item per thread");

struct DlArgs {
  const float* x;
  const unsigned short* wx;
  void* y;
  const float* scale;
  const float* shift;
  const void* res_pre;
  const void* res_post;
  float slope;
  int N, Cin, Cout, NCH;
  int Di, Hi, Wi;
  int nTD, nTH, nTW;
};

template <typename MT> struct Dl;
template <> struct Dl<__bf16> {
  typedef dl_bf16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Dl<_Float16> {
  typedef dl_f16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <typename MT> __device__ __forceinline__ unsigned dl_pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef MT mtx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mtx2));
}
template <typename MT> __device__ __forceinline__ float dl_lo(unsigned w) {
  return (float)__builtin_bit_cast(MT, (unsigned short)(w & 0xffffu));
}
template <typename MT> __device__ __forceinline__ float dl_hi(unsigned w) {
  return (float)__builtin_bit_cast(MT, (unsigned short)(w >> 16));
}

template <typename MT, bool VEC>
__global__ __launch_bounds__(512) void deconv3_lp_kernel(DlArgs a) {
  typedef typename Dl<MT>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;                 // two coarse halo images
  char* a_lds = smem + 2 * B_IMG;     // weight fragments of all chunks
  __shared__ float aff_lds[64];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  if (t_begin >= t_end) return;

  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  if (tid < 64) {
    const int co = min(tid & 31, a.Cout - 1);
    aff_lds[tid] = has_aff ? (tid < 32 ? a.scale[co] : a.shift[co]) : (tid < 32 ? 1.f : 0.f);
  }
  // this lane's coarse position inside the tile, and the byte offset of its voxel in a halo image
  // (second h-row of the column tile rotated by -1 voxel: its row starts IW*16 = 272 B = 16 B (mod 256) after the first,
  // and ds_read_b128's non-contiguous 16-lane groups would otherwise see 2-way bank conflicts -- see conv3d_lp.hip)
  const int rr = wv * 2 + (l31 >> 4), dl = rr >> 3, hl = rr & 7, wl = (l31 & 16) ? (((l31 & 15) - 1) & 15) : (l31 & 15);
  const int boff = (half * NVOX + (dl * IH + hl) * IW + wl) * 16;

  const int cstride = a.Di * a.Hi * a.Wi;
  const long sample = (long)a.Cin * cstride;

  // resident weights: all chunks, loaded once
  {
    const long wbytes = (long)a.NCH * A_SLAB;
    const __amdgpu_buffer_rsrc_t wr = dca_rsrc(a.wx, wbytes);
    for (int it = tid; it < a.NCH * (A_SLAB / 16); it += 512)
      *(float4*)(a_lds + it * 16) = dca_bload4(wr, it * 16, 1);
  }

  unsigned rq[VEC ? 8 : 1][4];
  unsigned rb[VEC ? 1 : KB][8];
  int item_crd[VEC ? 1 : KB];
  if constexpr (VEC) {
    const int row = tid / 5, q = tid - row * 5;
    const int kh = row / (ID * IH), rem = row - kh * (ID * IH), id = rem / IH, ih = rem - id * IH;
    item_crd[0] = (tid < NQ) ? (id | (ih << 8) | (q << 16) | (kh << 24)) : -1;
  } else {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int it = tid + 512 * k;
      const int kh = it / NVOX, v = it - kh * NVOX;
      const int id = v / (IH * IW), rem = v - id * (IH * IW), ih = rem / IW, iw = rem - ih * IW;
      item_crd[k] = (it < NB_ITEMS) ? (id | (ih << 8) | (iw << 16) | (kh << 24)) : -1;
    }
  }
  auto item_off = [&](int crd, int d0, int h0, int w0, int chunk, int& okv) __attribute__((always_inline)) {
    const int di = d0 + (crd & 255), hi = h0 + ((crd >> 8) & 255);
    const int wi = w0 + (VEC ? 4 : 1) * ((crd >> 16) & 255);
    const int c0 = chunk * 16 + ((crd >> 24) & 1) * 8;
    okv = (int)(crd >= 0) & (int)(di < a.Di) & (int)(hi < a.Hi) & (int)(wi < a.Wi);
    return (c0 * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4;
  };
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    // channel >= Cin lands beyond the descriptor's range -> zero (partial last chunk)
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
    if constexpr (VEC) {
      int okv;
      const int base = dca_pred_off(item_off(item_crd[0], d0, h0, w0, chunk, okv), okv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, base + j * cstride * 4, 0, 0);
        rq[j][0] = v.x; rq[j][1] = v.y; rq[j][2] = v.z; rq[j][3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        int okv;
        const int base = dca_pred_off(item_off(item_crd[k], d0, h0, w0, chunk, okv), okv);
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[k][j] = __builtin_amdgcn_raw_buffer_load_b32(xr, base + j * cstride * 4, 0, 0);
      }
    }
  };
  auto store_B = [&](int buf) __attribute__((always_inline)) {
    char* img = b_lds + buf * B_IMG;
    if constexpr (VEC) {
      const int crd = item_crd[0];
      if (crd >= 0) {
        const int q = (crd >> 16) & 255;
        const int o_base = ((((crd >> 24) & 1) * ID + (crd & 255)) * IH + ((crd >> 8) & 255)) * IW * 16 + 4 * q * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (q == 4 && i > 0) continue;            // the row has 17 voxels: the fifth quad contributes one
          u32x4 o;
          o.x = dl_pack2<MT>(__uint_as_float(rq[0][i]), __uint_as_float(rq[1][i]));
          o.y = dl_pack2<MT>(__uint_as_float(rq[2][i]), __uint_as_float(rq[3][i]));
          o.z = dl_pack2<MT>(__uint_as_float(rq[4][i]), __uint_as_float(rq[5][i]));
          o.w = dl_pack2<MT>(__uint_as_float(rq[6][i]), __uint_as_float(rq[7][i]));
          *(u32x4*)(img + o_base + 16 * i) = o;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int crd = item_crd[k];
        if (crd >= 0) {
          const int o = ((((crd >> 24) & 1) * ID + (crd & 255)) * IH + ((crd >> 8) & 255)) * IW * 16 + ((crd >> 16) & 255) * 16;
          u32x4 w;
          w.x = dl_pack2<MT>(__uint_as_float(rb[k][0]), __uint_as_float(rb[k][1]));
          w.y = dl_pack2<MT>(__uint_as_float(rb[k][2]), __uint_as_float(rb[k][3]));
          w.z = dl_pack2<MT>(__uint_as_float(rb[k][4]), __uint_as_float(rb[k][5]));
          w.w = dl_pack2<MT>(__uint_as_float(rb[k][6]), __uint_as_float(rb[k][7]));
          *(u32x4*)(img + o) = w;
        }
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
  struct Cursor { int tile, chunk, n, d0, h0, w0; };
  auto advance = [&](Cursor& c) __attribute__((always_inline)) {
    if (c.chunk + 1 < a.NCH) { ++c.chunk; return; }
    c.chunk = 0;
    c.tile += t_step;
    if (c.tile < t_end) decode(c.tile, c.n, c.d0, c.h0, c.w0);
  };

  int n, d0, h0, w0;
  decode(t_begin, n, d0, h0, w0);
  load_B(n, d0, h0, w0, 0);
  store_B(0);
  Cursor c1{t_begin, 0, n, d0, h0, w0}, c2;
  advance(c1);
  c2 = c1;
  if (c1.tile < t_end) {
    load_B(c1.n, c1.d0, c1.h0, c1.w0, c1.chunk);
    advance(c2);
  }
  __syncthreads();

  const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
  const int ostride = Do * Ho * Wo;
  int buf = 0;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
#pragma unroll 1
    for (int chunk = 0; chunk < a.NCH; ++chunk, buf ^= 1) {
      const char* ab = a_lds + chunk * A_SLAB + lane * 16;
      const char* bb = b_lds + buf * B_IMG + boff;
      vec8 fb[8];
#pragma unroll
      for (int dlt = 0; dlt < 8; ++dlt)
        fb[dlt] = *(const vec8*)(bb + ((((dlt >> 2) & 1) * IH + ((dlt >> 1) & 1)) * IW + (dlt & 1)) * 16);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        if (tap == 12) {   // middle of the step: pack the next step's tile into the other image, request the one after
          if (c1.tile < t_end) store_B(buf ^ 1);
          if (c2.tile < t_end) load_B(c2.n, c2.d0, c2.h0, c2.w0, c2.chunk);
          c1 = c2;
          advance(c2);
        }
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const int pc = ((kd != 1) * 2 + (kh != 1)) * 2 + (kw != 1);          // output parity class
        const int dlt = ((kd == 0) * 2 + (kh == 0)) * 2 + (kw == 0);         // coarse neighbour x[m + delta]
        const vec8 fa = *(const vec8*)(ab + tap * 1024);
        acc[pc] = Dl<MT>::mfma(fa, fb[dlt], acc[pc]);
      }
      __syncthreads();
    }

    // epilogue: y = act(acc * scale + shift + res_pre) + res_post; one dword = the two w-parities of (d-par, h-par, ch)
    const int md = d0 + dl, mh = h0 + hl, mw = w0 + wl;
    const int ok = (int)(md < a.Di) & (int)(mh < a.Hi) & (int)(mw < a.Wi);
    const long osample = (long)a.Cout * ostride;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc((char*)a.y + (long)n * osample * 2, osample * 2);
    const __amdgpu_buffer_rsrc_t pr = dca_rsrc((const char*)(has_pre ? a.res_pre : a.y) + (long)n * osample * 2, osample * 2);
    const __amdgpu_buffer_rsrc_t qr = dca_rsrc((const char*)(has_post ? a.res_post : a.y) + (long)n * osample * 2, osample * 2);
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        // channel >= Cout is beyond the descriptor's range: dropped / read as zero by the hardware
        const int base = dca_pred_off((((2 * md + pd) * Ho + 2 * mh + ph) * Wo + 2 * mw + 4 * half * ostride) * 2, ok);
        const int pc0 = (pd * 2 + ph) * 2;
#pragma unroll
        for (int rc = 0; rc < 16; rc += 8) {   // eight registers at a time: the 128 accumulators leave little room
          unsigned wp[8], wq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) wp[q] = wq[q] = 0u;
          if (has_pre) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = rc + q;
              wp[q] = __builtin_amdgcn_raw_buffer_load_b32(pr, base + ((r & 3) + 8 * (r >> 2)) * ostride * 2, 0, 0);
            }
          }
          if (has_post) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = rc + q;
              wq[q] = __builtin_amdgcn_raw_buffer_load_b32(qr, base + ((r & 3) + 8 * (r >> 2)) * ostride * 2, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = rc + q, cl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sc = aff_lds[cl], sh = aff_lds[32 + cl];
            const float v0 = act_apply(acc[pc0][r] * sc + sh + dl_lo<MT>(wp[q]), a.slope) + dl_lo<MT>(wq[q]);
            const float v1 = act_apply(acc[pc0 + 1][r] * sc + sh + dl_hi<MT>(wp[q]), a.slope) + dl_hi<MT>(wq[q]);
            __builtin_amdgcn_raw_buffer_store_b32(dl_pack2<MT>(v0, v1), yr, base + ((r & 3) + 8 * (r >> 2)) * ostride * 2, 0, 0);
          }
        }
      }
    if (tile + t_step < t_end) decode(tile + t_step, n, d0, h0, w0);
  }
}

template <typename MT>
int launch_dl(const DlArgs& a, bool vec, int gx, hipStream_t stream) {
  auto kern = vec ? deconv3_lp_kernel<MT, true> : deconv3_lp_kernel<MT, false>;
  const int lds = 2 * B_IMG + a.NCH * A_SLAB;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, stream, a);
  return dca_launch_status();
}

}  // namespace

// wx: dca_conv3d_lp_prep_weight(w, wx, Cin, Cout, src_ab = 1, flip = 0, dtype) of the ConvTranspose3d weight (Cin,Cout,3,3,3)
extern "C" int dca_deconv3d_lp_forward(const float* x, const void* wx, void* y, const float* scale, const float* shift,
                                       const void* res_pre, const void* res_post, float slope, int N, int Cin, int Cout,
                                       int Di, int Hi, int Wi, int dtype, hipStream_t stream) {
  DCA_REQUIRE(x && wx && y && N > 0 && Cin > 0 && Cout > 0 && Cout <= 32 && Di > 0 && Hi > 0 && Wi > 0);
  DCA_REQUIRE(dtype == DCA_BF16 || dtype == DCA_FP16);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE((Cin + 15) / 16 <= MAX_NCH);
  DCA_REQUIRE((long)(Cin > 8 ? Cin : 8) * Di * Hi * Wi * 4 < 0x7ffffff0L && 32L * 8 * Di * Hi * Wi * 2 < 0x7ffffff0L);
  DCA_REQUIRE((((uintptr_t)wx | (uintptr_t)y | (uintptr_t)res_pre | (uintptr_t)res_post) & 15) == 0);
  DlArgs a;
  a.x = x; a.wx = (const unsigned short*)wx; a.y = y; a.scale = scale; a.shift = shift;
  a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NCH = (Cin + 15) / 16;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi;
  a.nTD = cdiv(Di, TD); a.nTH = cdiv(Hi, TH); a.nTW = cdiv(Wi, TW);
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL);
  const bool vec = (Wi % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      ncu = v;
  }
  const int gx = (int)(tiles < ncu ? tiles : ncu);
  return dtype == DCA_BF16 ? launch_dl<__bf16>(a, vec, gx, stream) : launch_dl<_Float16>(a, vec, gx, stream);
}
