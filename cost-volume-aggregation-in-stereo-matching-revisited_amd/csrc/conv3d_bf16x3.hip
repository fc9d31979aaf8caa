// 3x3x3 stride-1 convolution on the bf16 matrix pipe with fp32-grade accuracy ("bf16x3" split emulation).
//
// Every fp32 operand is split exactly into three bf16 terms, x = h + m + l (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m); the residuals are exact in fp32, so the three terms carry 24 significand bits), and a product
// w*x is evaluated as the six partial products whose weight is >= 2^-16:
//     w_h x_h + (w_h x_m + w_m x_h) + (w_h x_l + w_m x_m + w_l x_h)
// Each partial product is exact in the MFMA's fp32 accumulator; the dropped terms (w_m x_l, w_l x_m, w_l x_l) are
// <= 2^-23 relative, the size of one fp32 rounding.  No range restriction: bf16 has the fp32 exponent.
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace sixteen v_mfma_f32_32x32x2f32 (64 cycles each) per
// 32 input channels, tap and 32-voxel tile: 2.67x fewer matrix-pipe cycles than the fp32 kernel of conv3d_mfma.hip,
// which is pinned to its matrix-pipe ceiling.
//
// Reference operators served: nn.Conv3d(k=3, s=1, p=1) of convbn_3d (models/submodule.py:121-124) in dres0/dres1,
// Multi_Aggregation and the cva blocks (models/augment/cva.py:13-55), and their backward-data.
//
// Work decomposition: one workgroup (8 waves) per 4 x 8 x 16 output tile (512 voxels = 16 MFMA column tiles, two per
// wave) and 32 output channels; one workgroup per CU (LDS bound), two waves per SIMD.  Input channels go through LDS
// in chunks of 16 (one MFMA K): the 6 x 10 x 18 halo tile of the chunk, pre-split into three bf16 images laid out
// [term][k half][voxel][8 bf16] so that a lane's B fragment (8 consecutive k of its voxel) is one ds_read_b128 and 16
// consecutive lanes read 256 contiguous bytes.  The weights arrive pre-split and pre-swizzled into MFMA A fragments
// (x3_prep_weight_kernel) and stream through a double-buffered LDS slab of 9 taps (one kd plane) per phase, so a
// channel chunk is three phases of 108 MFMAs per wave with one barrier each; the next slab / next halo tile are
// fetched into registers (hardware-predicated buffer loads) while the current phase's MFMAs run.
#include "dca_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TD = 4, TH = 8, TW = 16;
constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
constexpr int NVOX = ID * IH * IW;                 // 1080 halo voxels
constexpr int B_TERM = 2 * NVOX * 16;              // bytes of one bf16 term image (2 k halves x voxels x 16 B)
constexpr int B_BYTES = 3 * B_TERM;                // 103680
constexpr int A_SLAB = 9 * 3 * 1024;               // 9 taps x 3 terms x (64 lanes x 16 B)
constexpr int LDS_BYTES = B_BYTES + 2 * A_SLAB;    // 158976 of the CU's 163840
constexpr int NB_ITEMS = 2 * NVOX;                 // (k half, voxel) staging items of 8 channels
constexpr int KB = (NB_ITEMS + 511) / 512;         // 5
constexpr int NA_ITEMS = A_SLAB / 16;              // 1728 b128 per slab
constexpr int KA = (NA_ITEMS + 511) / 512;         // 4

struct X3Args {
  const float* x;
  const unsigned short* wx;
  float* y;
  const float* scale;
  const float* shift;
  const float* res_pre;
  const float* res_post;
  float slope;
  int N, Cin, Cout, NCH;
  int D, H, W;
  int nTD, nTH, nTW;
};

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  const float r1 = v - (float)h;   // exact
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;  // exact
  l = (__bf16)r2;
}

__global__ __launch_bounds__(512) void conv3_bf16x3_kernel(X3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;
  char* a_lds = smem + B_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tw = bid % a.nTW; bid /= a.nTW;
  const int th = bid % a.nTH; bid /= a.nTH;
  const int td = bid % a.nTD;
  const int n = bid / a.nTD;
  const int cblk = blockIdx.y;
  const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-lane byte offset of the lane's voxel inside a term image, for its two column tiles
  int boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (wv * 2 + t) * 2 + (l31 >> 4), dl = r >> 3, hl = r & 7;
    boff[t] = (half * NVOX + (dl * IH + hl) * IW + (l31 & 15)) * 16;
  }

  const int cstride = a.D * a.H * a.W;
  const long sample = (long)a.Cin * cstride;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
  const int P = a.NCH * 3;
  const long wbytes = (long)P * A_SLAB;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc((const char*)a.wx + (long)cblk * wbytes, wbytes);

  float4 ra[KA];
  float rb[KB][8];
  auto load_A = [&](int p) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      ra[k] = dca_bload4(wr, p * A_SLAB + it * 16, (int)(it < NA_ITEMS));
    }
  };
  auto store_A = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      if (it < NA_ITEMS) *(float4*)(a_lds + buf * A_SLAB + it * 16) = ra[k];
    }
  };
  auto load_B = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int it = tid + 512 * k;
      const int kh = it / NVOX, v = it - kh * NVOX;
      const int id = v / (IH * IW), rem = v - id * (IH * IW), ih = rem / IW, iw = rem - ih * IW;
      const int di = d0 - 1 + id, hi = h0 - 1 + ih, wi = w0 - 1 + iw;
      const int okv = (int)(it < NB_ITEMS) & (int)((unsigned)di < (unsigned)a.D) & (int)((unsigned)hi < (unsigned)a.H) &
                      (int)((unsigned)wi < (unsigned)a.W);
      const int sp = (di * a.H + hi) * a.W + wi;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = chunk * 16 + kh * 8 + j;
        rb[k][j] = dca_bload1(xr, (c * cstride + sp) * 4, okv & (int)(c < a.Cin));
      }
    }
  };
  auto store_B = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int it = tid + 512 * k;
      bf16x8 hv, mv, lv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 h, m, l;
        split3(rb[k][j], h, m, l);
        hv[j] = h; mv[j] = m; lv[j] = l;
      }
      if (it < NB_ITEMS) {
        *(bf16x8*)(b_lds + it * 16) = hv;
        *(bf16x8*)(b_lds + B_TERM + it * 16) = mv;
        *(bf16x8*)(b_lds + 2 * B_TERM + it * 16) = lv;
      }
    }
  };

  load_B(0);
  load_A(0);
  store_B();
  store_A(0);
  if (P > 1) load_A(1);
  __syncthreads();

#pragma unroll 1
  for (int p = 0; p < P; ++p) {
    const int chunk = p / 3, kd = p - chunk * 3, buf = p & 1;
    const bool next_chunk = (kd == 2) && (chunk + 1 < a.NCH);
    if (p + 1 < P) store_A(buf ^ 1);  // slab p+1 (in registers since the previous phase) -> the buffer phase p-1 used
    if (p + 2 < P) load_A(p + 2);
    if (next_chunk) load_B(chunk + 1);

    const char* ab = a_lds + buf * A_SLAB + lane * 16;
    const char* bb = b_lds + kd * (IH * IW * 16);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap9 = kh * 3 + kw;
        const bf16x8 ah = *(const bf16x8*)(ab + (tap9 * 3 + 0) * 1024);
        const bf16x8 am = *(const bf16x8*)(ab + (tap9 * 3 + 1) * 1024);
        const bf16x8 al = *(const bf16x8*)(ab + (tap9 * 3 + 2) * 1024);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* bp = bb + boff[t] + (kh * IW + kw) * 16;
          const bf16x8 bh = *(const bf16x8*)(bp);
          const bf16x8 bm = *(const bf16x8*)(bp + B_TERM);
          const bf16x8 bl = *(const bf16x8*)(bp + 2 * B_TERM);
          // smallest terms first
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        }
      }
    }
    if (next_chunk) {
      __syncthreads();  // every wave is done reading the halo tile of this chunk
      store_B();
    }
    __syncthreads();
  }

  // Epilogue (same contract as conv3d_mfma.hip): y = act(acc * scale + shift + res_pre) + res_post
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  float sc[16], sh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = min(cblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
    sc[r] = has_aff ? a.scale[co] : 1.f;
    sh[r] = has_aff ? a.shift[co] : 0.f;
  }
  const long plane = cstride;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r0 = (wv * 2 + t) * 2 + (l31 >> 4), d = d0 + (r0 >> 3), h = h0 + (r0 & 7), w = w0 + (l31 & 15);
    const bool ok = d < a.D && h < a.H && w < a.W;
    const long base = (long)n * a.Cout * plane + ((long)(ok ? d : 0) * a.H + (ok ? h : 0)) * a.W + (ok ? w : 0);
    float rp[16], rq[16];
    if (has_pre) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rp[r] = a.res_pre[base + min(cblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1) * plane];
    }
    if (has_post) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rq[r] = a.res_post[base + min(cblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1) * plane];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[t][r] * sc[r] + sh[r];
      if (has_pre) v += rp[r];
      v = act_apply(v, a.slope);
      if (has_post) v += rq[r];
      if (ok && co < a.Cout) a.y[base + co * plane] = v;
    }
  }
}

// wx[cblk][chunk][tap][term][lane][j] (bf16): lane (r = lane & 31, h = lane >> 5) holds A[row = output channel
// cblk*32 + r][k = input channel chunk*16 + 8h + j] of the tap, split into term 0/1/2 = h/m/l; zero padded.
// Source indexing as dca_conv3d_prep_weight: src_ab ? src[a][b][27] : src[b][a][27]; flip reverses the tap order.
__global__ void x3_prep_weight_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int A, int Bn,
                                      int NCH, int src_ab, int flip, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    long t = idx >> 9;
    const int term = t % 3; t /= 3;
    const int tap = t % 27; t /= 27;
    const int chunk = t % NCH;
    const int cblk = (int)(t / NCH);
    const int bi = cblk * 32 + (lane & 31), ai = chunk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (ai < A && bi < Bn) {
      const int st = flip ? 26 - tap : tap;
      v = src_ab ? src[((long)ai * Bn + bi) * 27 + st] : src[((long)bi * A + ai) * 27 + st];
    }
    __bf16 h, m, l;
    split3(v, h, m, l);
    const __bf16 o = term == 0 ? h : (term == 1 ? m : l);
    dst[idx] = __builtin_bit_cast(unsigned short, o);
  }
}

}  // namespace

extern "C" long dca_conv3d_x3_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (long)((Cout + 31) / 32) * ((Cin + 15) / 16) * 27 * 3 * 1024;
}

extern "C" int dca_conv3d_x3_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip,
                                         hipStream_t stream) {
  DCA_REQUIRE(w && wx && A > 0 && B > 0);
  const int NCH = (A + 15) / 16;
  const long total = dca_conv3d_x3_weight_bytes(A, B) / 2;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(x3_prep_weight_kernel, dim3(grid), dim3(256), 0, stream, w, (unsigned short*)wx, A, B, NCH, src_ab,
                     flip, total);
  return dca_launch_status();
}

extern "C" int dca_conv3d_x3_forward(const float* x, const void* wx, float* y, const float* scale, const float* shift,
                                     const float* res_pre, const float* res_post, float slope, int N, int Cin, int Cout,
                                     int D, int H, int W, hipStream_t stream) {
  DCA_REQUIRE(x && wx && y && N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE((long)Cin * D * H * W * 4 < 0x7ffffff0L);  // 32-bit byte offsets inside one sample
  DCA_REQUIRE((((uintptr_t)wx) & 15) == 0);
  X3Args a;
  a.x = x; a.wx = (const unsigned short*)wx; a.y = y;
  a.scale = scale; a.shift = shift; a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NCH = (Cin + 15) / 16;
  a.D = D; a.H = H; a.W = W;
  a.nTD = cdiv(D, TD); a.nTH = cdiv(H, TH); a.nTW = cdiv(W, TW);
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL && (Cout + 31) / 32 <= 65535);
  // per device, so not cached in a static
  hipError_t e = hipFuncSetAttribute((const void*)conv3_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(conv3_bf16x3_kernel, dim3((unsigned)tiles, (Cout + 31) / 32), dim3(512), LDS_BYTES, stream, a);
  return dca_launch_status();
}
