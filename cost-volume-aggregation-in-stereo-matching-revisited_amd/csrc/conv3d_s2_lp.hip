// Reduced-precision 3x3x3 convolution with stride 2, padding 1, of the inference path: 2-byte fine input (1/4 resolution),
// ONE native MFMA product per multiply, fp32 accumulation, fp32 epilogue (folded BatchNorm affine + activation), fp32 coarse
// output (the 1/8-resolution interior of a DCA block stays fp32).
//
// Reference operator served (eval mode): `cost_agg.conv1` = convbn_3d(32, 64, 3, stride 2, pad 1) + ReLU
// (models/augment/cva.py:16-17).
//
// y[o] = sum_k w[k] x[2o - 1 + k].  Along W the two fine voxels x[2m], x[2m+1] of a coarse column m are adjacent in
// memory, so the MFMA K axis is (w parity) x (8 input channels): K-step "delta_w = 0" pairs tap kw = 1 (x[2o], parity 0)
// with tap kw = 2 (x[2o+1], parity 1); K-step "delta_w = -1" carries tap kw = 0 (x[2o-1] = parity 1 of column o-1) and a
// zero weight for the unused parity-0 half -- 3 useful taps in 4 half-steps, 75 % of the matrix pipe, and every global
// and LDS access stays contiguous (a strided "every other voxel" gather never appears).  Per 8-channel chunk that is
// 3 x 3 x 2 = 18 K-steps; a wave owns one column tile of 32 coarse positions and both 32-channel output blocks.
// LDS: the chunk's fine 5 x 17 x 34 halo image as [voxel][8 x 2 B] (45 KB) + the chunk's 36 KB of pre-swizzled weight
// fragments; the next chunk's image and weights are fetched into registers while the current chunk's MFMAs run.
#include "dca_common.h"
#include "../../include/dca_hip.h"

typedef __bf16 s2_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TD = 2, TH = 8, TW = 16;                    // coarse tile: 256 positions = 8 column tiles, one per wave
constexpr int FD = 2 * TD + 1, FH = 2 * TH + 1, FW = 2 * TW + 2;   // fine halo: w' = 0 is fine column 2*w0 - 2
constexpr int NVOXF = FD * FH * FW;                       // 2890
constexpr int B_IMG = NVOXF * 16;                         // 46240 B
constexpr int NKS = 18;                                   // K-steps per 8-channel chunk
constexpr int A_CHUNK = NKS * 2 * 1024;                   // 36864 B: [ks][cblk][lane][8 x 2 B]
constexpr int LDS_BYTES = B_IMG + A_CHUNK;                // 83104
constexpr int NROWS = FD * FH;                            // 85 fine rows; a row = 9 aligned quads from fine column 2*w0 - 4
constexpr int NQ = NROWS * 9;                             // 765 quad items of 8 channel loads (b64): two per thread
constexpr int KA = (A_CHUNK / 16 + 511) / 512;            // 5 b128 per thread
static_assert(NQ <= 1024, "two quad items per thread");

struct S2Args {
  const void* x;
  const unsigned short* wx;
  float* y;
  const float* scale;
  const float* shift;
  float slope;
  int N, Cin, Cout, NCH;
  int D, H, W, Do, Ho, Wo;
  int nTD, nTH, nTW;
};

template <typename MT> struct S2;
template <> struct S2<__bf16> {
  typedef s2_bf16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct S2<_Float16> {
  typedef s2_f16x8 vec8;
  static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// W % 4 == 0 and an 8-byte aligned x (the caller checks)
template <typename MT>
__global__ __launch_bounds__(512) void conv3_s2_lp_kernel(S2Args a) {
  typedef typename S2<MT>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;
  char* a_lds = smem + B_IMG;
  __shared__ float aff_lds[128];    // scale[64] | shift[64]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  if (t_begin >= t_end) return;

  const bool has_aff = a.scale != nullptr;
  if (tid < 128) {
    const int co = min(tid & 63, a.Cout - 1);
    aff_lds[tid] = has_aff ? (tid < 64 ? a.scale[co] : a.shift[co]) : (tid < 64 ? 1.f : 0.f);
  }
  const int rr = wv * 2 + (l31 >> 4), dl = rr >> 3, hl = rr & 7, wl = l31 & 15;
  // byte offset of fine voxel (d' = 2dl, h' = 2hl, w' = 2wl + half) in the image; K-steps add (kd, kh, 2 * [delta_w = 0])
  const int boff = ((2 * dl * FH + 2 * hl) * FW + 2 * wl + half) * 16;

  const int cstride = a.D * a.H * a.W;
  const long sample = (long)a.Cin * cstride;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc(a.wx, (long)a.NCH * A_CHUNK);

  float4 ra[KA];
  auto load_A = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      ra[k] = dca_bload4(wr, chunk * A_CHUNK + it * 16, (int)(it < A_CHUNK / 16));
    }
  };
  auto store_A = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      if (it < A_CHUNK / 16) *(float4*)(a_lds + it * 16) = ra[k];
    }
  };
  unsigned rq[2][8][2];
  int item_crd[2];    // fd | fh << 8 | quad << 16
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int it = tid + 512 * k, row = it / 9, q = it - row * 9;
    const int fd = row / FH, fh = row - fd * FH;
    item_crd[k] = (it < NQ) ? (fd | (fh << 8) | (q << 16)) : -1;
  }
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    // channel >= Cin lands beyond the descriptor's range -> zero (partial last chunk)
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc((const char*)a.x + (long)n * sample * 2, sample * 2);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int crd = item_crd[k];
      const int di = 2 * d0 - 1 + (crd & 255), hi = 2 * h0 - 1 + ((crd >> 8) & 255), wi = 2 * w0 - 4 + 4 * ((crd >> 16) & 255);
      const int okv = (int)(crd >= 0) & (int)((unsigned)di < (unsigned)a.D) & (int)((unsigned)hi < (unsigned)a.H) &
                      (int)((unsigned)wi < (unsigned)a.W);       // a quad is inside W or outside as a whole
      const int base = dca_pred_off((chunk * 8 * cstride + (di * a.H + hi) * a.W + wi) * 2, okv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xr, base + j * cstride * 2, 0, 0);
        rq[k][j][0] = v.x; rq[k][j][1] = v.y;
      }
    }
  };
  auto store_B = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int crd = item_crd[k];
      if (crd >= 0) {
        const int q = (crd >> 16) & 255;
        // quad q holds fine columns 2*w0 - 4 + 4q + i = image column w' = 4q - 2 + i; w' in [0, FW)
        const int o_base = (((crd & 255) * FH + ((crd >> 8) & 255)) * FW + 4 * q - 2) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (q == 0 && i < 2) continue;
          constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;
          const unsigned sel = (i & 1) ? HI : LO;
          u32x4 o;
          o.x = __builtin_amdgcn_perm(rq[k][1][i >> 1], rq[k][0][i >> 1], sel);
          o.y = __builtin_amdgcn_perm(rq[k][3][i >> 1], rq[k][2][i >> 1], sel);
          o.z = __builtin_amdgcn_perm(rq[k][5][i >> 1], rq[k][4][i >> 1], sel);
          o.w = __builtin_amdgcn_perm(rq[k][7][i >> 1], rq[k][6][i >> 1], sel);
          *(u32x4*)(b_lds + o_base + 16 * i) = o;
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

  int n, d0, h0, w0;
  decode(t_begin, n, d0, h0, w0);
  load_B(n, d0, h0, w0, 0);
  load_A(0);
  store_B();
  store_A();
  __syncthreads();

  const int ostride = a.Do * a.Ho * a.Wo;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool more_tiles = tile + t_step < t_end;
    int nn = n, nd0 = d0, nh0 = h0, nw0 = w0;
    if (more_tiles) decode(tile + t_step, nn, nd0, nh0, nw0);
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll 1
    for (int chunk = 0; chunk < a.NCH; ++chunk) {
      const bool next_chunk = chunk + 1 < a.NCH, stage = next_chunk || more_tiles;
      if (stage) {
        if (next_chunk) load_B(n, d0, h0, w0, chunk + 1); else load_B(nn, nd0, nh0, nw0, 0);
        load_A(next_chunk ? chunk + 1 : 0);
      }
      const char* ab = a_lds + lane * 16;
      const char* bb = b_lds + boff;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int kd = ks / 6, kh = (ks / 2) % 3, ws = ks & 1;     // ws 0: delta_w = 0 (taps kw 1 | 2), ws 1: delta_w = -1 (tap kw 0)
        const vec8 fb = *(const vec8*)(bb + ((kd * FH + kh) * FW + (ws ? 0 : 2)) * 16);
        const vec8 fa0 = *(const vec8*)(ab + (ks * 2 + 0) * 1024), fa1 = *(const vec8*)(ab + (ks * 2 + 1) * 1024);
        acc[0] = S2<MT>::mfma(fa0, fb, acc[0]);
        acc[1] = S2<MT>::mfma(fa1, fb, acc[1]);
      }
      __syncthreads();     // every wave is done with this chunk's image and weights
      if (stage) {
        store_B();
        store_A();
      }
      __syncthreads();
    }

    // epilogue: y = act(acc * scale + shift), fp32; channel >= Cout is beyond the descriptor's range (dropped)
    const int od = d0 + dl, oh = h0 + hl, ow = w0 + wl;
    const int ok = (int)(od < a.Do) & (int)(oh < a.Ho) & (int)(ow < a.Wo);
    const long osample = (long)a.Cout * ostride;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.y + (long)n * osample, osample * 4);
    const int base = dca_pred_off((((od * a.Ho + oh) * a.Wo + ow) + 4 * half * ostride) * 4, ok);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = act_apply(acc[c][r] * aff_lds[cl] + aff_lds[64 + cl], a.slope);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, base + (c * 32 + (r & 3) + 8 * (r >> 2)) * ostride * 4, 0, 0);
      }
    n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
  }
}

// wx[chunk][ks][cblk][lane][j] (2-byte): lane (r = lane & 31, h = lane >> 5) holds A[row = output channel cblk*32 + r]
// [k = h*8 + j] of K-step ks = (kd*3 + kh)*2 + ws for input channel ci = chunk*8 + j:
//   ws 0: h 0 -> tap kw = 1, h 1 -> tap kw = 2;   ws 1: h 0 -> zero, h 1 -> tap kw = 0.      w: (Cout, Cin, 3, 3, 3) fp32.
template <typename MT>
__global__ void s2_prep_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cin, int Cout,
                                      long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, cblk = (idx >> 9) & 1;
    long t = idx >> 10;
    const int ks = t % NKS;
    const int chunk = (int)(t / NKS);
    const int h = lane >> 5, co = cblk * 32 + (lane & 31), ci = chunk * 8 + j;
    const int kd = ks / 6, kh = (ks / 2) % 3, ws = ks & 1;
    const int kw = ws ? (h ? 0 : -1) : (h ? 2 : 1);
    float v = 0.f;
    if (kw >= 0 && co < Cout && ci < Cin) v = w[((long)co * Cin + ci) * 27 + (kd * 3 + kh) * 3 + kw];
    const MT m = (MT)v;
    dst[idx] = __builtin_bit_cast(unsigned short, m);
  }
}

template <typename MT>
int launch_s2(const S2Args& a, int gx, hipStream_t stream) {
  auto kern = conv3_s2_lp_kernel<MT>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(gx), dim3(512), LDS_BYTES, stream, a);
  return dca_launch_status();
}

}  // namespace

extern "C" long dca_conv3d_s2_lp_weight_bytes(int Cin) {
  if (Cin <= 0) return 0;
  return (long)((Cin + 7) / 8) * A_CHUNK;
}

extern "C" int dca_conv3d_s2_lp_prep_weight(const float* w, void* wx, int Cin, int Cout, int dtype, hipStream_t stream) {
  DCA_REQUIRE(w && wx && Cin > 0 && Cout > 0 && Cout <= 64 && (dtype == DCA_BF16 || dtype == DCA_FP16));
  const long total = dca_conv3d_s2_lp_weight_bytes(Cin) / 2;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == DCA_BF16)
    hipLaunchKernelGGL(s2_prep_weight_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, w, (unsigned short*)wx, Cin, Cout, total);
  else
    hipLaunchKernelGGL(s2_prep_weight_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, w, (unsigned short*)wx, Cin, Cout, total);
  return dca_launch_status();
}

extern "C" int dca_conv3d_s2_lp_forward(const void* x, const void* wx, float* y, const float* scale, const float* shift,
                                        float slope, int N, int Cin, int Cout, int D, int H, int W, int dtype,
                                        hipStream_t stream) {
  DCA_REQUIRE(x && wx && y && N > 0 && Cin > 0 && Cout > 0 && Cout <= 64 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE(dtype == DCA_BF16 || dtype == DCA_FP16);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE(W % 4 == 0 && ((((uintptr_t)x) & 7) == 0) && ((((uintptr_t)wx | (uintptr_t)y) & 15) == 0));
  DCA_REQUIRE((long)(Cin > 8 ? Cin : 8) * D * H * W * 2 < 0x7ffffff0L);
  S2Args a;
  a.x = x; a.wx = (const unsigned short*)wx; a.y = y; a.scale = scale; a.shift = shift; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NCH = (Cin + 7) / 8;
  a.D = D; a.H = H; a.W = W; a.Do = (D + 1) / 2; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;
  DCA_REQUIRE(64L * a.Do * a.Ho * a.Wo * 4 < 0x7ffffff0L);
  a.nTD = cdiv(a.Do, TD); a.nTH = cdiv(a.Ho, TH); a.nTW = cdiv(a.Wo, TW);
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL);
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      ncu = v;
  }
  const int gx = (int)(tiles < ncu ? tiles : ncu);
  return dtype == DCA_BF16 ? launch_s2<__bf16>(a, gx, stream) : launch_s2<_Float16>(a, gx, stream);
}
