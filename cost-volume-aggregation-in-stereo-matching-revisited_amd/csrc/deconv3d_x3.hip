// Transposed 3x3x3 convolution, stride 2, padding 1, output_padding 1 (out = 2 * in), fp32 tensors, on the bf16 matrix
// pipe with fp32-grade accuracy: the exact three-way bf16 split of conv3d_bf16x3.hip (six partial products per fp32
// product, fp32 accumulation; dropped terms <= 2^-23 relative).
//
// Reference operators served: `cost_agg.conv3` = ConvTranspose3d(64, 32, 3, padding=1, output_padding=1, stride=2)
// (models/augment/cva.py:21-29) forward, and the backward-data of `cost_agg.conv1` = Conv3d(32, 64, 3, stride 2, pad 1)
// (cva.py:16-17), which is the same operator with the weight read as [contraction][output].
//
// out[o] += x[i] w[k] with o = 2i - 1 + k.  Per dimension k = 1 feeds the even outputs o = 2m from x[m]; k = 0 / k = 2
// feed the odd outputs o = 2m + 1 from x[m+1] / x[m]: every output-parity class of a coarse position m is a small
// convolution over the 2x2x2 coarse neighbourhood x[m + delta] -- 27 (tap -> class, delta) pairs, no multiply by zero.
//
// Work decomposition: persistent workgroups of 8 waves (one per CU: LDS bound); a tile is 2 x 8 x 16 coarse positions
// = 8 MFMA column tiles, one per wave.  Eight parity classes x 16 accumulator registers do not fit beside the staging
// registers, so a tile runs as two depth-parity passes of four accumulators: pass 0 = even output planes (taps kd = 1),
// pass 1 = odd planes (taps kd = 0 and kd = 2), each followed by its own epilogue (whole output rows either way).
// A "step" is one (16-channel chunk, kd) pair: 9 taps x 6 products = 54 MFMAs per wave on a 27 KB slab of pre-split
// weight fragments (the layout of dca_conv3d_x3_prep_weight, so the forward conv's packed weights are reused) and the
// chunk's coarse 3 x 9 x 17 halo, kept in LDS as FP32 (29 KB, [k half][channel quad][voxel][4 floats]: a lane's 8-channel
// fragment is two conflict-free ds_read_b128) and split into the three bf16 terms IN REGISTERS when a fragment is used
// (11 VALU instructions per channel pair, spread between the MFMAs of all eight waves; splitting at staging time
// instead put ~250 VALU instructions per step on the four waves that own staging items and cost 25 % of the run time).
// Both are double buffered in LDS: in the MIDDLE of step s the registers holding step s+1's data (requested in the
// middle of step s-1, so a whole step of latency budget) are written to the idle buffers and step s+2's loads are
// issued; one barrier per step.
#include "dca_common.h"
#include "../../include/dca_hip.h"

typedef __bf16 dx_bf16x8 __attribute__((ext_vector_type(8)));

#ifndef DX3_NT
#define DX3_NT 1
#endif
// timing ablations (tools/dx3_ablate.sh; results are garbage): 1 no MFMAs, 2 no mid-step staging, 4 no epilogue,
// 8 no per-tap weight-fragment LDS reads, 16 no halo-fragment LDS reads
#ifndef DX3_ABL
#define DX3_ABL 0
#endif

namespace {

constexpr int TD = 2, TH = 8, TW = 16;              // coarse tile: 256 positions = 8 column tiles, one per wave
constexpr int ID = TD + 1, IH = TH + 1, IW = TW + 1;
constexpr int NVOX = ID * IH * IW;                  // 459 coarse halo voxels
constexpr int B_PLANE = NVOX * 16;                  // 7344 B: voxel x 4 fp32 channels
constexpr int B_IMG = 4 * B_PLANE;                  // 29376 B: planes (k half, channel quad) of a 16-channel chunk
constexpr int A_SLAB = 9 * 3 * 1024;                // 9 taps x 3 terms x (64 lanes x 16 B)
constexpr int LDS_BYTES = 2 * B_IMG + 2 * A_SLAB;   // 114048
constexpr int NROWS = 2 * ID * IH;                  // 54 (k half, d, h) rows of IW = 17 voxels: 5 aligned quads
constexpr int NQ = NROWS * 5;                       // 270 quad items (8 x b128 loads each): one per thread
constexpr int KA = (A_SLAB / 16 + 511) / 512;       // 4 b128 per thread (1728 per slab)
static_assert(NQ <= 512, "one quad item per thread");

struct DxArgs {
  const float* x;
  const unsigned short* wx;
  float* y;
  const float* scale;
  const float* shift;
  const float* res_pre;
  const float* res_post;
  float slope;
  int N, Cin, Cout, NCH;
  int Di, Hi, Wi;
  int nTD, nTH, nTW;
};

constexpr int q0(int t) { return t; }
__device__ __forceinline__ unsigned dx_pack2(float a, float b) {   // v_cvt_pk_bf16_f32
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bfx2));
}
// exact three-way split of 8 fp32 values (channel pairs packed): f[0] + f[1] + f[2] == v to 24 bits
__device__ __forceinline__ void dx_split8(const float4 lo, const float4 hi, dx_bf16x8 (&f)[3]) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  u32x4 H, M, L;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float va = v[2 * j], vb = v[2 * j + 1];
    const unsigned h2 = dx_pack2(va, vb);
    const float ra = va - __uint_as_float(h2 << 16), rb = vb - __uint_as_float(h2 & 0xffff0000u);      // exact
    const unsigned m2 = dx_pack2(ra, rb);
    const unsigned l2 = dx_pack2(ra - __uint_as_float(m2 << 16), rb - __uint_as_float(m2 & 0xffff0000u));
    H[j] = h2; M[j] = m2; L[j] = l2;
  }
  f[0] = __builtin_bit_cast(dx_bf16x8, H); f[1] = __builtin_bit_cast(dx_bf16x8, M); f[2] = __builtin_bit_cast(dx_bf16x8, L);
}

// Wi % 4 == 0 and a 16-byte aligned x (the caller checks)
__global__ __launch_bounds__(512) void deconv3_bf16x3_kernel(DxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;                 // two fp32 coarse halo images
  char* a_lds = smem + 2 * B_IMG;     // two weight slabs
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
  // this lane's coarse position inside the tile, and the byte offset of its voxel in a term image (second h-row of the
  // column tile rotated by -1 voxel: its row starts IW*16 = 272 B = 16 B (mod 256) after the first, and ds_read_b128's
  // non-contiguous 16-lane groups would otherwise see 2-way bank conflicts -- see conv3d_bf16x3.hip)
  const int rr = wv * 2 + (l31 >> 4), dl = rr >> 3, hl = rr & 7, wl = (l31 & 16) ? (((l31 & 15) - 1) & 15) : (l31 & 15);
  const int boff = 2 * half * B_PLANE + ((dl * IH + hl) * IW + wl) * 16;   // + B_PLANE: channels 4..7 of the k half

  const int cstride = a.Di * a.Hi * a.Wi;
  const long sample = (long)a.Cin * cstride;
  const int S = 3 * a.NCH;                       // steps per tile
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc(a.wx, (long)a.NCH * 3 * A_SLAB);

  // step s of a tile: s < NCH: pass 0 (kd = 1), chunk s; else pass 1: chunk (s - NCH) / 2, kd = 0 then 2
  auto step_chunk = [&](int s) __attribute__((always_inline)) { return s < a.NCH ? s : (s - a.NCH) >> 1; };
  auto step_kd = [&](int s) __attribute__((always_inline)) { return s < a.NCH ? 1 : ((s - a.NCH) & 1) * 2; };
  auto step_newb = [&](int s) __attribute__((always_inline)) { return s < a.NCH || ((s - a.NCH) & 1) == 0; };

  float4 ra[KA];
  auto load_A = [&](int s) __attribute__((always_inline)) {
    const int base = (step_chunk(s) * 3 + step_kd(s)) * A_SLAB;
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      ra[k] = dca_bload4(wr, base + it * 16, (int)(it < A_SLAB / 16));
    }
  };
  auto store_A = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      if (it < A_SLAB / 16) *(float4*)(a_lds + buf * A_SLAB + it * 16) = ra[k];
    }
  };

  float4 rq[8];
  int item_crd;   // id | ih << 8 | quad << 16 | k half << 24
  {
    const int row = tid / 5, q = tid - row * 5;
    const int kh = row / (ID * IH), rem = row - kh * (ID * IH), id = rem / IH, ih = rem - id * IH;
    item_crd = (tid < NQ) ? (id | (ih << 8) | (q << 16) | (kh << 24)) : -1;
  }
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    // channel >= Cin lands beyond the descriptor's range -> zero (partial last chunk)
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
    const int crd = item_crd;
    const int di = d0 + (crd & 255), hi = h0 + ((crd >> 8) & 255), wi = w0 + 4 * ((crd >> 16) & 255);
    const int c0 = chunk * 16 + ((crd >> 24) & 1) * 8;
    const int okv = (int)(crd >= 0) & (int)(di < a.Di) & (int)(hi < a.Hi) & (int)(wi < a.Wi);   // Wi % 4 == 0
    const int base = dca_pred_off((c0 * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4, okv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, base + j * cstride * 4, 0, 0);
      rq[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto store_B = [&](int buf) __attribute__((always_inline)) {
    const int crd = item_crd;
    if (crd >= 0) {
      const int q = (crd >> 16) & 255;
      char* p0 = b_lds + buf * B_IMG + 2 * ((crd >> 24) & 1) * B_PLANE + (((crd & 255) * IH + ((crd >> 8) & 255)) * IW + 4 * q) * 16;
      *(float4*)p0 = make_float4(rq[0].x, rq[1].x, rq[2].x, rq[3].x);
      *(float4*)(p0 + B_PLANE) = make_float4(rq[4].x, rq[5].x, rq[6].x, rq[7].x);
      if (q < 4) {   // the row has 17 voxels: the fifth quad contributes one
        *(float4*)(p0 + 16) = make_float4(rq[0].y, rq[1].y, rq[2].y, rq[3].y);
        *(float4*)(p0 + B_PLANE + 16) = make_float4(rq[4].y, rq[5].y, rq[6].y, rq[7].y);
        *(float4*)(p0 + 32) = make_float4(rq[0].z, rq[1].z, rq[2].z, rq[3].z);
        *(float4*)(p0 + B_PLANE + 32) = make_float4(rq[4].z, rq[5].z, rq[6].z, rq[7].z);
        *(float4*)(p0 + 48) = make_float4(rq[0].w, rq[1].w, rq[2].w, rq[3].w);
        *(float4*)(p0 + B_PLANE + 48) = make_float4(rq[4].w, rq[5].w, rq[6].w, rq[7].w);
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
  struct Cursor { int tile, s, n, d0, h0, w0; };
  auto advance = [&](Cursor& c) __attribute__((always_inline)) {
    if (c.s + 1 < S) { ++c.s; return; }
    c.s = 0;
    c.tile += t_step;
    if (c.tile < t_end) decode(c.tile, c.n, c.d0, c.h0, c.w0);
  };

  Cursor cur{t_begin, 0, 0, 0, 0, 0}, c1, c2;
  decode(t_begin, cur.n, cur.d0, cur.h0, cur.w0);
  load_B(cur.n, cur.d0, cur.h0, cur.w0, 0);
  load_A(0);
  store_B(0);
  store_A(0);
  c1 = cur;
  advance(c1);       // S >= 3: the second step always exists
  load_A(c1.s);
  if (step_newb(c1.s)) load_B(c1.n, c1.d0, c1.h0, c1.w0, step_chunk(c1.s));
  c2 = c1;
  advance(c2);
  __syncthreads();

  const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
  const int ostride = Do * Ho * Wo;
  const long osample = (long)a.Cout * ostride;
  int bufA = 0, bufB = 0;
  f32x16 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

#pragma unroll 1
  while (cur.tile < t_end) {
    const int kd = step_kd(cur.s);
    const char* ab = a_lds + bufA * A_SLAB + lane * 16;
    const char* bb = b_lds + bufB * B_IMG + boff + (kd == 0 ? IH * IW * 16 : 0);   // taps kd = 0 read x[m + 1] along d
    // the four (h, w) neighbours x[m + delta] of this lane's position, raw fp32; each is split right before the taps
    // that use it.  Taps are visited grouped by neighbour: delta 3 (tap 0), 2 (taps 1, 2), 1 (taps 3, 6), 0 (4, 5, 7, 8).
    float4 raw[4][2];
#pragma unroll
    for (int dlt = 0; dlt < ((DX3_ABL & 16) ? 1 : 4); ++dlt) {
      raw[dlt][0] = *(const float4*)(bb + ((dlt >> 1) * IW + (dlt & 1)) * 16);
      raw[dlt][1] = *(const float4*)(bb + ((dlt >> 1) * IW + (dlt & 1)) * 16 + B_PLANE);
    }
    constexpr int TAPS[9] = {0, 1, 2, 3, 6, 4, 5, 7, 8};
    dx_bf16x8 fa[2][3], fb[3];
#pragma unroll
    for (int term = 0; term < 3; ++term) fa[0][term] = *(const dx_bf16x8*)(ab + (TAPS[0] * 3 + term) * 1024);
    const Cursor nxt = c1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int tap9 = TAPS[i];
      if (i == 4) {   // middle of the step: next step's data -> the idle buffers, request the step after
        if (c1.tile < t_end && !(DX3_ABL & 2)) {
          store_A(bufA ^ 1);
          if (step_newb(c1.s)) store_B(bufB ^ 1);
        }
        if (c2.tile < t_end && !(DX3_ABL & 2)) {
          load_A(c2.s);
          if (step_newb(c2.s)) load_B(c2.n, c2.d0, c2.h0, c2.w0, step_chunk(c2.s));
        }
        c1 = c2;
        advance(c2);
      }
      const int cs = i & 1;
      if (i < 8 && !(DX3_ABL & 8)) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
          fa[cs ^ 1][term] = *(const dx_bf16x8*)(ab + (TAPS[i + 1] * 3 + term) * 1024);
      }
      const int kh = tap9 / 3, kw = tap9 % 3;
      const int pc = (kh != 1) * 2 + (kw != 1);          // output parity class (h, w) of this pass
      const int dlt = (kh == 0) * 2 + (kw == 0);         // coarse neighbour x[m + delta] (h, w)
      if (i == 0 || i == 1 || i == 3 || i == 5) dx_split8(raw[(DX3_ABL & 16) ? 0 : dlt][0], raw[(DX3_ABL & 16) ? 0 : dlt][1], fb);
      // smallest terms first
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
      if (!(DX3_ABL & 1)) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
          acc[pc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(DX3_ABL & 8) ? 0 : cs][PA[q]], fb[PB[q]], acc[pc], 0, 0, 0);
      } else {
        acc[pc][q0(tap9)] += (float)fa[(DX3_ABL & 8) ? 0 : cs][0][0] * (float)fb[0][0];
      }
    }
    __syncthreads();

    if ((cur.s == a.NCH - 1 || cur.s == S - 1) && (!(DX3_ABL & 4) || a.slope == 123.f)) {
      // epilogue of the pass: y = act(acc * scale + shift + res_pre) + res_post on output planes 2m + pd; a lane holds
      // the two w-parities of (h-parity, channel) = 8 contiguous bytes, 16 lanes 128 contiguous bytes
      const int pd = cur.s == S - 1;
      const int md = cur.d0 + dl, mh = cur.h0 + hl, mw = cur.w0 + wl;
      const int ok = (int)(md < a.Di) & (int)(mh < a.Hi) & (int)(mw < a.Wi);
      const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.y + (long)cur.n * osample, osample * 4);
      const __amdgpu_buffer_rsrc_t pr = dca_rsrc((has_pre ? a.res_pre : a.y) + (long)cur.n * osample, osample * 4);
      const __amdgpu_buffer_rsrc_t qr = dca_rsrc((has_post ? a.res_post : a.y) + (long)cur.n * osample, osample * 4);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        // channel >= Cout is beyond the descriptor's range: dropped / read as zero by the hardware
        const int base = dca_pred_off((((2 * md + pd) * Ho + 2 * mh + ph) * Wo + 2 * mw + 4 * half * ostride) * 4, ok);
#pragma unroll
        for (int rc = 0; rc < 16; rc += 8) {
          u32x2 wp[8], wq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) wp[q] = wq[q] = u32x2{0u, 0u};
          if (has_pre) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = rc + q;
              wp[q] = __builtin_amdgcn_raw_buffer_load_b64(pr, base + ((r & 3) + 8 * (r >> 2)) * ostride * 4, 0, 0);
            }
          }
          if (has_post) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = rc + q;
              wq[q] = __builtin_amdgcn_raw_buffer_load_b64(qr, base + ((r & 3) + 8 * (r >> 2)) * ostride * 4, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = rc + q, cl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sc = aff_lds[cl], sh = aff_lds[32 + cl];
            const float v0 = act_apply(acc[ph * 2][r] * sc + sh + __uint_as_float(wp[q].x), a.slope) + __uint_as_float(wq[q].x);
            const float v1 = act_apply(acc[ph * 2 + 1][r] * sc + sh + __uint_as_float(wp[q].y), a.slope) + __uint_as_float(wq[q].y);
            const u32x2 o = {__float_as_uint(v0), __float_as_uint(v1)};
            __builtin_amdgcn_raw_buffer_store_b64(o, yr, base + ((r & 3) + 8 * (r >> 2)) * ostride * 4, 0, DX3_NT ? 2 : 0);
          }
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    }
    bufA ^= 1;
    if (nxt.tile < t_end && step_newb(nxt.s)) bufB ^= 1;
    cur = nxt;
  }
}

}  // namespace

// wx: dca_conv3d_x3_prep_weight(w, wx, A = Cin, B = Cout, src_ab, flip = 0) -- [chunk][tap][term][lane][8 bf16]
extern "C" int dca_deconv3d_x3_forward(const float* x, const void* wx, float* y, const float* scale, const float* shift,
                                       const float* res_pre, const float* res_post, float slope, int N, int Cin, int Cout,
                                       int Di, int Hi, int Wi, hipStream_t stream) {
  DCA_REQUIRE(x && wx && y && N > 0 && Cin > 0 && Cout > 0 && Cout <= 32 && Di > 0 && Hi > 0 && Wi > 0);
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE(Wi % 4 == 0 && ((((uintptr_t)x | (uintptr_t)wx) & 15) == 0));
  DCA_REQUIRE((((uintptr_t)y | (uintptr_t)res_pre | (uintptr_t)res_post) & 7) == 0);
  DCA_REQUIRE((long)(Cin > 8 ? Cin : 8) * Di * Hi * Wi * 4 < 0x7ffffff0L && 32L * 8 * Di * Hi * Wi * 4 < 0x7ffffff0L);
  DxArgs a;
  a.x = x; a.wx = (const unsigned short*)wx; a.y = y; a.scale = scale; a.shift = shift;
  a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NCH = (Cin + 15) / 16;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi;
  a.nTD = cdiv(Di, TD); a.nTH = cdiv(Hi, TH); a.nTW = cdiv(Wi, TW);
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
  hipError_t e = hipFuncSetAttribute((const void*)deconv3_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(deconv3_bf16x3_kernel, dim3(gx), dim3(512), LDS_BYTES, stream, a);
  return dca_launch_status();
}
