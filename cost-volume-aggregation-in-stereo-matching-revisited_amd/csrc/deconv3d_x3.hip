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
// chunk's coarse 3 x 9 x 17 halo, kept in LDS as FP32 (29 KB, [k half][channel quad][voxel][4 floats]: a lane's
// 8-channel fragment is two conflict-free ds_read_b128); both double buffered.
//
// Schedule: the two waves that share a SIMD (wave w and w + 4) share its matrix pipe AND its vector issue, so running
// all eight waves through "54 MFMAs, then stage" in lockstep leaves the matrix pipe idle while everybody stages
// (measured: MFMA core 411 us + staging 180 us + epilogues 125 us = the whole 706 us at batch 4, 30 % MFMA busy).
// Instead the workgroup runs as two half-groups in opposite phases (MI355X_MICROARCH.md, "Two waves per SIMD"): in
// slot 2s waves 0-3 issue the MFMAs of step s while waves 4-7 are in their "load slot" -- write their half of step
// s+1's data (fetched into registers one slot earlier) into the idle LDS buffers and request their half of step s+2
// -- and in slot 2s+1 the roles swap; one barrier per slot.  In-kernel s_memtime stamps (DX3_STAMP, tools/dx3_stamps.py)
// showed what a load slot can afford: beside the partner's MFMA stream a vector instruction of the loading wave takes
// ~18 cycles, so splitting the halo into bf16 terms there (90 instructions) made the load slot (2500 cycles) longer
// than the compute slot (2250); the computing wave's own vector instructions between its MFMAs are nearly free.  Hence
// the split into the three bf16 terms happens IN REGISTERS in the compute slot, per neighbour fragment, interleaved
// with the MFMAs (11 instructions per channel pair), the pass epilogue runs at the end of the compute slot, and a load
// slot is nothing but waits, register moves and LDS stores.
#include "dca_common.h"
#include "bn_fused_stats.h"
#include "../../include/dca_hip.h"

typedef __bf16 dx_bf16x8 __attribute__((ext_vector_type(8)));

// non-temporal output stores: off, as in conv3d_bf16x3.hip (the consumer finds y in the infinity cache)
#ifndef DX3_NT
#define DX3_NT 0
#endif
#ifndef DX3_SOFF
#define DX3_SOFF 1
#endif

// DX3_STAMP (debug build, tools/dx3_stamps.py): res_post is reinterpreted as an unsigned long long buffer that receives
// s_memtime stamps of the first 96 slots of workgroup 0, waves 0 and 4
#ifndef DX3_STAMP
#define DX3_STAMP 0
#endif
#if DX3_STAMP
#define DX3_MARK(i) do { if (stamp_on && k < 96) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[(grp * 96 + k) * 6 + (i)] = t_; } } while (0)
#else
#define DX3_MARK(i) do { } while (0)
#endif

namespace {

constexpr int TD = 2, TH = 8, TW = 16;              // coarse tile: 256 positions = 8 column tiles, one per wave
constexpr int ID = TD + 1, IH = TH + 1, IW = TW + 1;
constexpr int NVOX = ID * IH * IW;                  // 459 coarse halo voxels
constexpr int B_PLANE = NVOX * 16;                  // 7344 B: voxel x 4 fp32 channels
constexpr int B_IMG = 4 * B_PLANE;                  // 29376 B: planes (k half, channel quad) of a 16-channel chunk
constexpr int A_SLAB = 9 * 3 * 1024;                // 9 taps x 3 terms x (64 lanes x 16 B)
constexpr int LDS_BYTES = 2 * B_IMG + 2 * A_SLAB;   // 114048
constexpr int NROWS = 2 * ID * IH;                  // 54 (k half, d, h) rows of IW = 17 voxels: 9 aligned pairs
constexpr int NPAIR = NROWS * 9;                    // 486 pair items (8 x b64 loads each)
constexpr int HALF_PAIR = NPAIR / 2;                // 243 per half-group: one per thread
constexpr int HALF_A = A_SLAB / 16 / 2;             // 864 b128 of a slab per half-group
constexpr int KA = (HALF_A + 255) / 256;            // 4 per thread
static_assert(HALF_PAIR <= 256 && NPAIR % 2 == 0, "one pair item per thread");

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
  double* stat_part;         // STATS: one partial {K, n, s, q} per (channel, workgroup): bn_fused_stats.h
};

constexpr int STAT_LDS = 8 * FS_WAVE_FLOATS * 4;

__device__ __forceinline__ unsigned dx_pack2(float a, float b) {   // v_cvt_pk_bf16_f32
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bfx2));
}
__device__ __forceinline__ float dx_sub(float a, float b) {   // plain v_sub_f32: hipcc's SLP pass would pair these into
  float r;                                                     // v_pk_add_f32, which is slow beside MFMAs
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// exact three-way split of channel pair j of a lane's 8 fp32 values (lo = channels 0..3, hi = 4..7) into dword j of the
// three bf16 fragments: H + M + L == v to 24 bits
__device__ __forceinline__ void dx_split_h(const float4 lo, const float4 hi, int j, u32x4& H) {
  const float va = j == 0 ? lo.x : (j == 1 ? lo.z : (j == 2 ? hi.x : hi.z));
  const float vb = j == 0 ? lo.y : (j == 1 ? lo.w : (j == 2 ? hi.y : hi.w));
  H[j] = dx_pack2(va, vb);
}
__device__ __forceinline__ void dx_split_ml(const float4 lo, const float4 hi, int j, const u32x4& H, u32x4& M, u32x4& L) {
  const float va = j == 0 ? lo.x : (j == 1 ? lo.z : (j == 2 ? hi.x : hi.z));
  const float vb = j == 0 ? lo.y : (j == 1 ? lo.w : (j == 2 ? hi.y : hi.w));
  const unsigned h2 = H[j];
  const float ra = dx_sub(va, __uint_as_float(h2 << 16)), rb = dx_sub(vb, __uint_as_float(h2 & 0xffff0000u));      // exact
  const unsigned m2 = dx_pack2(ra, rb);
  M[j] = m2;
  L[j] = dx_pack2(dx_sub(ra, __uint_as_float(m2 << 16)), dx_sub(rb, __uint_as_float(m2 & 0xffff0000u)));
}
__device__ __forceinline__ void dx_split_pair(const float4 lo, const float4 hi, int j, u32x4& H, u32x4& M, u32x4& L) {
  dx_split_h(lo, hi, j, H);
  dx_split_ml(lo, hi, j, H, M, L);
}

// Wi % 4 == 0 and a 16-byte aligned x (the caller checks).  STATS: the (plain) output feeds a training-mode BatchNorm and
// the kernel also emits the per-channel partial sums of (y - K_c), (y - K_c)^2 (see conv3d_bf16x3.hip).
template <bool STATS>
__global__ __launch_bounds__(512) void deconv3_bf16x3_kernel(DxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;                 // two fp32 coarse halo images
  char* a_lds = smem + 2 * B_IMG;     // two weight slabs
  float* stat_lds = (float*)(smem + LDS_BYTES);   // STATS only (the launch adds STAT_LDS bytes)
  __shared__ float aff_lds[64];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the slot role and the cursors stay scalar
  const int grp = wv >> 2, gt = tid & 255;
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  float* stat_w = stat_lds + wv * FS_WAVE_FLOATS;   // this wave's slots
  bool stat_first = true;
  float st_s[STATS ? 16 : 1], st_q[STATS ? 16 : 1], st_n = 0.f;
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st_s[r] = st_q[r] = 0.f;
    for (int i = tid; i < 8 * FS_WAVE_FLOATS; i += 512) stat_lds[i] = 0.f;
    __syncthreads();
  }
  auto flush_stats = [&]() __attribute__((always_inline)) {
    // per-lane running sums -> wave-private slots (DPP reduction over the wave half), then the workgroup's partial
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float rs = fs_half_sum(st_s[r]), rq2 = fs_half_sum(st_q[r]);
      if ((lane & 31) == 0) {
        fs_slot(stat_w, half, r)[1] = rs;
        fs_slot(stat_w, half, r)[2] = rq2;
      }
    }
    const float rn = fs_half_sum(st_n);
    if ((lane & 31) == 0) stat_w[96 + half] = rn;
    __syncthreads();
    fs_flush(stat_lds, 8, tid, 0, a.Cout, a.stat_part, gridDim.x, blockIdx.x);
  };
  if (t_begin >= t_end) {
    if constexpr (STATS) flush_stats();
    return;
  }

#if DX3_STAMP
  unsigned long long* stamps = (unsigned long long*)a.res_post;
  a.res_post = nullptr;
  const bool stamp_on = blockIdx.x == 0 && (wv == 0 || wv == 4);
#endif
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
  const bool cin_full = (a.Cin & 15) == 0, cout_full = a.Cout == 32;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc(a.wx, (long)a.NCH * 3 * A_SLAB);

  // step s of a tile: s < NCH: pass 0 (kd = 1), chunk s; else pass 1: chunk (s - NCH) / 2, kd = 0 then 2
  auto step_chunk = [&](int s) __attribute__((always_inline)) { return s < a.NCH ? s : (s - a.NCH) >> 1; };
  auto step_kd = [&](int s) __attribute__((always_inline)) { return s < a.NCH ? 1 : ((s - a.NCH) & 1) * 2; };
  auto step_newb = [&](int s) __attribute__((always_inline)) { return s < a.NCH || ((s - a.NCH) & 1) == 0; };

  // ---- staging: each half-group moves half of a step's data, global -> registers -> (split) -> LDS
  float4 ra[KA];
  int a_voff[KA];      // gt * 16, or out of range for the pieces beyond the half slab
#pragma unroll
  for (int k = 0; k < KA; ++k) a_voff[k] = dca_pred_off(gt * 16, (int)(gt + 256 * k < HALF_A));
  auto load_A = [&](int s) __attribute__((always_inline)) {
    // the slab / piece part of every address is a scalar offset: no vector instruction per load (a vector instruction of
    // a loading wave costs ~18 cycles beside the partner's MFMAs)
    const int sbase = (step_chunk(s) * 3 + step_kd(s)) * A_SLAB + grp * HALF_A * 16;
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const u32x4 v = DX3_SOFF ? __builtin_amdgcn_raw_buffer_load_b128(wr, a_voff[k], sbase + 256 * 16 * k, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(wr, a_voff[k] + sbase + 256 * 16 * k, 0, 0);
      ra[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto store_A = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = gt + 256 * k;
      if (it < HALF_A) *(float4*)(a_lds + buf * A_SLAB + (grp * HALF_A + it) * 16) = ra[k];
    }
  };
  u32x2 rq[8];
  int item_crd;   // id | ih << 8 | pair << 16 | k half << 24
  {
    const int it = grp * HALF_PAIR + gt, row = it / 9, p = it - row * 9;
    const int kh = row / (ID * IH), rem = row - kh * (ID * IH), id = rem / IH, ih = rem - id * IH;
    item_crd = (gt < HALF_PAIR) ? (id | (ih << 8) | (p << 16) | (kh << 24)) : -1;
  }
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    // channel >= Cin lands beyond the descriptor's range -> zero (partial last chunk)
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
    const int crd = item_crd;
    const int di = d0 + (crd & 255), hi = h0 + ((crd >> 8) & 255), wi = w0 + 2 * ((crd >> 16) & 255);
    const int c0 = chunk * 16 + ((crd >> 24) & 1) * 8;
    const int okv = (int)(crd >= 0) & (int)(di < a.Di) & (int)(hi < a.Hi) & (int)(wi < a.Wi);   // Wi even: a pair is in or out
    const int base = dca_pred_off((c0 * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4, okv);
    // the channel part of the address as a scalar offset (no vector instruction per load) -- but scalar offsets are
    // excluded from the hardware range check, which a partial last chunk relies on, so only when Cin fills its chunks
    if (cin_full && DX3_SOFF) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rq[j] = __builtin_amdgcn_raw_buffer_load_b64(xr, base, j * cstride * 4, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) rq[j] = __builtin_amdgcn_raw_buffer_load_b64(xr, base + j * cstride * 4, 0, 0);
    }
  };
  auto store_B = [&](int buf) __attribute__((always_inline)) {
    const int crd = item_crd;
    if (crd >= 0) {
      const int p = (crd >> 16) & 255;
      char* p0 = b_lds + buf * B_IMG + 2 * ((crd >> 24) & 1) * B_PLANE + (((crd & 255) * IH + ((crd >> 8) & 255)) * IW + 2 * p) * 16;
      *(u32x4*)p0 = u32x4{rq[0].x, rq[1].x, rq[2].x, rq[3].x};
      *(u32x4*)(p0 + B_PLANE) = u32x4{rq[4].x, rq[5].x, rq[6].x, rq[7].x};
      if (p < 8) {   // the row has 17 voxels: the ninth pair contributes one
        *(u32x4*)(p0 + 16) = u32x4{rq[0].y, rq[1].y, rq[2].y, rq[3].y};
        *(u32x4*)(p0 + B_PLANE + 16) = u32x4{rq[4].y, rq[5].y, rq[6].y, rq[7].y};
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
  // a step and the LDS buffers it lives in (pa: weight slab buffer, toggles every step; pb: halo image, toggles when
  // the step brings a new chunk)
  struct Cursor { int tile, s, n, d0, h0, w0, pa, pb; };
  auto advance = [&](Cursor& c) __attribute__((always_inline)) {
    c.pa ^= 1;
    if (c.s + 1 < S) {
      ++c.s;
    } else {
      c.s = 0;
      c.tile += t_step;
      if (c.tile < t_end) decode(c.tile, c.n, c.d0, c.h0, c.w0);
    }
    if (step_newb(c.s)) c.pb ^= 1;
  };

  const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
  const int ostride = Do * Ho * Wo;
  const long osample = (long)a.Cout * ostride;

  Cursor cur{t_begin, 0, 0, 0, 0, 0, 0, 0}, st, ld;
  decode(t_begin, cur.n, cur.d0, cur.h0, cur.w0);
  load_B(cur.n, cur.d0, cur.h0, cur.w0, 0);
  load_A(0);
  store_B(0);
  store_A(0);
  st = cur;
  advance(st);       // S >= 3: the second step always exists
  load_A(st.s);
  if (step_newb(st.s)) load_B(st.n, st.d0, st.h0, st.w0, step_chunk(st.s));
  ld = st;
  advance(ld);
  __syncthreads();

  f32x16 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  const int my_tiles = (t_end - t_begin + t_step - 1) / t_step;
  const int nslots = 2 * my_tiles * S;
  // slot k: half-group 0 computes at even k and is in its load slot at odd k; half-group 1 the other way round
#pragma unroll 1
  for (int k = 0; k < nslots; ++k) {
    DX3_MARK(0);
    if (((k + grp) & 1) == 0) {
      // ---- compute slot: fragment reads, in-register split, 54 MFMAs.  The computing wave outranks its loading partner
      // on the SIMD's vector issue (waves 4-7 otherwise lose the age arbitration and run ~12 % longer slots).
      __builtin_amdgcn_s_setprio(1);
      const int kd = step_kd(cur.s);
      const char* ab = a_lds + cur.pa * A_SLAB + lane * 16;
      const char* bb = b_lds + cur.pb * B_IMG + boff + (kd == 0 ? IH * IW * 16 : 0);   // taps kd = 0 read x[m + 1] along d
      // the four (h, w) neighbours x[m + delta] of this lane's position, raw fp32.  Taps are visited grouped by neighbour --
      // delta 0 (taps 4, 5, 7, 8), 1 (3, 6), 2 (1, 2), 3 (0).  Only the high term of the first neighbour is built before
      // the first MFMA; its other two terms and the neighbours still to come are split beside the MFMAs of the first seven
      // taps, two channel pairs (22 vector instructions) per tap: under 5 per MFMA gap.
      float4 raw[4][2];
#pragma unroll
      for (int dlt = 0; dlt < 4; ++dlt) {
        raw[dlt][0] = *(const float4*)(bb + ((dlt >> 1) * IW + (dlt & 1)) * 16);
        raw[dlt][1] = *(const float4*)(bb + ((dlt >> 1) * IW + (dlt & 1)) * 16 + B_PLANE);
      }
      constexpr int TAPS[9] = {4, 5, 7, 8, 3, 6, 1, 2, 0};
      constexpr int SLOT[9] = {0, 0, 0, 0, 1, 1, 2, 2, 0};          // fragment slot a tap multiplies with
      constexpr int SPLIT_D[7] = {0, 1, 1, 2, 2, 3, 3};             // neighbour split beside tap i < 7 ...
      constexpr int SPLIT_S[7] = {0, 1, 1, 2, 2, 0, 0};             // ... into this slot
      dx_bf16x8 fa[2][3];
      u32x4 fw[3][3];                                               // [slot][term]
#pragma unroll
      for (int term = 0; term < 3; ++term) fa[0][term] = *(const dx_bf16x8*)(ab + (TAPS[0] * 3 + term) * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) dx_split_h(raw[0][0], raw[0][1], j, fw[0][0]);
      // register double buffer for the weight fragments: the three reads of the next tap are issued in front of the six
      // MFMAs of this one (sched_barrier / sched_group_barrier pin the order; left alone hipcc re-orders the taps, issues
      // every LDS read right before its first use and waits for it -- with one computing wave per SIMD nothing would hide
      // that latency)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int tap9 = TAPS[i], cs = i & 1;
        if (i < 8) {
#pragma unroll
          for (int term = 0; term < 3; ++term)
            fa[cs ^ 1][term] = *(const dx_bf16x8*)(ab + (TAPS[i + 1] * 3 + term) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int kh = tap9 / 3, kw = tap9 % 3;
        const int pc = (kh != 1) * 2 + (kw != 1);          // output parity class (h, w) of this pass
        if (i == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dx_split_ml(raw[0][0], raw[0][1], j, fw[0][0], fw[0][1], fw[0][2]);
        } else if (i < 7) {
          const int d = SPLIT_D[i], sl = SPLIT_S[i], j0 = ((i - 1) & 1) * 2;
          dx_split_pair(raw[d][0], raw[d][1], j0, fw[sl][0], fw[sl][1], fw[sl][2]);
          dx_split_pair(raw[d][0], raw[d][1], j0 + 1, fw[sl][0], fw[sl][1], fw[sl][2]);
        }
        // smallest terms first -- except in the slot's first tap, which starts with the products of the halo fragment's
        // high term: that one is ready four instructions after the fragment arrives, the others follow in the MFMA gaps
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PA0[6] = {0, 1, 2, 0, 1, 0}, PB0[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
        for (int q = 0; q < 6; ++q)
          acc[pc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cs][i == 0 ? PA0[q] : PA[q]],
                                                            __builtin_bit_cast(dx_bf16x8, fw[SLOT[i]][i == 0 ? PB0[q] : PB[q]]),
                                                            acc[pc], 0, 0, 0);
        if (i == 0) {   // high-term products first; the chains of the middle and low terms fill the gaps in that order
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          }
        } else if (i < 7) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      DX3_MARK(3);
      if (cur.s == a.NCH - 1 || cur.s == S - 1) {
        // end of a pass: y = act(acc * scale + shift + res_pre) + res_post on output planes 2m + pd; a lane holds the two
        // w-parities of (h-parity, channel) = 8 contiguous bytes, 16 lanes 128 contiguous bytes.  With all 32 output
        // channels present the channel part of a plain store's address is a scalar offset (no vector instruction per store);
        // otherwise it stays in the vector offset, where the hardware range check drops channels >= Cout.
        const int pd = cur.s == S - 1;
        const int md = cur.d0 + dl, mh = cur.h0 + hl, mw = cur.w0 + wl;
        const int ok = (int)(md < a.Di) & (int)(mh < a.Hi) & (int)(mw < a.Wi);
        const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.y + (long)cur.n * osample, osample * 4);
        const bool plain = !has_aff && !has_pre && !has_post && a.slope == 1.f;
        if (plain) {
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            // channel >= Cout is beyond the descriptor's range: dropped by the hardware
            const int base = dca_pred_off((((2 * md + pd) * Ho + 2 * mh + ph) * Wo + 2 * mw + 4 * half * ostride) * 4, ok);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const u32x2 o = {__float_as_uint(acc[ph * 2][r]), __float_as_uint(acc[ph * 2 + 1][r])};
              const int coff = ((r & 3) + 8 * (r >> 2)) * ostride * 4;
              if (cout_full) __builtin_amdgcn_raw_buffer_store_b64(o, yr, base, coff, DX3_NT ? 2 : 0);
              else __builtin_amdgcn_raw_buffer_store_b64(o, yr, base + coff, 0, DX3_NT ? 2 : 0);
            }
          }
          if constexpr (STATS) {   // per-lane running sums (reduced over lanes and waves once, at the end of the kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (stat_first) {   // the wave's first pass: the shift of (half, r) = what lane 0 of the half produced
                const float kf = fs_half_first(acc[0][r], half);
                if ((lane & 31) == 0) fs_slot(stat_w, half, r)[0] = kf;
              }
              const float k = fs_slot(stat_w, half, r)[0];
              const float d0 = ok ? acc[0][r] - k : 0.f, d1 = ok ? acc[1][r] - k : 0.f;
              const float d2 = ok ? acc[2][r] - k : 0.f, d3 = ok ? acc[3][r] - k : 0.f;
              st_s[r] += (d0 + d1) + (d2 + d3);
              st_q[r] += fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3);
            }
            st_n += 4.f * (float)ok;
            stat_first = false;
          }
        } else {
          const __amdgpu_buffer_rsrc_t pr = dca_rsrc((has_pre ? a.res_pre : a.y) + (long)cur.n * osample, osample * 4);
          const __amdgpu_buffer_rsrc_t qr = dca_rsrc((has_post ? a.res_post : a.y) + (long)cur.n * osample, osample * 4);
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
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
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
      }
      DX3_MARK(2);
      advance(cur);
    } else {
      // ---- load slot: this half-group's share of the next step -> LDS, request the step after
#if DX3_STAMP
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
      DX3_MARK(4);
#endif
      if (st.tile < t_end) {
        store_A(st.pa);
        if (step_newb(st.s)) store_B(st.pb);
      }
      DX3_MARK(1);
      if (ld.tile < t_end) {
        load_A(ld.s);
        if (step_newb(ld.s)) load_B(ld.n, ld.d0, ld.h0, ld.w0, step_chunk(ld.s));
      }
      st = ld;
      advance(ld);
      DX3_MARK(3);
    }
    __syncthreads();
    DX3_MARK(5);
  }
  if constexpr (STATS) flush_stats();
}

}  // namespace

namespace {

int dx_grid(long tiles) {
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      ncu = v;
  }
  return (int)(tiles < ncu ? tiles : ncu);
}

int dx_launch(const float* x, const void* wx, float* y, const float* scale, const float* shift, const float* res_pre,
              const float* res_post, float slope, double* stat_part, int N, int Cin, int Cout, int Di, int Hi, int Wi,
              hipStream_t stream) {
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
  a.stat_part = stat_part;
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL);
  const bool stats = stat_part != nullptr;
  auto kern = stats ? deconv3_bf16x3_kernel<true> : deconv3_bf16x3_kernel<false>;
  const int lds = LDS_BYTES + (stats ? STAT_LDS : 0);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(dx_grid(tiles)), dim3(512), lds, stream, a);
  return dca_launch_status();
}

}  // namespace

// wx: dca_conv3d_x3_prep_weight(w, wx, A = Cin, B = Cout, src_ab, flip = 0) -- [chunk][tap][term][lane][8 bf16]
extern "C" int dca_deconv3d_x3_forward(const float* x, const void* wx, float* y, const float* scale, const float* shift,
                                       const float* res_pre, const float* res_post, float slope, int N, int Cin, int Cout,
                                       int Di, int Hi, int Wi, hipStream_t stream) {
  return dx_launch(x, wx, y, scale, shift, res_pre, res_post, slope, nullptr, N, Cin, Cout, Di, Hi, Wi, stream);
}

// nchunk of the statistics dca_deconv3d_x3_forward_stats produces (one partial per workgroup of the launch it will make)
extern "C" long dca_deconv3d_x3_stats_chunks(int N, int Di, int Hi, int Wi) {
  if (N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
  return dx_grid((long)N * cdiv(Di, TD) * cdiv(Hi, TH) * cdiv(Wi, TW));
}

// y = deconv(x, w) (no epilogue) plus the BatchNorm batch statistics of y: part (Cout * nchunk * 4 doubles) = one
// {K, n, sum (y - K), sum (y - K)^2} per (channel, workgroup), for dca_bn_finalize_centered (bn_fused_stats.h)
extern "C" int dca_deconv3d_x3_forward_stats(const float* x, const void* wx, float* y, double* stat_part, int N, int Cin,
                                             int Cout, int Di, int Hi, int Wi, hipStream_t stream) {
  DCA_REQUIRE(stat_part);
  return dx_launch(x, wx, y, nullptr, nullptr, nullptr, nullptr, 1.f, stat_part, N, Cin, Cout, Di, Hi, Wi, stream);
}
