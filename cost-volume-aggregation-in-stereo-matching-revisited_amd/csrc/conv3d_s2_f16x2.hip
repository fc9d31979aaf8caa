// 3x3x3 STRIDE-2 convolution (padding 1) on the f16 matrix pipe with fp32-grade accuracy: the "f16x2" split arithmetic of
// conv3d_f16x2.hip (per-channel power-of-two scales, two f16 terms per operand, three MFMA products per fp32 product, fp32
// accumulation, accumulator scaled back by 2^-f_o) for the one convolution shape that kernel does not serve.
//
// Reference operators served: `cost_agg.conv1` = Conv3d(32, 64, 3, stride 2, padding 1) forward (models/augment/cva.py:16-17)
// and the backward-data of `cost_agg.conv3` = ConvTranspose3d(64, 32, 3, padding 1, output_padding 1, stride 2)
// (cva.py:21-29), which is the same operator over dy with the weight read as [output][contraction].  Until round 3 both ran
// on the fp32 MFMA kernel (conv3d_mfma.hip, 0.94 ms per batch-4 launch, six launches per training step).
//
// y[o] = sum_k w[k] x[2o + k - 1]: a stride-2 convolution reads EIGHT times more input than it writes output, so the LDS
// image of a tile's input is what limits the tile.  Work decomposition: persistent workgroups of 8 waves, one per CU; a tile
// is 2 x 4 x 32 OUTPUT voxels (8 MFMA column tiles of one W row each, one per wave) times 64 output channels (two
// accumulators per wave), its input the 5 x 9 x 65 fine halo.  Input channels go through LDS in chunks of FOUR (an 8-channel
// image of that halo would take 94 KB per buffer): the K = 16 of one MFMA is 4 channels x 4 taps, seven K-steps cover the 27
// taps (the 28th has zero weights), 42 MFMAs per wave and chunk behind ONE barrier.  Both images of a chunk are double
// buffered: the halo tile, pre-split into two f16 term images (2 x 48 KB), and the weight fragments (2 x 28 KB): 152 KB of
// the CU's 160.
//
// Halo image layout [term][w parity][row = (d, h)][34 slots][4 f16 channels]: output w reads fine w' = 2w + kw - 1, so
// along W the 32 lanes of a column tile read every second fine voxel; de-interleaved by the parity of w' their 8-byte
// fragments are contiguous (256 B per wave half: conflict free).  kw = 1 reads the odd plane at slot w, kw = 0 / 2 the even
// plane at slots w + 1 / w + 2 (the even plane is stored one slot up, which also makes the staging stores 16-byte aligned).
// A lane's B fragment = its voxel's 4 channels at two taps = two ds_read_b64; lanes 0-31 hold taps 4s, 4s+1 of K-step s,
// lanes 32-63 taps 4s+2, 4s+3.
//
// Staging (fp32 x, Wi % 4 == 0): a fine halo row is one edge voxel + 16 aligned quads; 720 quad items (4 voxels x 4 channels
// = four 16-byte loads) and 45 edge items per chunk, at most two quad items per thread; scaled by 2^xexps[channel], split and
// written as four ds_write_b128 per quad item.  The staging runs TWO chunks ahead (three cursors, see the kernel): during chunk
// c the registers' data (chunk c + 1) is split and stored behind K-steps 0-2 and the loads of chunk c + 2 are issued behind
// K-steps 3-5 (compile-time schedule, as in conv3d_f16x2.hip).
//
// Measured (tools/s2_time.py, 32 -> 64 at 48x136x240, batch 4): 0.45 ms against 0.87 ms for the fp32 MFMA kernel.  Ablations
// (DCA_S2_ABL): without staging loads 0.34, without split + LDS stores 0.38, with neither 0.24 ms -- the MFMAs alone need 0.14 ms
// at the clock held.  The kernel moves 1 KB of LDS reads per MFMA (a 32 x 32 accumulator tile reuses nothing) and 78 KB of LDS
// stores per chunk of 42 MFMAs per wave (the weights' 28 KB are re-staged for every tile): LDS traffic, not the matrix pipe,
// sets its speed.
#include "dca_common.h"

typedef _Float16 s2_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TD = 2, TH = 4, TW = 32;             // output tile: 256 voxels = 8 column tiles of one W row
constexpr int ID = 2 * TD + 1, IH = 2 * TH + 1;    // 5 x 9 fine halo rows of 65 voxels
constexpr int NROW = ID * IH;                      // 45
constexpr int ROWB = 34 * 8;                       // 272 B per (row, parity): 33 positions x 4 f16 channels + one pad slot
constexpr int B_PLANE = NROW * ROWB;               // 12240
constexpr int B_TERM = 2 * B_PLANE;                // 24480: even plane, odd plane
constexpr int B_BYTES = 2 * B_TERM;                // 48960: two terms
constexpr int NSTEP = 7;                           // K-steps of 4 taps x 4 channels
constexpr int A_CHUNK = NSTEP * 2 * 2 * 1024;      // [step][channel block of 32][term][64 lanes x 16 B] = 28672
constexpr int LDS_BYTES = 2 * B_BYTES + 2 * A_CHUNK;   // 155264
constexpr int MAX_CIN = 256;
constexpr int TAB_BYTES = (MAX_CIN + 64 + 128 + 8 * 64) * 4;   // xexps[MAX_CIN] | f_o[64] | scale[64] shift[64] | per-wave maxima [8][64] (EPI)
constexpr int NQ = NROW * 16;                      // 720 quad items per chunk
constexpr int NE0 = NQ - 512;                      // threads [NE0, NE0 + NROW) carry the 45 edge items (they have one quad item)
constexpr int NA_ITEMS = A_CHUNK / 16;             // 1792
constexpr int KA = (NA_ITEMS + 511) / 512;         // 4
static_assert(NQ <= 1024 && NE0 + NROW <= 512, "two quad items per thread at most, edge items on threads with one");
static_assert(LDS_BYTES + TAB_BYTES <= 160 * 1024, "LDS");

struct S2Args {
  const float* x;
  const unsigned short* wx;
  float* y;
  const float* res_post;
  int N, Cin, Cout, NC4;
  int Di, Hi, Wi, Do, Ho, Wo;
  int nTD, nTH, nTW;
  const int* xexps;          // scale exponent of every input channel (Cin ints)
  const int* ofo;            // behind the packed weight image: f_o of every output channel (blocks of 64)
  const float* scale;        // EPI (inference): y = act(conv * scale[c] + shift[c]) [+ res_post]; null: identity affine
  const float* shift;
  float slope;
  unsigned* y_cmax;          // EPI: optional per-channel slots [c][blockIdx.x] <- max |y| (dca_common.h)
  int abl;                   // ablation switches (DCA_S2_ABL, tools/s2_time.py): 1 no staging loads, 2 no LDS stores, 4 no split
};

__device__ __forceinline__ void s2_split(float v, int e, _Float16& h, _Float16& l) {   // v 2^e = h + l (+ <= 2^-22 relative)
  const float u = ldexpf(v, e);
  h = (_Float16)u;
  l = (_Float16)(u - (float)h);
}
__device__ __forceinline__ void s2_split_abl(float v, int e, _Float16& h, _Float16& l, int abl) {
  if (abl & 4) { h = (_Float16)v; l = h; return; }
  s2_split(v, e, h, l);
}

// byte offset inside a term image of tap t's fragment relative to the lane's base (2 dl, 2 hl, slot w)
__device__ __forceinline__ constexpr int s2_toff(int t) {
  const int tt = t < 27 ? t : 26;          // the 28th tap has zero weights: any finite data
  const int kd = tt / 9, kh = (tt / 3) % 3, kw = tt % 3;
  return (kw == 1 ? B_PLANE : 0) + (kd * IH + kh) * ROWB + (kw == 1 ? 0 : 1 + kw / 2) * 8;
}

// EPI: the inference epilogue (folded BatchNorm + activation, per-channel output maxima for the f16x2 convolution that reads
// y) -- compiled out of the training launches, whose output goes to a separate BatchNorm pass
template <bool EPI>
__global__ __launch_bounds__(512) void conv3s2_f16x2_kernel(S2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;
  char* a_lds = smem + 2 * B_BYTES;
  int* xe_lds = (int*)(smem + LDS_BYTES);
  int* fo_lds = xe_lds + MAX_CIN;
  float* aff_lds = (float*)(fo_lds + 64);         // scale[64] | shift[64]
  float* ycm_lds = aff_lds + 128;                 // [wave][64 channels]

  // tid >> 6 stays a per-lane value on purpose (no readfirstlane): see conv3d_wgrad_f16x2.hip
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int cblk = blockIdx.y;
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  if (t_begin >= t_end) {
    if constexpr (EPI) {   // a workgroup without tiles still owns its slot of the per-channel maxima
      if (a.y_cmax && tid < 64 && cblk * 64 + tid < a.Cout) a.y_cmax[(long)(cblk * 64 + tid) * DCA_AMAX_CSLOTS + blockIdx.x] = 0u;
    }
    return;
  }
  for (int i = tid; i < a.NC4 * 4; i += 512) xe_lds[i] = i < a.Cin ? dca_coherent_loadi(a.xexps + i) : 0;
  if (tid < 64) fo_lds[tid] = dca_coherent_loadi(a.ofo + cblk * 64 + tid);
  if constexpr (EPI) {
    if (tid < 64) {
      const int co = min(cblk * 64 + tid, a.Cout - 1);
      aff_lds[tid] = a.scale ? a.scale[co] : 1.f;
      aff_lds[64 + tid] = a.scale ? a.shift[co] : 0.f;
    }
    ycm_lds[tid] = 0.f;      // 8 x 64 = 512 entries
  }

  const int dl = wv >> 2, hl = wv & 3;
  const int lanebase = ((2 * dl) * IH + 2 * hl) * ROWB + l31 * 8;
  const int cstride = a.Di * a.Hi * a.Wi, ostride = a.Do * a.Ho * a.Wo;
  const long sample = (long)a.Cin * cstride, osample = (long)a.Cout * ostride;
  const int NC = a.NC4;
  const long wbytes = (long)NC * A_CHUNK;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc((const char*)a.wx + (long)cblk * wbytes, wbytes);

  // ---- staging items of this thread (fixed for the whole kernel)
  int qrow[2], qq[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int it = tid + 512 * k;
    qrow[k] = it < NQ ? it >> 4 : -1;
    qq[k] = it & 15;
  }
  const int erow = (tid >= NE0 && tid < NE0 + NROW) ? tid - NE0 : -1;
  float4 rq[2][4];
  float re[4];
  float4 ra[KA];
  auto load_quad = [&](int k, __amdgpu_buffer_rsrc_t xr, int d0, int h0, int w0, int chunk, int on) __attribute__((always_inline)) {
    const int row = qrow[k], id = row / IH, ih = row - id * IH;
    const int di = 2 * d0 - 1 + id, hi = 2 * h0 - 1 + ih, wi = 2 * w0 + 4 * qq[k];      // fine w of the quad's first voxel (iw = 1 + 4q)
    const int ok = (int)(row >= 0) & (int)((unsigned)di < (unsigned)a.Di) & (int)((unsigned)hi < (unsigned)a.Hi) &
                   (int)(wi < a.Wi) & on;                                                  // Wi % 4 == 0: a quad is in or out
    const int off = (chunk * 4 * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) rq[k][j] = dca_bload4(xr, off + j * cstride * 4, ok & (int)(chunk * 4 + j < a.Cin));
  };
  auto load_edge = [&](__amdgpu_buffer_rsrc_t xr, int d0, int h0, int w0, int chunk, int on) __attribute__((always_inline)) {
    const int id = erow / IH, ih = erow - id * IH;
    const int di = 2 * d0 - 1 + id, hi = 2 * h0 - 1 + ih, wi = 2 * w0 - 1;
    const int ok = (int)(erow >= 0) & (int)((unsigned)di < (unsigned)a.Di) & (int)((unsigned)hi < (unsigned)a.Hi) &
                   (int)(wi >= 0) & on;
    const int off = (chunk * 4 * cstride + (di * a.Hi + hi) * a.Wi + wi) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) re[j] = dca_bload1(xr, off + j * cstride * 4, ok & (int)(chunk * 4 + j < a.Cin));
  };
  auto load_A = [&](int chunk, int on) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      ra[k] = dca_bload4(wr, chunk * A_CHUNK + it * 16, (int)(it < NA_ITEMS) & on);
    }
  };
  auto store_A = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int it = tid + 512 * k;
      if (it < NA_ITEMS) *(float4*)(a_lds + buf * A_CHUNK + it * 16) = ra[k];
    }
  };
  // quad item k: voxels v = 0..3 at iw = 1 + 4q + v; v = 0, 2 are odd fine positions (plane 1, slots 2q, 2q + 1), v = 1, 3 even
  // ones (plane 0, positions 2q + 1, 2q + 2 = slots 2q + 2, 2q + 3): one 16-byte store per (term, plane)
  auto store_quad = [&](int k, int chunk, int buf) __attribute__((always_inline)) {
    if (qrow[k] < 0) return;
    const int4 e4 = *(const int4*)(xe_lds + chunk * 4);
    const int e[4] = {e4.x, e4.y, e4.z, e4.w};
    char* img = b_lds + buf * B_BYTES + qrow[k] * ROWB;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int va = par ? 0 : 1, vb = va + 2;
      s2_f16x8 hv, lv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* c = (const float*)&rq[k][j];
        _Float16 h, l;
        s2_split_abl(c[va], e[j], h, l, a.abl); hv[j] = h; lv[j] = l;
        s2_split_abl(c[vb], e[j], h, l, a.abl); hv[4 + j] = h; lv[4 + j] = l;
      }
      char* dst = img + (par ? B_PLANE + (2 * qq[k]) * 8 : (2 * qq[k] + 2) * 8);
      *(s2_f16x8*)dst = hv;
      *(s2_f16x8*)(dst + B_TERM) = lv;
    }
  };
  auto store_edge = [&](int chunk, int buf) __attribute__((always_inline)) {      // iw = 0: even plane, position 0 = slot 1
    if (erow < 0) return;
    const int4 e4 = *(const int4*)(xe_lds + chunk * 4);
    const int e[4] = {e4.x, e4.y, e4.z, e4.w};
    s2_f16x4 hv, lv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      _Float16 h, l;
      s2_split(re[j], e[j], h, l); hv[j] = h; lv[j] = l;
    }
    char* dst = b_lds + buf * B_BYTES + erow * ROWB + 8;
    *(s2_f16x4*)dst = hv;
    *(s2_f16x4*)(dst + B_TERM) = lv;
  };
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    const int tw = tile % a.nTW; tile /= a.nTW;
    const int th = tile % a.nTH; tile /= a.nTH;
    const int td = tile % a.nTD;
    n = tile / a.nTD;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
  };

  // Three cursors walk the workgroup's chunk sequence (tile by tile, NC chunks each): cur = the chunk being computed, cur + 1 =
  // the chunk whose data sits in the staging registers and is split + stored into the other buffer pair during cur, cur + 2 =
  // the chunk whose loads are issued during cur, once the registers are free.  A global load thus has a whole chunk (~3 k
  // cycles) to arrive; requested and consumed inside ONE chunk of 42 MFMAs per wave (the schedule of conv3d_f16x2.hip, whose
  // chunks are 84 MFMAs long) the kernel waited for memory in every chunk: 67 k cycles per tile against 21.5 k of MFMAs.
  struct Cursor { int tile, chunk, n, d0, h0, w0; };
  auto advance = [&](Cursor& c) __attribute__((always_inline)) {
    if (c.chunk + 1 < NC) {
      ++c.chunk;
    } else {
      c.chunk = 0;
      c.tile += t_step;
      if (c.tile < t_end) decode(c.tile, c.n, c.d0, c.h0, c.w0);
    }
  };
  auto load_all = [&](const Cursor& c, int part) __attribute__((always_inline)) {     // part 0: quad 0; 1: quad 1 + edge; 2: weights
    const int on = (c.tile < t_end && !(a.abl & 1)) ? 1 : 0;
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)c.n * sample, sample * 4);
    if (part == 0) load_quad(0, xr, c.d0, c.h0, c.w0, c.chunk, on);
    if (part == 1) { load_quad(1, xr, c.d0, c.h0, c.w0, c.chunk, on); load_edge(xr, c.d0, c.h0, c.w0, c.chunk, on); }
    if (part == 2) load_A(c.chunk, on);
  };
  Cursor cur{t_begin, 0, 0, 0, 0, 0}, st, ld;
  decode(t_begin, cur.n, cur.d0, cur.h0, cur.w0);
  load_all(cur, 0); load_all(cur, 1); load_all(cur, 2);
  __syncthreads();      // the exponent tables
  store_quad(0, 0, 0);
  store_quad(1, 0, 0);
  store_edge(0, 0);
  store_A(0);
  st = cur;
  advance(st);          // NC >= 2: the second chunk always exists
  load_all(st, 0); load_all(st, 1); load_all(st, 2);
  ld = st;
  advance(ld);
  __syncthreads();

  const bool has_post = a.res_post != nullptr;
  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  int buf = 0;
#pragma unroll 1
  for (; cur.tile < t_end; buf ^= 1) {
    const char* ab = a_lds + buf * A_CHUNK + lane * 16;
    const char* bb = b_lds + buf * B_BYTES + lanebase;
    const bool st_on = st.tile < t_end && !(a.abl & 2);
    s2_f16x8 fa[2][2][2], fb[2][2];      // [slot][channel block][term], [slot][term]
    auto load_frag = [&](int s, int slot) __attribute__((always_inline)) {
      const int oa = half ? s2_toff(4 * s + 2) : s2_toff(4 * s), ob = half ? s2_toff(4 * s + 3) : s2_toff(4 * s + 1);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int term = 0; term < 2; ++term) fa[slot][cb][term] = *(const s2_f16x8*)(ab + ((s * 2 + cb) * 2 + term) * 1024);
#pragma unroll
      for (int term = 0; term < 2; ++term) {
        const s2_f16x4 lo = *(const s2_f16x4*)(bb + term * B_TERM + oa), hi = *(const s2_f16x4*)(bb + term * B_TERM + ob);
        fb[slot][term] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    load_frag(0, 0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int cs = s & 1;
      if (s + 1 < NSTEP) load_frag(s + 1, cs ^ 1);
      constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};      // smallest terms first
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cs][cb][PA[q]], fb[cs][PB[q]], acc[cb], 0, 0, 0);
      // this K-step's share of the staging (compile-time schedule): the registers' chunk (cur + 1) goes to LDS, then the
      // registers are refilled with chunk cur + 2
      if (st_on) {
        if (s == 0) store_quad(0, st.chunk, buf ^ 1);
        if (s == 1) store_quad(1, st.chunk, buf ^ 1);
        if (s == 2) { store_edge(st.chunk, buf ^ 1); store_A(buf ^ 1); }
      }
      if (s == 3) load_all(ld, 0);
      if (s == 4) load_all(ld, 1);
      if (s == 5) load_all(ld, 2);
      if (s + 1 < NSTEP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // two LDS reads of the next K-step's fragments
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (cur.chunk + 1 == NC) {
      // epilogue: y = acc 2^-f_o [+ res_post]; a register = one output channel at the 32 consecutive w of the wave's row
      const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.y + (long)cur.n * osample, osample * 4);
      const __amdgpu_buffer_rsrc_t qr = dca_rsrc((has_post ? a.res_post : a.y) + (long)cur.n * osample, osample * 4);
      const int d = cur.d0 + dl, h = cur.h0 + hl, w = cur.w0 + l31;
      const int ok = (int)(d < a.Do) & (int)(h < a.Ho) & (int)(w < a.Wo);
      const int vbase = ((d * a.Ho + h) * a.Wo + w) * 4;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, co = cblk * 64 + cl;
          const int okc = ok & (int)(co < a.Cout);
          float v = ldexpf(acc[cb][r], -fo_lds[cl]);
          if constexpr (EPI) v = act_apply(v * aff_lds[cl] + aff_lds[64 + cl], a.slope);
          if (has_post) v += dca_bload1(qr, vbase + co * ostride * 4, okc);
          dca_bstore1(yr, v, vbase + co * ostride * 4, okc);
          if constexpr (EPI) {
            if (a.y_cmax) {      // wave-half maximum of this channel -> the wave's own LDS row (one writer lane, no atomics)
              float m = okc ? fabsf(v) : 0.f;
#pragma unroll
              for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
              if (l31 == 0) ycm_lds[wv * 64 + cl] = fmaxf(ycm_lds[wv * 64 + cl], m);
            }
          }
          acc[cb][r] = 0.f;
        }
      }
    }
    __syncthreads();     // chunk cur + 1's images are complete, chunk cur's are free
    cur = st;
    st = ld;
    advance(ld);
  }
  if constexpr (EPI) {
    if (a.y_cmax) {      // (the loop's last barrier has passed: every wave's row is complete)
      if (tid < 64 && cblk * 64 + tid < a.Cout) {
        float m = ycm_lds[tid];
        for (int w = 1; w < 8; ++w) m = fmaxf(m, ycm_lds[w * 64 + tid]);
        a.y_cmax[(long)(cblk * 64 + tid) * DCA_AMAX_CSLOTS + blockIdx.x] = __float_as_uint(m);
      }
    }
  }
}

// Weight packing, once per launch (the image folds the operand's per-channel exponents in, as x2_prep_weight_kernel of
// conv3d_f16x2.hip): wx[block of 64 output channels][chunk of 4 input channels][K-step s][block of 32][term][lane][j] (f16):
// lane (r = lane & 31, h = lane >> 5) holds row = output channel, k = h * 8 + j: j < 4: tap 4s + 2h, channel chunk*4 + j;
// j >= 4: tap 4s + 2h + 1, channel chunk*4 + j - 4; value w 2^(f_o - xexps[channel]) split into two f16 terms; zero padded
// (channels beyond Cin, rows beyond Cout, the 28th tap).  f_o goes behind the images (ofo[block * 64 + row]).
constexpr int PREP_ROWS = 4;
__global__ __launch_bounds__(512) void s2x2_prep_weight_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                                int A, int Bn, int NC4, int src_ab, int flip,
                                                                const unsigned* __restrict__ slots, int nslots,
                                                                int* __restrict__ xexps, int xexps_given, int* __restrict__ ofo) {
  __shared__ int xe[MAX_CIN];
  __shared__ int rowmax[PREP_ROWS][2];
  const int tid = threadIdx.x, cblk = blockIdx.x / (64 / PREP_ROWS), r0 = (blockIdx.x % (64 / PREP_ROWS)) * PREP_ROWS;
  if (slots && !xexps_given) {
    for (int c0 = 0; c0 < A; c0 += 32) {
      const int c = c0 + (tid >> 4), l = tid & 15;
      unsigned v = 0;
      if (c < A)
        for (int i = l; i < nslots; i += 16) { const unsigned u = slots[(long)c * DCA_AMAX_CSLOTS + i]; v = v > u ? v : u; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { const unsigned u = (unsigned)__shfl_xor((int)v, o, 64); v = v > u ? v : u; }
      if (c < A && l == 0) {
        const int e = x2_scale_exp(v);
        xe[c] = e;
        if (blockIdx.x == 0) xexps[c] = e;
      }
    }
  } else {
    for (int c = tid; c < A; c += 512) xe[c] = dca_coherent_loadi(xexps + c);
  }
  __syncthreads();
  {
    const int r = tid >> 7, l = tid & 127, bi = cblk * 64 + r0 + r;
    int m = -100000;
    if (bi < Bn) {
      for (int i = l; i < A * 27; i += 128) {
        const int ai = i / 27, tap = i - ai * 27;
        const float v = src_ab ? src[((long)ai * Bn + bi) * 27 + tap] : src[((long)bi * A + ai) * 27 + tap];
        const int be = (int)((__float_as_uint(v) >> 23) & 255);
        const int e = be == 0 ? -100000 : be - 127 - xe[ai];
        m = m > e ? m : e;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(m, o, 64); m = m > u ? m : u; }
    if ((l & 63) == 0) rowmax[r][l >> 6] = m;
  }
  __syncthreads();
  if (tid < PREP_ROWS) {
    const int m = rowmax[tid][0] > rowmax[tid][1] ? rowmax[tid][0] : rowmax[tid][1];
    const int fo = m <= -100000 ? 0 : 14 - m;
    rowmax[tid][0] = fo;
    ofo[cblk * 64 + r0 + tid] = fo;
  }
  __syncthreads();
  const int nitems = NC4 * NSTEP * 2 * PREP_ROWS;
  unsigned short* out = dst + (long)cblk * NC4 * (A_CHUNK / 2);
  for (int it = tid; it < nitems; it += 512) {
    const int r = it % PREP_ROWS, hf = (it / PREP_ROWS) & 1, t = it / (2 * PREP_ROWS), s = t % NSTEP, chunk = t / NSTEP;
    const int rr = r0 + r, cb = rr >> 5, rl = rr & 31, bi = cblk * 64 + rr, fo = rowmax[r][0];
    s2_f16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 4 * s + 2 * hf + (j >> 2), ai = chunk * 4 + (j & 3);
      float v = 0.f;
      int e = 0;
      if (ai < A && bi < Bn && tap < 27) {
        const int st = flip ? 26 - tap : tap;
        v = src_ab ? src[((long)ai * Bn + bi) * 27 + st] : src[((long)bi * A + ai) * 27 + st];
        e = fo - xe[ai];
      }
      _Float16 h, l;
      s2_split(v, e, h, l);
      hv[j] = h; lv[j] = l;
    }
    const long o = (((long)(chunk * NSTEP + s) * 2 + cb) * 2) * 512 + (hf * 32 + rl) * 8;      // f16 elements; term stride 512
    *(s2_f16x8*)(out + o) = hv;
    *(s2_f16x8*)(out + o + 512) = lv;
  }
}

}  // namespace

// bytes of the packed image: fragments, then f_o (one int per output channel, padded to blocks of 64)
extern "C" long dca_conv3d_s2x2_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (long)((Cout + 63) / 64) * ((Cin + 3) / 4) * A_CHUNK + (long)((Cout + 63) / 64) * 64 * 4;
}

// Packs w (A input x B output channels; src_ab ? src[a][b][27] : src[b][a][27]; flip reverses the taps) for ONE launch of
// dca_conv3d_s2x2_forward over an operand with per-channel exponents xexps: given (x_slots == null) or derived from the
// operand's per-channel maxima and written (x_slots != null), exactly as dca_conv3d_x2_prep_weight.
extern "C" int dca_conv3d_s2x2_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip,
                                           const unsigned* x_slots, int nslots, int* xexps, hipStream_t stream) {
  DCA_REQUIRE(w && wx && xexps && A > 0 && A <= MAX_CIN && B > 0 && ((((uintptr_t)wx) & 15) == 0));
  DCA_REQUIRE(x_slots == nullptr || (nslots > 0 && nslots <= DCA_AMAX_CSLOTS));
  const int NC4 = (A + 3) / 4, cblks = (B + 63) / 64;
  int* ofo = (int*)((char*)wx + (long)cblks * NC4 * A_CHUNK);
  hipLaunchKernelGGL(s2x2_prep_weight_kernel, dim3(cblks * (64 / PREP_ROWS)), dim3(512), 0, stream, w, (unsigned short*)wx, A,
                     B, NC4, src_ab, flip, x_slots, nslots, xexps, x_slots == nullptr ? 1 : 0, ofo);
  return dca_launch_status();
}

// slots per channel that dca_conv3d_s2x2_forward fills in y_cmax (= its workgroups per block of 64 output channels)
extern "C" long dca_conv3d_s2x2_out_slots(int N, int Cout, int Di, int Hi, int Wi) {
  if (N <= 0 || Cout <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
  const long tiles = (long)N * cdiv((Di + 1) / 2, TD) * cdiv((Hi + 1) / 2, TH) * cdiv((Wi + 1) / 2, TW);
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        v > 0)
      ncu = v;
  }
  const int cblks = (Cout + 63) / 64;
  long gx = ncu / cblks > 0 ? ncu / cblks : 1;
  return gx < tiles ? gx : tiles;
}

// y = act(conv3d(x, w, stride 2, padding 1) * scale[c] + shift[c]) + res_post (scale / shift / res_post may be null, slope 1 =
// no activation), fp32 tensors: x (N, Cin, Di, Hi, Wi) with Wi % 4 == 0 and 16-byte aligned, y (N, Cout, (Di+1)/2, (Hi+1)/2,
// (Wi+1)/2); xexps / wx from dca_conv3d_s2x2_prep_weight for THIS operand; y_cmax (may be null): per-channel slots [c][s],
// s < dca_conv3d_s2x2_out_slots(...), that receive max |y|.
extern "C" int dca_conv3d_s2x2_forward(const float* x, const int* xexps, const void* wx, float* y, const float* scale,
                                       const float* shift, float slope, const float* res_post, unsigned* y_cmax, int N,
                                       int Cin, int Cout, int Di, int Hi, int Wi, hipStream_t stream) {
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE(x && xexps && wx && y && N > 0 && Cin > 4 && Cin <= MAX_CIN && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0);   // >= 2 chunks: the staging runs two chunks ahead
  DCA_REQUIRE(Wi % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)wx)) & 15) == 0);
  S2Args a;
  a.x = x; a.wx = (const unsigned short*)wx; a.y = y; a.res_post = res_post;
  a.scale = scale; a.shift = shift; a.slope = slope; a.y_cmax = y_cmax;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NC4 = (Cin + 3) / 4;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi;
  a.Do = (Di + 1) / 2; a.Ho = (Hi + 1) / 2; a.Wo = (Wi + 1) / 2;
  DCA_REQUIRE((long)(Cin + 3) * Di * Hi * Wi * 4 < 0x7ffffff0L && (long)(Cout + 63) * a.Do * a.Ho * a.Wo * 4 < 0x7ffffff0L);
  a.nTD = cdiv(a.Do, TD); a.nTH = cdiv(a.Ho, TH); a.nTW = cdiv(a.Wo, TW);
  a.xexps = xexps;
  {
    static const int abl = [] { const char* e = getenv("DCA_S2_ABL"); return e ? atoi(e) : 0; }();
    a.abl = abl;
  }
  const int cblks = (Cout + 63) / 64;
  a.ofo = (const int*)((const char*)wx + (long)cblks * a.NC4 * A_CHUNK);
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL && cblks <= 65535);
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        v > 0)
      ncu = v;
  }
  int gx = ncu / cblks > 0 ? ncu / cblks : 1;
  if (gx > tiles) gx = (int)tiles;
  const int lds = LDS_BYTES + TAB_BYTES;
  const bool epi = scale != nullptr || slope != 1.f || y_cmax != nullptr;
  DCA_REQUIRE(y_cmax == nullptr || gx <= DCA_AMAX_CSLOTS);
  auto kern = epi ? conv3s2_f16x2_kernel<true> : conv3s2_f16x2_kernel<false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(gx, cblks), dim3(512), lds, stream, a);
  return dca_launch_status();
}
