// 3x3x3 stride-1 convolution on the f16 matrix pipe with fp32-grade accuracy ("f16x2" split emulation).
//
// Every fp32 operand is first scaled by a power of two (exact) and then split into two f16 terms, x 2^e = h + l with
// h = f16(x 2^e), l = f16(x 2^e - h): 11 + 11 significand bits plus the sign of l, |x 2^e - h - l| <= 2^-22 |x 2^e| (an fp32
// operand itself carries a 2^-24 rounding).  The scales are PER CHANNEL (round 3): input channel k of x is scaled by
// 2^xexps[k], taken from that channel's max |.| (scaled maximum in [2^14, 2^15): inside the f16 range whatever the magnitude
// -- loss gradients of 1e-20 included -- and whatever the OTHER channels' magnitudes), and the weight row of output channel
// o is packed as w[o][k] 2^(f_o - xexps[k]) with f_o chosen so that the row's largest scaled entry lies in [2^14, 2^15)
// (dca_conv3d_x2_prep_weight, per launch: it needs the operand's exponents).  The accumulator of output channel o is
// scaled back by 2^-f_o (v_ldexp_f32).  So every output channel is computed to 2^-22 of ITS largest term, like an fp32
// convolution -- the per-tensor scale of round 2 lost the low term of channels 2^-18 below the tensor's maximum
// (tests/test_gpu_parity.py::test_f16x2_per_channel_scales).  A product w*x is evaluated as the
// three partial products of weight >= 2^-11,
//     w_h x_h + (w_h x_l + w_l x_h)
// each exact in the MFMA's fp32 accumulator; the dropped w_l x_l is <= 2^-22 relative.  Three v_mfma_f32_32x32x16_f16
// replace the six bf16 products of conv3d_bf16x3.hip (and sixteen v_mfma_f32_32x32x2f32 of conv3d_mfma.hip) per 32 input
// channels, tap and 32-voxel tile: half the matrix-pipe cycles of the bf16x3 kernel, 5.3x fewer than fp32 MFMA.  Against
// fp64 convolutions the kernel measures the same error as the fp32 MFMA kernel (the fp32 accumulation of 27*Cin products
// dominates both), tests/test_gpu_parity.py.
//
// The operand arrives either as fp32 (scaled and split while it is staged into LDS) or in the packed px2 format
// (dca_common.h: already scaled, split and laid out [voxel][8 channels] by the BatchNorm kernel that wrote it -- staging is a
// copy of 16-byte words, a halo row one contiguous run).  Per-channel maxima travel as slots (dca_common.h) filled by the
// producers (bn_apply / bn_bwd_apply, pointwise.hip; this kernel's own epilogue) or by dca_cmax_f32.  Nothing is read back
// by the host; no atomics.
//
// Reference operators served: nn.Conv3d(k=3, s=1, p=1) of convbn_3d (models/submodule.py:121-124) in dres0/dres1,
// Multi_Aggregation and the cva blocks (models/augment/cva.py:13-55), and their backward-data.
//
// Work decomposition: one persistent workgroup (8 waves) per CU; a tile is 4 x 8 x 16 output voxels (512 = 16 MFMA column
// tiles, two per wave) times 32 output channels.  Input channels go through LDS in chunks of EIGHT: the K = 16 of one MFMA is
// 8 channels x 2 taps, so a chunk is 14 tap pairs (the 28th tap has zero weights) = 84 MFMAs per wave behind ONE barrier
// (round 2: 16-channel chunks in three phases of 54 MFMAs with a barrier each and two more around the halo store).  Per
// chunk LDS holds the 6 x 10 x 18 halo tile, pre-split into two f16 term images [term][voxel][8 f16] (34.5 KB: a lane's B
// fragment -- its voxel's 8 channels at the pair's tap of its wave half -- is one ds_read_b128), and the chunk's weight
// fragments (28 KB, pre-scaled, pre-split and pre-swizzled by x2_prep_weight_kernel); both are double buffered (126 KB), so the
// loads, the [split and] LDS stores of the next chunk and the MFMAs of this one need no ordering among themselves.  The
// staging steps sit at compile-time positions between the tap pairs (one global load or one LDS store per pair): issued as a
// burst they held the matrix pipe for 1-2.4 k cycles per chunk (s_memtime stamps, tools/x2_stamps.py).
#include "dca_common.h"
#include "bn_fused_stats.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Non-temporal output stores (X2_NT=1; measured on the bf16x3 kernel, whose epilogue this is): y is written once, so it need not displace the halo lines the neighbouring tiles
// re-read from this XCD's L2 -- 591 -> 568 us on the fused-epilogue 32->32 launch at 48x136x240 in isolation
// (tools/nt_ablate.sh).  In the network the consumer of y (BatchNorm statistics + apply in training, the next convolution
// in inference) runs right behind and finds a 200 MB output partly in the 256 MB infinity cache when it was stored with
// the default policy: training step 37.86 -> 37.38 ms per pair at batch 1, eval forward 7.63 -> 7.51 ms, batch-4 step
// 143.7 -> 143.4 ms (tools/nt_bench_ab.sh).  So the default is plain stores.
#ifndef X2_NT
#define X2_NT 0
#endif
// X2_STAMP (debug build, tools/x2_stamps.py): dca_x2_debug_set_stamps(ptr) names a buffer of 2 x 96 x 8 unsigned long long that
// receives s_memtime stamps of the first 96 chunks (8 marks per chunk) of workgroup 0, waves 0 and 7
#ifndef X2_STAMP
#define X2_STAMP 0
#endif
// X2_LD_FRONT (A/B builds): 1 = the staging loads of a phase in one burst in front of the slab stores, 2 = two per tap
// behind the first four taps, 0 (shipped) = one per tap
#ifndef X2_LD_FRONT
#define X2_LD_FRONT 0
#endif
#if X2_STAMP
#define X2_MARK(i) do { if (stamp_on && stamp_k < 96) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[stamp_k * 8 + (i)] = t_; } } while (0)
#else
#define X2_MARK(i) do { } while (0)
#endif

namespace {

constexpr int TD = 4, TH = 8, TW = 16;
constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
constexpr int NVOX = ID * IH * IW;                 // 1080 halo voxels
constexpr int NT = 2;                              // terms per operand
constexpr int NPAIR = 14;                          // tap pairs (2p, 2p+1): the K = 16 of one MFMA is 8 channels x 2 taps
constexpr int B_TERM = NVOX * 16;                  // bytes of one f16 term image of an 8-channel chunk (voxels x 16 B)
constexpr int B_BYTES = NT * B_TERM;               // 34560
constexpr int A_CHUNK = NPAIR * NT * 1024;         // weight fragments of a chunk: 14 pairs x 2 terms x (64 lanes x 16 B) = 28672
constexpr int LDS_BYTES = 2 * B_BYTES + 2 * A_CHUNK;   // both images double buffered: 126464 of the CU's 163840
constexpr int MAX_CIN = 256;                       // the per-channel exponent table below
constexpr int TAB_BYTES = (MAX_CIN + 32 + 8 * 32) * 4;   // xexps[MAX_CIN] | f_o[32] | per-wave channel maxima [8][32]
constexpr int NP_ITEMS = NT * NVOX;                // packed-input staging: 2160 16-byte words per chunk = the LDS image itself
constexpr int KP = (NP_ITEMS + 511) / 512;         // 5 per thread
constexpr int KB = (NVOX + 511) / 512;             // 3 single-voxel staging items of 8 channels per thread (unaligned path)
constexpr int NROWS = ID * IH;                     // 60 (d, h) halo rows: 4 aligned quads + 2 edge voxels each
constexpr int NQUAD = NROWS * 4, NEDGE = NROWS * 2;  // aligned path: 240 quad items (8 x b128), 120 edge items (8 x b32)
static_assert(NQUAD <= 512 && NEDGE % 8 == 0 && NEDGE / 8 <= 64, "one quad / edge item per thread");
constexpr int NA_ITEMS = A_CHUNK / 16;             // 1792 b128 per chunk
constexpr int KA = (NA_ITEMS + 511) / 512;         // 4

struct X2Args {
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
  int order;                 // tile traversal: 0 = w fastest (w, h, d, n), 1 = d fastest (d, w, h, n)
  double* stat_part;         // STATS: one partial {K, n, s, q} per (channel, workgroup): bn_fused_stats.h
  const int* xexps;          // fp32 x: the scale exponent of every input channel (Cin ints); unused for packed x
  const int* ofo;            // behind the packed weight image: f_o of every output channel (dca_conv3d_x2_prep_weight)
  unsigned* y_cmax;          // optional (EPI 1): per-channel slots [c][blockIdx.x] that receive max |y| (dca_common.h)
#if X2_STAMP
  unsigned long long* stamps;   // debug build: s_memtime stamps of workgroup 0, waves 0 and 7 (2 x 96 x 8 words)
#endif
};

constexpr int STAT_LDS = 8 * FS_WAVE_FLOATS * 4;

// x 2^e = h + l (+ <= 2^-22 relative)
__device__ __forceinline__ void split2(float v, int e, _Float16& h, _Float16& l) {
  const float u = ldexpf(v, e);   // exact (v_ldexp_f32); scaled maximum < 2^15
  h = (_Float16)u;
  l = (_Float16)(u - (float)h);   // the residual is exact in fp32
}


// STATS: the raw convolution output feeds a training-mode BatchNorm -- the kernel also produces, per channel and workgroup,
// the partial statistics {K, n, sum (y - K), sum (y - K)^2} that dca_bn_finalize_centered consumes (bn_fused_stats.h), so the
// 200 MB statistics pass over y disappears.  Per tile: 3 vector instructions per output value, a DPP reduction over the 32
// positions of a wave half, one ds_add_f32 per (half, channel) into a wave-private LDS slot.
// EPI = 0: no epilogue (the training-mode launches: forward with BatchNorm statistics, backward-data) -- the epilogue's
// arrays, descriptors and branches are compiled out; 1: y = act(v scale + shift + res_pre) + res_post (inference); 2: y = v +
// res_post only (backward-data that also sums the other consumers' gradient: ops._Conv3d alias, _ConvPair)
// PIN: x is in the packed px2 format (dca_common.h) -- staging copies 16-byte words
template <bool VEC, bool STATS, int EPI, bool PIN>
__global__ __launch_bounds__(512) void conv3_f16x2_kernel(X2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_lds = smem;                               // two halo images (one 8-channel chunk each)
  char* a_lds = smem + 2 * B_BYTES;                 // two weight-fragment images
  int* xe_lds = (int*)(smem + LDS_BYTES);           // scale exponents of the input channels (fp32 x)
  int* fo_lds = xe_lds + MAX_CIN;                   // f_o of this block's 32 output channels
  float* ycm_lds = (float*)(fo_lds + 32);           // [wave][channel] maxima (y_cmax)
  float* stat_lds = (float*)(smem + LDS_BYTES + TAB_BYTES);     // STATS only (the launch adds STAT_LDS bytes)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int cblk = blockIdx.y;
  // persistent: the tile space (w fastest) is cut into one contiguous range per XCD (workgroups are dealt round-robin
  // over the 8 XCDs, each with its own L2); inside its XCD's range, workgroup j of cnt takes tiles j, j + cnt, ... so
  // at any time the CUs of an XCD work on neighbouring tiles and share their halos in that L2.  The load / MFMA
  // pipeline runs straight across tile boundaries.
  const long T = (long)a.N * a.nTD * a.nTH * a.nTW;
  const int nx = gridDim.x >= 8 ? 8 : 1, xcd = blockIdx.x % nx;
  const int cnt = (gridDim.x - xcd + nx - 1) / nx;          // workgroups on this XCD
  const int t_begin = (int)(T * xcd / nx) + blockIdx.x / nx, t_end = (int)(T * (xcd + 1) / nx), t_step = cnt;
  float* stat_w = stat_lds + wv * FS_WAVE_FLOATS;   // this wave's slots
  bool stat_first = true;
  // STATS: per-lane running sums of (y - K) and (y - K)^2 over all tiles of the workgroup (the epilogue-free variants have
  // the 32 registers to spare), reduced over the wave halves once, after the tile loop
  float st_s[STATS ? 16 : 1], st_q[STATS ? 16 : 1], st_k[STATS ? 16 : 1], st_n = 0.f;   // st_k: the shifts K, in registers
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st_s[r] = st_q[r] = st_k[r] = 0.f;
  }
  float ycm[EPI == 1 ? 16 : 1];          // max |y| this lane has written, per accumulator register = output channel (y_cmax)
#pragma unroll
  for (int r = 0; r < (EPI == 1 ? 16 : 1); ++r) ycm[r] = 0.f;
  if constexpr (STATS) {
    for (int i = tid; i < 8 * FS_WAVE_FLOATS; i += 512) stat_lds[i] = 0.f;
  }
  // the exponent tables were written by the kernels right in front of this one (dca_conv3d_x2_prep_weight / the producer
  // of x): device-coherent vector loads, never s_load (dca_common.h)
  if constexpr (!PIN) {
    for (int i = tid; i < a.NCH * 8; i += 512) xe_lds[i] = i < a.Cin ? dca_coherent_loadi(a.xexps + i) : 0;
  }
  if (tid < 32) fo_lds[tid] = dca_coherent_loadi(a.ofo + cblk * 32 + tid);
  __syncthreads();
  if (t_begin >= t_end) {
    if constexpr (STATS) {
      __syncthreads();
      fs_flush(stat_lds, 8, tid, cblk * 32, a.Cout, a.stat_part, gridDim.x, blockIdx.x);
    }
    if constexpr (EPI == 1) {   // a workgroup without tiles still owns its slot of the per-channel maxima
      if (a.y_cmax && tid < 32 && cblk * 32 + tid < a.Cout) a.y_cmax[(long)(cblk * 32 + tid) * DCA_AMAX_CSLOTS + blockIdx.x] = 0u;
    }
    return;
  }

  // per-lane byte offset of the lane's voxel inside a term image, for its two column tiles
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
    boff[t] = ((dl * IH + hl) * IW + wlane) * 16;
  }

  const int cstride = a.D * a.H * a.W;
  const long sample = (long)a.Cin * cstride;
  const int NC = a.NCH;      // channel chunks of 8 (one MFMA K half)
  const long wbytes = (long)NC * A_CHUNK;
  const __amdgpu_buffer_rsrc_t wr = dca_rsrc((const char*)a.wx + (long)cblk * wbytes, wbytes);

#if X2_STAMP
  unsigned long long* stamps = a.stamps;
  const bool stamp_on = stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (wv == 0 || wv == 7);
  if (stamp_on && wv == 7) stamps += 96 * 8;
  int stamp_k = 0;
#endif
  const bool has_aff = EPI == 1 && a.scale != nullptr, has_pre = EPI == 1 && a.res_pre != nullptr;
  const bool has_post = EPI == 2 || (EPI == 1 && a.res_post != nullptr);
  // weight fragments of a chunk: global -> registers (load_A) -> the A image that is not being read (store_A)
  float4 ra[KA];
  // `on` (0 / 1, wave uniform): the last chunk of a workgroup's last tile has nothing to stage -- its loads are masked off
  // (no control flow inside the MFMA stream) and its stores write zeros into an image nobody reads any more
  auto load_A_item = [&](int k, int chunk, int on) __attribute__((always_inline)) {
    const int it = tid + 512 * k;
    ra[k] = dca_bload4(wr, chunk * A_CHUNK + it * 16, (int)(it < NA_ITEMS) & on);
  };
  auto store_A_item = [&](int k, int buf) __attribute__((always_inline)) {
    const int it = tid + 512 * k;
    if (it < NA_ITEMS) *(float4*)(a_lds + buf * A_CHUNK + it * 16) = ra[k];
  };
  auto load_A = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) load_A_item(k, chunk, 1);
  };
  auto store_A = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KA; ++k) store_A_item(k, buf);
  };

  // Staging of a chunk's halo tile (8 channels), global -> registers (load_B) -> [split ->] the B image that is not being
  // read (store_B).
  //  PIN (packed x): words of the chunk's [term][voxel] image, word i = LDS byte 16 i: 5 per thread, pure copies.
  //  VEC (fp32 x, W % 4 == 0, 16-byte aligned): threads 0-239 own one aligned quad of 4 voxels along W (8 x b128, one per
  //  channel), 15 lanes of every wave one of the 120 edge voxels (8 x b32).  Otherwise: 3 single-voxel items of 8 x b32.
  float4 rq[(VEC && !PIN) ? 8 : 1];
  float re[(VEC && !PIN) ? 8 : 1];
  float rb[(VEC || PIN) ? 1 : KB][8];
  float4 rp[PIN ? KP : 1];
  int item_crd[PIN ? KP : (VEC ? 2 : KB)];  // packed halo coordinates of the thread's items, fixed for the whole kernel
  if constexpr (PIN) {
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int it = tid + 512 * k, term = it / NVOX, v = it - term * NVOX;
      const int id = v / (IH * IW), r2 = v - id * (IH * IW), ih = r2 / IW, iw = r2 - ih * IW;
      item_crd[k] = (it < NP_ITEMS) ? (id | (ih << 8) | (iw << 16) | (term << 25)) : -1;
    }
  } else if constexpr (VEC) {
    {
      const int row = tid >> 2, q = tid & 3, id = row / IH, ih = row - id * IH;
      item_crd[0] = (tid < NQUAD) ? (id | (ih << 8) | ((1 + 4 * q) << 16)) : -1;
    }
    {  // the 120 edge items dealt evenly over the eight waves (15 lanes each): their scattered 4-byte loads cost the
       // memory pipe one cache line per lane, so no wave should carry more of them than the others
      const int e = wv * (NEDGE / 8) + lane;
      const int row = e >> 1, side = e & 1, id = row / IH, ih = row - id * IH;
      item_crd[1] = (lane < NEDGE / 8) ? (id | (ih << 8) | ((side ? IW - 1 : 0) << 16)) : -1;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int v = tid + 512 * k;
      const int id = v / (IH * IW), rem = v - id * (IH * IW), ih = rem / IW, iw = rem - ih * IW;
      item_crd[k] = (v < NVOX) ? (id | (ih << 8) | (iw << 16)) : -1;
    }
  }
  auto item_off = [&](int crd, int d0, int h0, int w0, int chunk, int& okv) __attribute__((always_inline)) {
    const int di = d0 - 1 + (crd & 255), hi = h0 - 1 + ((crd >> 8) & 255), wi = w0 - 1 + ((crd >> 16) & 255);
    okv = (int)(crd >= 0) & (int)((unsigned)di < (unsigned)a.D) & (int)((unsigned)hi < (unsigned)a.H) &
          (int)((unsigned)wi < (unsigned)a.W);
    return (chunk * 8 * cstride + (di * a.H + hi) * a.W + wi) * 4;
  };
  // one staging step of the halo tile, k = 0 .. NLB-1: PIN: word k of the thread; VEC: channel k of its quad and of its
  // edge voxel; (otherwise load_B loads everything at once)
  constexpr int NLB = PIN ? KP : 8;
  auto load_B_item = [&](int k, __amdgpu_buffer_rsrc_t xr, int d0, int h0, int w0, int chunk, int on) __attribute__((always_inline)) {
    if constexpr (PIN) {
      // [term][channel group][voxel][8 f16]: the word of (term, group g, voxel v) sits at term * (Cin * S * 2) + (g * S + v) * 16
      const int tbytes = a.Cin * cstride * 2;
      const int crd = item_crd[k];
      const int di = d0 - 1 + (crd & 255), hi = h0 - 1 + ((crd >> 8) & 255), wi = w0 - 1 + ((crd >> 16) & 255);
      const int ok = (int)(crd >= 0) & (int)((unsigned)di < (unsigned)a.D) & (int)((unsigned)hi < (unsigned)a.H) &
                     (int)((unsigned)wi < (unsigned)a.W) & on;
      rp[k] = dca_bload4(xr, ((crd >> 25) & 1) * tbytes + (chunk * cstride + (di * a.H + hi) * a.W + wi) * 16, ok);
    } else if constexpr (VEC) {
      int okq, oke;
      const int offq = item_off(item_crd[0], d0, h0, w0, chunk, okq);  // a quad is inside W or outside as a whole
      const int offe = item_off(item_crd[1], d0, h0, w0, chunk, oke);
      const int okc = (int)(chunk * 8 + k < a.Cin) & on;
      rq[k] = dca_bload4(xr, offq + k * cstride * 4, okq & okc);
      re[k] = dca_bload1(xr, offe + k * cstride * 4, oke & okc);
    }
  };
  auto load_B = [&](int n, int d0, int h0, int w0, int chunk) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = dca_rsrc(a.x + (long)n * sample, sample * 4);
    if constexpr (PIN || VEC) {
#pragma unroll
      for (int k = 0; k < NLB; ++k) load_B_item(k, xr, d0, h0, w0, chunk, 1);
    } else {
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        int okv;
        const int off = item_off(item_crd[k], d0, h0, w0, chunk, okv);
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[k][j] = dca_bload1(xr, off + j * cstride * 4, okv & (int)(chunk * 8 + j < a.Cin));
      }
    }
  };
  auto split_store = [&](const float (&v)[8], const int* ex, char* dst) __attribute__((always_inline)) {
    const int4 e0 = *(const int4*)ex, e1 = *(const int4*)(ex + 4);     // the 8 channels' exponents (LDS table)
    const int e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    f16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 h, l;
      split2(v[j], e[j], h, l);
      hv[j] = h; lv[j] = l;
    }
    *(f16x8*)dst = hv;
    *(f16x8*)(dst + B_TERM) = lv;
  };
  auto crd_lds = [&](int crd) __attribute__((always_inline)) {  // byte offset of the item's (first) voxel in a term image
    return (((crd & 255) * IH + ((crd >> 8) & 255)) * IW + ((crd >> 16) & 255)) * 16;
  };
  // chunk = the channel chunk the staged registers belong to (its exponents); buf = the B image to fill.
  // store_B_item: one step, k = 0 .. NSB-1 (PIN: word k; VEC: voxel k of the quad, k = 4: the edge voxel)
  constexpr int NSB = PIN ? KP : 5;
  auto store_B_item = [&](int k, int chunk, int buf) __attribute__((always_inline)) {
    char* img = b_lds + buf * B_BYTES;
    if constexpr (PIN) {
      if (item_crd[k] >= 0) *(float4*)(img + (tid + 512 * k) * 16) = rp[k];
    } else if constexpr (VEC) {
      const int* ex = xe_lds + chunk * 8;
      if (k < 4) {
        if (item_crd[0] >= 0) {
          const float* q0 = (const float*)&rq[0];
          const float v[8] = {q0[k], q0[4 + k], q0[8 + k], q0[12 + k], q0[16 + k], q0[20 + k], q0[24 + k], q0[28 + k]};
          split_store(v, ex, img + crd_lds(item_crd[0]) + 16 * k);
        }
      } else if (item_crd[1] >= 0) {
        const float v[8] = {re[0], re[1], re[2], re[3], re[4], re[5], re[6], re[7]};
        split_store(v, ex, img + crd_lds(item_crd[1]));
      }
    }
  };
  auto store_B = [&](int chunk, int buf) __attribute__((always_inline)) {
    char* img = b_lds + buf * B_BYTES;
    if constexpr (PIN || VEC) {
#pragma unroll
      for (int k = 0; k < NSB; ++k) store_B_item(k, chunk, buf);
    } else {
      const int* ex = xe_lds + chunk * 8;
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (item_crd[k] >= 0) split_store(rb[k], ex, img + crd_lds(item_crd[k]));
    }
  };
  auto decode = [&](int tile, int& n, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
    int td, th, tw;
    if (a.order == 0) {
      tw = tile % a.nTW; tile /= a.nTW;
      th = tile % a.nTH; tile /= a.nTH;
      td = tile % a.nTD;
      n = tile / a.nTD;
    } else {
      td = tile % a.nTD; tile /= a.nTD;
      tw = tile % a.nTW; tile /= a.nTW;
      th = tile % a.nTH;
      n = tile / a.nTH;
    }
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
  };

  int n, d0, h0, w0;
  decode(t_begin, n, d0, h0, w0);
  load_B(n, d0, h0, w0, 0);
  load_A(0);
  store_B(0, 0);
  store_A(0);
  __syncthreads();

  int buf = 0;  // image pair (A, B) of the current chunk; chunks alternate buffers across tile boundaries
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    const bool more_tiles = tile + t_step < t_end;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int nn = n, nd0 = d0, nh0 = h0, nw0 = w0;  // coordinates of the next tile (valid when more_tiles)
    if (more_tiles) decode(tile + t_step, nn, nd0, nh0, nw0);

#pragma unroll 1
    for (int chunk = 0; chunk < NC; ++chunk, buf ^= 1) {
      const bool last_chunk = chunk + 1 == NC;
      // The halo tile and the weight fragments of the NEXT chunk (of chunk 0 of the next tile) are fetched while this chunk
      // computes and written into the other image pair: nobody reads that pair before the barrier at the end of the chunk,
      // so the loads, the [split and] LDS stores and the MFMAs of a chunk need no ordering among themselves -- ONE barrier
      // per chunk of 84 MFMAs per wave (round 2: three phases of 54 with a barrier each, the slab stores in front of each
      // phase and the halo image written between two extra barriers after every third).
      const bool stage = !last_chunk || more_tiles;
      const bool next_tile = last_chunk && more_tiles;
      X2_MARK(0);
      // staging steps ride between the MFMAs (one global load / one LDS store per tap pair, see the pair loop): issued as a
      // burst in front of the chunk, 9 loads and their address arithmetic kept BOTH waves of a SIMD -- and the matrix pipe --
      // busy for 1.0-2.4 k of a chunk's 8.7 k cycles, and the stores behind the MFMAs cost another 0.6 k (s_memtime stamps,
      // tools/x2_stamps.py); only the unaligned fp32 path (W % 4 != 0) still stages that way
      const int s_n = next_tile ? nn : n, s_d0 = next_tile ? nd0 : d0, s_h0 = next_tile ? nh0 : h0, s_w0 = next_tile ? nw0 : w0;
      const int s_chunk = next_tile ? 0 : chunk + 1, s_on = stage ? 1 : 0;
      const __amdgpu_buffer_rsrc_t s_xr = dca_rsrc(a.x + (long)s_n * sample, sample * 4);
      if constexpr (!PIN && !VEC) {
        if (stage) {
          load_B(s_n, s_d0, s_h0, s_w0, s_chunk);
          load_A(s_chunk);
        }
      }
      const char* ab = a_lds + buf * A_CHUNK + lane * 16;
      const char* bb = b_lds + buf * B_BYTES;
      // 14 tap pairs with a register double buffer: the 6 ds_read_b128 of pair p+1 go one per MFMA between the 6 MFMAs of
      // pair p, the staging loads one per pair behind them.  sched_group_barrier pins the order; left alone, hipcc issues
      // each LDS read right before its first use and waits on it.
      f16x8 fa[2][NT], fb[2][2][NT];
      auto load_frag = [&](int p, int slot) __attribute__((always_inline)) {
        // lanes 0-31 hold k = the chunk's 8 channels at tap 2p, lanes 32-63 the same channels at tap 2p+1 (the 28th "tap"
        // has zero weights: any finite B data will do, tap 26's)
        const int t0 = 2 * p, t1 = (2 * p + 1 < 27) ? 2 * p + 1 : 26;
        const int o0 = (((t0 / 9) * IH + (t0 / 3) % 3) * IW + t0 % 3) * 16, o1 = (((t1 / 9) * IH + (t1 / 3) % 3) * IW + t1 % 3) * 16;
        const int toff = half ? o1 : o0;
#pragma unroll
        for (int term = 0; term < NT; ++term) fa[slot][term] = *(const f16x8*)(ab + (p * NT + term) * 1024);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int term = 0; term < NT; ++term)
            fb[slot][t][term] = *(const f16x8*)(bb + boff[t] + toff + term * B_TERM);
      };
      X2_MARK(1);
      load_frag(0, 0);
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        const int cur = p & 1;
        if (p + 1 < NPAIR) load_frag(p + 1, cur ^ 1);
        // smallest terms first; the two column tiles alternate so consecutive MFMAs never chain on one accumulator
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][PA[q]], fb[cur][t][PB[q]], acc[t], 0, 0, 0);
        // this pair's share of the staging (compile-time schedule; p is a constant of the unrolled loop):
        //   loads:  halo step p (p < NLB), weight word p - LA0: everything is requested by pair 7
        //   stores: halo step p - SB0 and weight word p - SA0, a few pairs (~0.4 k cycles each) after their loads
        constexpr int SB0 = PIN ? 6 : 9, SA0 = 10, LA0 = PIN ? 0 : 4;
        if constexpr (PIN || VEC) {
          if (p >= LA0 && p - LA0 < KA) load_A_item(p - LA0, s_chunk, s_on);
          if (p < NLB) load_B_item(p, s_xr, s_d0, s_h0, s_w0, s_chunk, s_on);
          if (p >= SB0 && p - SB0 < NSB) store_B_item(p - SB0, s_chunk, buf ^ 1);
          if (p >= SA0 && p - SA0 < KA) store_A_item(p - SA0, buf ^ 1);
        }
        if (p + 1 < NPAIR) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read (the next pair's fragments)
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // nothing moves across pairs: the staging steps stay where they are written
      }
      X2_MARK(2);
      if constexpr (!PIN && !VEC) {
        if (stage) {
          store_B(s_chunk, buf ^ 1);
          store_A(buf ^ 1);
        }
      }
      X2_MARK(3);
      // the barrier of a tile's last chunk comes AFTER the epilogue: the epilogue touches neither image pair, so the older
      // wave of a SIMD (which wins the matrix pipe and finishes its 84 MFMAs ~2 k cycles early) stores its results while the
      // younger one still computes
      if (!last_chunk) __syncthreads();
      X2_MARK(5);
#if X2_STAMP
      if (chunk + 1 < NC) ++stamp_k;
#endif
    }
    X2_MARK(6);

    // Epilogue (same contract as conv3d_mfma.hip): y = act(acc * scale + shift + res_pre) + res_post.  The stores
    // drain while the next tile's first phase runs.
    // 32-bit offsets into this sample's output through buffer descriptors: nothing 64-bit for the compiler to hoist
    // out of the tile loop (that cost ~60 spilled registers), and the bounds masks ride on the hardware range check.
    const long osample = (long)a.Cout * cstride;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.y + (long)n * osample, osample * 4);
    const __amdgpu_buffer_rsrc_t pr = dca_rsrc((has_pre ? a.res_pre : a.y) + (long)n * osample, osample * 4);
    const __amdgpu_buffer_rsrc_t qr = dca_rsrc((has_post ? a.res_post : a.y) + (long)n * osample, osample * 4);
    float sc[EPI == 1 ? 16 : 1], sh[EPI == 1 ? 16 : 1];  // (re)loaded per tile: holding them across the MFMA phases costs 32 registers
    if constexpr (EPI == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = min(cblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
        sc[r] = has_aff ? a.scale[co] : 1.f;
        sh[r] = has_aff ? a.shift[co] : 0.f;
      }
    }
    int nfo[16];   // -f_o of the lane's 16 output channels: the accumulator is scaled back with v_ldexp_f32 (exact)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 f4 = *(const int4*)(fo_lds + 8 * q + 4 * half);
      nfo[4 * q] = -f4.x; nfo[4 * q + 1] = -f4.y; nfo[4 * q + 2] = -f4.z; nfo[4 * q + 3] = -f4.w;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r0 = (wv * 2 + t) * 2 + (l31 >> 4), d = d0 + (r0 >> 3), h = h0 + (r0 & 7), w = w0 + wlane;
      const int ok = (int)(d < a.D) & (int)(h < a.H) & (int)(w < a.W);
      if constexpr (STATS) st_n += (float)ok;
      const int voff = ((d * a.H + h) * a.W + w + (cblk * 32 + 4 * half) * cstride) * 4;
      float rp[16], rq[16];
      if (has_pre) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = (r & 3) + 8 * (r >> 2);
          rp[r] = dca_bload1(pr, voff + cu * cstride * 4, ok & (int)(cblk * 32 + cu + 4 * half < a.Cout));
        }
      }
      if (has_post) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = (r & 3) + 8 * (r >> 2);
          rq[r] = dca_bload1(qr, voff + cu * cstride * 4, ok & (int)(cblk * 32 + cu + 4 * half < a.Cout));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = (r & 3) + 8 * (r >> 2);
        float v = ldexpf(acc[t][r], nfo[r]);
        if constexpr (EPI == 1) {
          v = v * sc[r] + sh[r];
          if (has_pre) v += rp[r];
          v = act_apply(v, a.slope);
          if (has_post) v += rq[r];
        } else if constexpr (EPI == 2) {
          v += rq[r];
        }
        if constexpr (STATS) {
          if (stat_first && t == 0) {   // the wave's first tile: the shift of (half, r) = what lane 0 of the half produced
            st_k[r] = fs_half_first(v, half);
            if ((lane & 31) == 0) fs_slot(stat_w, half, r)[0] = st_k[r];     // fs_flush reads it there
          }
          const float dlt = ok ? v - st_k[r] : 0.f;     // (an LDS read of K per value cost the statistics epilogue ~1 k cycles)
          st_s[r] += dlt;
          st_q[r] = fmaf(dlt, dlt, st_q[r]);
        }
        const int okc = ok & (int)(cblk * 32 + cu + 4 * half < a.Cout);
        if constexpr (EPI == 1) ycm[r] = fmaxf(ycm[r], okc ? fabsf(v) : 0.f);
#if X2_NT
        dca_bstore1_nt(yr, v, voff + cu * cstride * 4, okc);
#else
        dca_bstore1(yr, v, voff + cu * cstride * 4, okc);
#endif
      }
    }
    if constexpr (STATS) stat_first = false;
    X2_MARK(7);
    __syncthreads();     // (the last chunk's barrier: the next tile's first images are complete, this tile's are free)
#if X2_STAMP
    ++stamp_k;
#endif
    n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
  }
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float rs = fs_half_sum(st_s[r]), rq2 = fs_half_sum(st_q[r]);
      if ((lane & 31) == 0) {   // wave-private slot, one writer lane
        fs_slot(stat_w, half, r)[1] = rs;
        fs_slot(stat_w, half, r)[2] = rq2;
      }
    }
    const float rn = fs_half_sum(st_n);
    if ((lane & 31) == 0) stat_w[96 + half] = rn;
    __syncthreads();
    fs_flush(stat_lds, 8, tid, cblk * 32, a.Cout, a.stat_part, gridDim.x, blockIdx.x);
  }
  if constexpr (EPI == 1) {
    if (a.y_cmax) {   // the consumer's per-channel operand maxima, for the next f16x2 convolution: slot [channel][workgroup]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float m = ycm[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));     // over the 32 lanes of the wave half
        if (l31 == 0) ycm_lds[wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = m;
      }
      __syncthreads();
      if (tid < 32 && cblk * 32 + tid < a.Cout) {
        float m = ycm_lds[tid];
        for (int w = 1; w < 8; ++w) m = fmaxf(m, ycm_lds[w * 32 + tid]);
        a.y_cmax[(long)(cblk * 32 + tid) * DCA_AMAX_CSLOTS + blockIdx.x] = __float_as_uint(m);
      }
    }
  }
}

// Weight packing, once per launch (it needs the exponents of the operand the convolution is about to read).  One workgroup
// per block of 32 output channels:
//   1. xexps[k]: given, or (slots != null) derived here from the operand's per-channel maxima and written out for the
//      weight-gradient kernel that reads the same operand later;
//   2. f_o = 14 - max over (k, tap) of (exponent of w[o][k][tap]) - xexps[k]: the row's largest scaled entry in [2^14, 2^15);
//   3. wx[cblk][chunk of 8 channels][tap pair p][term][lane][j] (f16): lane (r = lane & 31, h = lane >> 5) holds
//      A[row = output channel cblk*32 + r][k = (input channel chunk*8 + j, tap 2p + h)] = w 2^(f_o - xexps[channel]), split
//      into term 0/1 = h/l; zero padded (channels beyond Cin, the 28th tap);  f_o goes behind the images (ofo[cblk*32 + r]).
// Source indexing as dca_conv3d_prep_weight: src_ab ? src[a][b][27] : src[b][a][27]; flip reverses the tap order.
constexpr int PREP_ROWS = 4;       // output channels per workgroup of the packing kernel (8 workgroups per block of 32)
__global__ __launch_bounds__(512) void x2_prep_weight_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                              int A, int Bn, int NCH, int src_ab, int flip,
                                                              const unsigned* __restrict__ slots, int nslots,
                                                              int* __restrict__ xexps, int xexps_given, int* __restrict__ ofo) {
  __shared__ int xe[MAX_CIN];
  __shared__ int rowmax[PREP_ROWS][2];
  const int tid = threadIdx.x, cblk = blockIdx.x / (32 / PREP_ROWS), r0 = (blockIdx.x % (32 / PREP_ROWS)) * PREP_ROWS;
  if (slots && !xexps_given) {     // 16 threads per channel, 32 channels per round; every workgroup derives its own copy
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
  // row r = tid >> 7 (128 threads = 2 waves per output channel): exponent of the largest |w 2^-xexps[k]| of the row
  {
    const int r = tid >> 7, l = tid & 127, bi = cblk * 32 + r0 + r;
    int m = -100000;
    if (bi < Bn) {
      for (int i = l; i < A * 27; i += 128) {
        const int ai = i / 27, tap = i - ai * 27;
        const float v = src_ab ? src[((long)ai * Bn + bi) * 27 + tap] : src[((long)bi * A + ai) * 27 + tap];
        const int be = (int)((__float_as_uint(v) >> 23) & 255);      // biased exponent; 0: zero / denormal -> ignored
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
    ofo[cblk * 32 + r0 + tid] = fo;
  }
  __syncthreads();
  // items (chunk of 8 channels, tap pair, k half = tap of the pair, row): the 8 input channels of one lane's fragment,
  // both terms: two 16-byte stores
  const int nitems = NCH * NPAIR * 2 * PREP_ROWS;
  unsigned short* out = dst + (long)cblk * NCH * NPAIR * NT * 512;
  for (int it = tid; it < nitems; it += 512) {
    const int r = it % PREP_ROWS, hf = (it / PREP_ROWS) & 1, t = it / (2 * PREP_ROWS), pair = t % NPAIR, chunk = t / NPAIR;
    const int tap = 2 * pair + hf;
    const int bi = cblk * 32 + r0 + r, st = flip ? 26 - tap : tap, fo = rowmax[r][0];
    f16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ai = chunk * 8 + j;
      float v = 0.f;
      int e = 0;
      if (ai < A && bi < Bn && tap < 27) {
        v = src_ab ? src[((long)ai * Bn + bi) * 27 + st] : src[((long)bi * A + ai) * 27 + st];
        e = fo - xe[ai];
      }
      _Float16 h, l;
      split2(v, e, h, l);
      hv[j] = h; lv[j] = l;
    }
    const long o = ((long)(chunk * NPAIR + pair) * NT) * 512 + (hf * 32 + r0 + r) * 8;      // f16 elements; term stride 512
    *(f16x8*)(out + o) = hv;
    *(f16x8*)(out + o + 512) = lv;
  }
}

}  // namespace

// bytes of the packed image: fragments, then f_o (one int per output channel, padded to blocks of 32)
extern "C" long dca_conv3d_x2_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (long)((Cout + 31) / 32) * ((Cin + 7) / 8) * A_CHUNK + (long)((Cout + 31) / 32) * 32 * 4;
}

// Packs w for ONE convolution launch over an operand with the per-channel scale exponents xexps (A ints).
//   x_slots == null: xexps is an input (the packed px2 operand's exponents, or a previous call's output);
//   x_slots != null: the operand's per-channel maxima (slots[c * DCA_AMAX_CSLOTS + s], s < nslots): xexps is derived from
//                    them and WRITTEN (the weight gradient of the same operand reads it later).
extern "C" int dca_conv3d_x2_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip,
                                         const unsigned* x_slots, int nslots, int* xexps, hipStream_t stream) {
  DCA_REQUIRE(w && wx && xexps && A > 0 && A <= MAX_CIN && B > 0 && ((((uintptr_t)wx) & 15) == 0));
  DCA_REQUIRE(x_slots == nullptr || (nslots > 0 && nslots <= DCA_AMAX_CSLOTS));
  const int NCH = (A + 7) / 8, cblks = (B + 31) / 32;
  int* ofo = (int*)((char*)wx + (long)cblks * NCH * A_CHUNK);
  hipLaunchKernelGGL(x2_prep_weight_kernel, dim3(cblks * (32 / PREP_ROWS)), dim3(512), 0, stream, w, (unsigned short*)wx, A, B,
                     NCH, src_ab, flip, x_slots, nslots, xexps, x_slots == nullptr ? 1 : 0, ofo);
  return dca_launch_status();
}

#if X2_STAMP
static unsigned long long* g_stamps = nullptr;
extern "C" void dca_x2_debug_set_stamps(unsigned long long* p) { g_stamps = p; }
#endif

namespace {

int x2_grid(long tiles, int cblks) {
  // persistent: one workgroup per CU, each looping over its share of the tiles
  int ncu = 256;   // per device, so not cached in a static
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      ncu = v;
  }
  int gx = ncu / cblks > 0 ? ncu / cblks : 1;
  if (gx > tiles) gx = (int)tiles;
  if (gx > DCA_AMAX_CSLOTS) gx = DCA_AMAX_CSLOTS;      // a workgroup index is also a slot of the per-channel output maxima
  return gx;
}

template <bool VEC, bool STATS, int EPI, bool PIN>
int x2_go(const X2Args& a, int gx, int cblks, int lds, hipStream_t stream) {
  auto kern = conv3_f16x2_kernel<VEC, STATS, EPI, PIN>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(gx, cblks), dim3(512), lds, stream, a);
  return dca_launch_status();
}

int x2_launch(const void* x, int packed, const int* xexps, const void* wx, float* y, const float* scale, const float* shift,
              const float* res_pre, const float* res_post, float slope, double* stat_part, unsigned* y_cmax, int N,
              int Cin, int Cout, int D, int H, int W, hipStream_t stream) {
  DCA_REQUIRE(x && wx && y && N > 0 && Cin > 0 && Cin <= MAX_CIN && Cout > 0 && D > 0 && H > 0 && W > 0);
  DCA_REQUIRE(packed ? (Cin % 8 == 0) : (xexps != nullptr));
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE((long)Cin * D * H * W * 4 < 0x7ffffff0L && (long)Cout * D * H * W * 4 < 0x7ffffff0L);  // 32-bit byte offsets inside one sample
  DCA_REQUIRE((((uintptr_t)wx) & 15) == 0 && (!packed || (((uintptr_t)x) & 15) == 0));
  X2Args a;
  a.x = (const float*)x; a.wx = (const unsigned short*)wx; a.y = y;
  a.scale = scale; a.shift = shift; a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.NCH = (Cin + 7) / 8;
  a.D = D; a.H = H; a.W = W;
  a.nTD = cdiv(D, TD); a.nTH = cdiv(H, TH); a.nTW = cdiv(W, TW);
  a.stat_part = stat_part;
  {
    static const int order = [] { const char* e = getenv("DCA_X2_ORDER"); return e ? atoi(e) : 0; }();
    a.order = order;
  }
  a.xexps = xexps;
  a.y_cmax = y_cmax;
#if X2_STAMP
  a.stamps = g_stamps;
#endif
  const int cblks = (Cout + 31) / 32;
  a.ofo = (const int*)((const char*)wx + (long)cblks * a.NCH * A_CHUNK);
  const long tiles = (long)N * a.nTD * a.nTH * a.nTW;
  DCA_REQUIRE(tiles < 0x7fffffffL && cblks <= 65535);
  const bool vec = (W % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
  const bool stats = stat_part != nullptr;
  const bool full = scale != nullptr || res_pre != nullptr || slope != 1.f || y_cmax != nullptr;
  const int epi = full ? 1 : (res_post != nullptr ? 2 : 0);
  DCA_REQUIRE(!(stats && epi));    // the statistics are those of the raw convolution output
  const int lds = LDS_BYTES + TAB_BYTES + (stats ? STAT_LDS : 0);
  const int gx = x2_grid(tiles, cblks);
  if (packed) {
    DCA_REQUIRE(epi != 1);         // the packed operand exists in training only (BatchNorm kernels write it)
    if (stats) return x2_go<true, true, 0, true>(a, gx, cblks, lds, stream);
    if (epi == 2) return x2_go<true, false, 2, true>(a, gx, cblks, lds, stream);
    return x2_go<true, false, 0, true>(a, gx, cblks, lds, stream);
  }
  if (stats) return vec ? x2_go<true, true, 0, false>(a, gx, cblks, lds, stream) : x2_go<false, true, 0, false>(a, gx, cblks, lds, stream);
  if (epi == 1) return vec ? x2_go<true, false, 1, false>(a, gx, cblks, lds, stream) : x2_go<false, false, 1, false>(a, gx, cblks, lds, stream);
  if (epi == 2) return vec ? x2_go<true, false, 2, false>(a, gx, cblks, lds, stream) : x2_go<false, false, 2, false>(a, gx, cblks, lds, stream);
  return vec ? x2_go<true, false, 0, false>(a, gx, cblks, lds, stream) : x2_go<false, false, 0, false>(a, gx, cblks, lds, stream);
}

}  // namespace

// y = act(conv(x, w) * scale[c] + shift[c] + res_pre) + res_post (the epilogue contract of dca_conv3d_forward) with the
// f16x2 split arithmetic.  x: fp32 (N,Cin,D,H,W) with xexps = its per-channel scale exponents (Cin ints: output of
// dca_conv3d_x2_prep_weight / dca_cmax_exps), or packed != 0: the px2 image of that tensor (xexps unused: the exponents
// are in the packed weights).  wx = dca_conv3d_x2_prep_weight(...) for THIS operand.  y_cmax (may be null): per-channel
// slots [c][s], s < dca_conv3d_x2_out_slots(...), that receive max |y| for the consumer of y.
extern "C" int dca_conv3d_x2_forward(const void* x, int packed, const int* xexps, const void* wx, float* y, const float* scale,
                                     const float* shift, const float* res_pre, const float* res_post, float slope,
                                     unsigned* y_cmax, int N, int Cin, int Cout, int D, int H, int W, hipStream_t stream) {
  return x2_launch(x, packed, xexps, wx, y, scale, shift, res_pre, res_post, slope, nullptr, y_cmax, N, Cin, Cout, D, H, W,
                   stream);
}

// nchunk of the statistics dca_conv3d_x2_forward_stats produces (one partial per workgroup of the launch it will make);
// also the number of y_cmax slots per channel dca_conv3d_x2_forward fills
extern "C" long dca_conv3d_x2_stats_chunks(int N, int Cout, int D, int H, int W) {
  if (N <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const long tiles = (long)N * cdiv(D, TD) * cdiv(H, TH) * cdiv(W, TW);
  return x2_grid(tiles, (Cout + 31) / 32);
}

// y = conv(x, w) (no epilogue) plus the BatchNorm batch statistics of y: part (Cout * nchunk * 4 doubles, nchunk =
// dca_conv3d_x2_stats_chunks) = one {K, n, sum (y - K), sum (y - K)^2} per (channel, workgroup), for
// dca_bn_finalize_centered (bn_fused_stats.h)
extern "C" int dca_conv3d_x2_forward_stats(const void* x, int packed, const int* xexps, const void* wx, float* y,
                                           double* stat_part, int N, int Cin, int Cout, int D, int H, int W,
                                           hipStream_t stream) {
  DCA_REQUIRE(stat_part);
  return x2_launch(x, packed, xexps, wx, y, nullptr, nullptr, nullptr, nullptr, 1.f, stat_part, nullptr, N, Cin, Cout, D, H, W,
                   stream);
}
