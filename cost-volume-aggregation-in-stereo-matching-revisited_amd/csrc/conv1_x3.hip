// 1x1x1 convolution (pointwise GEMM, <= 32 output channels per launch) with fp32 tensors and fp32-grade accuracy on the
// bf16 matrix pipe: both operands split exactly into three bf16 terms, the six partial products >= 2^-16 accumulated in
// fp32 (the "bf16x3" arithmetic of conv3d_bf16x3.hip), on the LDS-free data path of conv1_lp.hip.
//
// Reference operators served (forward and, with the transposed weight, backward-data): the q / k / v / out projections of
// SelfAttentionBlock (models/augment/SelfAttention_bn.py:136-160), `cva.fuse` over two inputs without the concat
// (models/augment/cva.py:55,69), `cost_agg.redir` (cva.py:23) and the tap-expansion GEMM of the 32 -> 1 logit heads.
//
// Why: the fp32-MFMA form (conv3d_mfma.hip, v_mfma_f32_32x32x2f32) needs 64 MFMAs = 4096 matrix-pipe cycles per 128
// voxels and wave, during which the wave has nothing in flight -- 3.2-3.6 TB/s at one wave per SIMD.  Here a 16-channel
// chunk is 24 MFMAs of 32 cycles (6 products x 4 column tiles), the split costs 11 VALU instructions per channel pair
// (v_cvt_pk_bf16_f32 on pairs, exact residuals), the register budget allows two workgroups per CU, and the kernel sits
// on its HBM stream.
//
// A wave owns 128 consecutive voxels; lane (r = lane & 31, h = lane >> 5) loads, for the 8 channels 8h..8h+7 of a chunk,
// the four voxels 4r..4r+3 as one 16-byte load.  MFMA column tile t is the voxel set {4r + t}: the lane's B fragment of
// a term is the 8 channels of its own voxel 4r + t -- four dwords, each the packed pair (channel 2j, 2j+1) -- built in
// registers.  Per output channel the four tiles give the lane's four voxels again: one 16-byte store.
#include "dca_common.h"
#include "bn_fused_stats.h"
#include "../../include/dca_hip.h"

typedef __bf16 cx_bf16x8 __attribute__((ext_vector_type(8)));

namespace {

struct C1XArgs {
  const float* x;       // (N, C1, S)
  const float* x2;      // (N, C2, S) or null
  const unsigned short* wfrag;   // [chunk][term][lane][8] bf16 A fragments
  float* y;             // (N, CoutTotal, S): this launch writes channels [co_off, co_off + Cout)
  const float* scale;
  const float* shift;
  const float* res_pre;
  const float* res_post;
  float slope;
  int N, C1, C2, Cout, CoutTotal, co_off;
  long S;
  double* stat_part;         // STATS: one partial {K, n, s, q} per (channel of CoutTotal, workgroup): bn_fused_stats.h
};

__device__ __forceinline__ unsigned cx_pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bfx2));
}

// NC1 / NC2: 16-channel chunks of the first / second input (C1 = 16 * NC1 exactly; S % 4 == 0, aligned bases).
// STATS: the output feeds a training-mode BatchNorm; the kernel also emits the per-(channel, workgroup) partial statistics
// of bn_fused_stats.h.
template <int NC1, int NC2, bool STATS>
__global__ __launch_bounds__(256, 2) void conv1_x3_kernel(C1XArgs a) {
  constexpr int NCH = NC1 + NC2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  __shared__ float aff[64];
  __shared__ float stat_lds[STATS ? 4 * FS_WAVE_FLOATS : 1];
  float* stat_w = stat_lds + (STATS ? wv * FS_WAVE_FLOATS : 0);
  bool stat_first = true;
  if constexpr (STATS) {
    for (int i = threadIdx.x; i < 4 * FS_WAVE_FLOATS; i += 256) stat_lds[i] = 0.f;
  }
  const bool has_aff = a.scale != nullptr, has_pre = a.res_pre != nullptr, has_post = a.res_post != nullptr;
  if (threadIdx.x < 64) {
    const int co = min((int)(threadIdx.x & 31), a.Cout - 1);
    aff[threadIdx.x] = has_aff ? (threadIdx.x < 32 ? a.scale[a.co_off + co] : a.shift[a.co_off + co])
                               : (threadIdx.x < 32 ? 1.f : 0.f);
  }
  cx_bf16x8 wf[NCH][3];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < 3; ++t) wf[c][t] = *(const cx_bf16x8*)(a.wfrag + (((long)c * 3 + t) * 64 + lane) * 8);
  __syncthreads();

  constexpr bool LANE_ACC = STATS && NCH <= 2;   // the 64-channel forms have no 32 registers to spare
  float st_s[LANE_ACC ? 16 : 1], st_q[LANE_ACC ? 16 : 1], st_n = 0.f;
  if constexpr (LANE_ACC) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st_s[r] = st_q[r] = 0.f;
  }
  const long ngroups = (a.S + 127) / 128, total = (long)a.N * ngroups;
  for (long g = (long)blockIdx.x * 4 + wv; g < total; g += (long)gridDim.x * 4) {
    const int n = (int)(g / ngroups);
    const long v = (g - (long)n * ngroups) * 128 + 4 * l31;
    const int inr = (int)(v < a.S);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const bool first = c < NC1;
      const int cc = first ? c : c - NC1, C = first ? a.C1 : a.C2;
      const __amdgpu_buffer_rsrc_t xr = dca_rsrc((first ? a.x : a.x2) + (long)n * C * a.S, (long)C * a.S * 4);
      const int base = dca_pred_off((int)((((long)(cc * 16 + 8 * half)) * a.S + v) * 4), inr);
      float4 q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = dca_bload4(xr, base + (int)(j * a.S * 4), 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        u32x4 H, M, L;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float va = t == 0 ? q[2 * j].x : (t == 1 ? q[2 * j].y : (t == 2 ? q[2 * j].z : q[2 * j].w));
          const float vb = t == 0 ? q[2 * j + 1].x : (t == 1 ? q[2 * j + 1].y : (t == 2 ? q[2 * j + 1].z : q[2 * j + 1].w));
          const unsigned h2 = cx_pack2(va, vb);
          const float ra = va - __uint_as_float(h2 << 16), rb = vb - __uint_as_float(h2 & 0xffff0000u);      // exact
          const unsigned m2 = cx_pack2(ra, rb);
          const unsigned l2 = cx_pack2(ra - __uint_as_float(m2 << 16), rb - __uint_as_float(m2 & 0xffff0000u));
          H[j] = h2; M[j] = m2; L[j] = l2;
        }
        const cx_bf16x8 bh = __builtin_bit_cast(cx_bf16x8, H), bm = __builtin_bit_cast(cx_bf16x8, M),
                        bl = __builtin_bit_cast(cx_bf16x8, L);
        // smallest terms first (as conv3d_bf16x3.hip): w_h x_l, w_l x_h, w_m x_m, w_h x_m, w_m x_h, w_h x_h
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][2], bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], bh, acc[t], 0, 0, 0);
      }
    }
    // epilogue: y = act(acc * scale + shift + res_pre) + res_post on this launch's channel slice.  One predicated base per
    // lane; an output channel >= Cout of the slice lies beyond the slice descriptor's range and is dropped.
    const long slice = (long)a.Cout * a.S;
    const long obase = ((long)n * a.CoutTotal + a.co_off) * a.S;
    const __amdgpu_buffer_rsrc_t yr = dca_rsrc(a.y + obase, slice * 4);
    const __amdgpu_buffer_rsrc_t pr = dca_rsrc((has_pre ? a.res_pre : a.y) + obase, slice * 4);
    const __amdgpu_buffer_rsrc_t qr = dca_rsrc((has_post ? a.res_post : a.y) + obase, slice * 4);
    const int ob = dca_pred_off((int)(((long)(4 * half) * a.S + v) * 4), inr);
#pragma unroll
    for (int rc = 0; rc < 16; rc += 4) {
      float4 rp[4], rq[4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) rp[q4] = rq[q4] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_pre) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int r = rc + q4;
          rp[q4] = dca_bload4(pr, ob + (int)(((r & 3) + 8 * (r >> 2)) * a.S * 4), 1);
        }
      }
      if (has_post) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int r = rc + q4;
          rq[q4] = dca_bload4(qr, ob + (int)(((r & 3) + 8 * (r >> 2)) * a.S * 4), 1);
        }
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int r = rc + q4, cl = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float sc = aff[cl], sh = aff[32 + cl];
        const float o0 = act_apply(acc[0][r] * sc + sh + rp[q4].x, a.slope) + rq[q4].x;
        const float o1 = act_apply(acc[1][r] * sc + sh + rp[q4].y, a.slope) + rq[q4].y;
        const float o2 = act_apply(acc[2][r] * sc + sh + rp[q4].z, a.slope) + rq[q4].z;
        const float o3 = act_apply(acc[3][r] * sc + sh + rp[q4].w, a.slope) + rq[q4].w;
        const u32x4 w_ = {__float_as_uint(o0), __float_as_uint(o1), __float_as_uint(o2), __float_as_uint(o3)};
        __builtin_amdgcn_raw_buffer_store_b128(w_, yr, ob + (int)(((r & 3) + 8 * (r >> 2)) * a.S * 4), 0, 0);
        if constexpr (STATS) {
          if (stat_first) {   // the wave's first group: the shift of (half, r) = what lane 0 of the half produced
            const float kf = fs_half_first(o0, half);
            if ((lane & 31) == 0) fs_slot(stat_w, half, r)[0] = kf;
          }
          const float k = fs_slot(stat_w, half, r)[0];
          const float d0 = inr ? o0 - k : 0.f, d1 = inr ? o1 - k : 0.f, d2 = inr ? o2 - k : 0.f, d3 = inr ? o3 - k : 0.f;
          const float ps = (d0 + d1) + (d2 + d3), pq = fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3);
          if constexpr (LANE_ACC) {   // per-lane running sums, reduced over lanes and waves once at the end of the kernel
            st_s[r] += ps;
            st_q[r] += pq;
          } else {                    // no registers to spare: reduce the wave half now, one LDS add per (half, channel)
            const float rs = fs_half_sum(ps), rq2 = fs_half_sum(pq);
            if ((lane & 31) == 0) {
              atomicAdd(fs_slot(stat_w, half, r) + 1, rs);   // ds_add_f32 without return; wave-private, one writer lane
              atomicAdd(fs_slot(stat_w, half, r) + 2, rq2);
            }
          }
        }
      }
    }
    if constexpr (STATS) {
      st_n += 4.f * (float)inr;
      stat_first = false;
    }
  }
  if constexpr (STATS) {
    if constexpr (LANE_ACC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rs = fs_half_sum(st_s[r]), rq2 = fs_half_sum(st_q[r]);
        if ((lane & 31) == 0) {
          fs_slot(stat_w, half, r)[1] = rs;
          fs_slot(stat_w, half, r)[2] = rq2;
        }
      }
    }
    const float rn = fs_half_sum(st_n);
    if ((lane & 31) == 0) stat_w[96 + half] = rn;
    __syncthreads();
    fs_flush(stat_lds, 4, threadIdx.x, a.co_off, a.co_off + a.Cout, a.stat_part, gridDim.x, blockIdx.x);
  }
}

__device__ __forceinline__ void cx_split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  const float r1 = v - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// wfrag[chunk][term][lane][j] = term of W[b = lane & 31][a = chunk*16 + 8*(lane >> 5) + j], W[b][a] = src_ab ?
// w[a*Btotal + b_off + b] : w[(b_off + b)*A + a], zero for b >= Bn or a >= A.
__global__ void conv1_x3_prep_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int A, int Bn,
                                     int src_ab, int Btotal, int b_off, int total) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, term = (idx >> 9) % 3, chunk = (idx >> 9) / 3;
    const int bi = lane & 31, ai = chunk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (ai < A && bi < Bn) v = src_ab ? w[(long)ai * Btotal + b_off + bi] : w[(long)(b_off + bi) * A + ai];
    __bf16 h, m, l;
    cx_split3(v, h, m, l);
    const __bf16 o = term == 0 ? h : (term == 1 ? m : l);
    dst[idx] = __builtin_bit_cast(unsigned short, o);
  }
}

}  // namespace

extern "C" long dca_conv1_x3_weight_bytes(int A) {
  if (A <= 0) return 0;
  return (long)((A + 15) / 16) * 3 * 1024;
}

extern "C" int dca_conv1_x3_prep_weight(const float* w, void* wfrag, int A, int Bn, int src_ab, int Btotal, int b_off,
                                        hipStream_t stream) {
  DCA_REQUIRE(w && wfrag && A > 0 && Bn > 0 && Bn <= 32 && b_off >= 0 && b_off + Bn <= Btotal);
  const int total = (int)(dca_conv1_x3_weight_bytes(A) / 2);
  hipLaunchKernelGGL(conv1_x3_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, w, (unsigned short*)wfrag, A, Bn,
                     src_ab, Btotal, b_off, total);
  return dca_launch_status();
}

namespace {

long c1x_blocks(int N, long S) {
  const long groups = (long)N * ((S + 127) / 128);
  long blocks = (groups + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return blocks;
}

int c1x_launch(const float* x, const float* x2, const void* wfrag, float* y, const float* scale, const float* shift,
               const float* res_pre, const float* res_post, float slope, double* stat_part, int N, int C1, int C2, int Cout,
               int CoutTotal, int co_off, long S, hipStream_t stream) {
  DCA_REQUIRE(x && wfrag && y && N > 0 && Cout > 0 && Cout <= 32 && co_off >= 0 && co_off + Cout <= CoutTotal && S > 0);
  DCA_REQUIRE((x2 != nullptr) == (C2 > 0));
  DCA_REQUIRE((C1 == 32 && C2 == 0) || (C1 == 64 && C2 == 0) || (C1 == 32 && C2 == 32));
  DCA_REQUIRE((scale == nullptr) == (shift == nullptr));
  DCA_REQUIRE(S % 4 == 0 && ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)y | (uintptr_t)res_pre | (uintptr_t)res_post |
                               (uintptr_t)wfrag)) & 15) == 0);
  DCA_REQUIRE(64L * S * 4 < 0x7ffffff0L);   // 32-bit byte offsets inside one sample (+ the out-of-range marker)
  C1XArgs a;
  a.x = x; a.x2 = x2; a.wfrag = (const unsigned short*)wfrag; a.y = y; a.scale = scale; a.shift = shift;
  a.res_pre = res_pre; a.res_post = res_post; a.slope = slope;
  a.N = N; a.C1 = C1; a.C2 = C2; a.Cout = Cout; a.CoutTotal = CoutTotal; a.co_off = co_off; a.S = S;
  a.stat_part = stat_part;
  const dim3 grid((int)c1x_blocks(N, S));
  if (stat_part) {
    if (C2) hipLaunchKernelGGL((conv1_x3_kernel<2, 2, true>), grid, dim3(256), 0, stream, a);
    else if (C1 == 64) hipLaunchKernelGGL((conv1_x3_kernel<4, 0, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1_x3_kernel<2, 0, true>), grid, dim3(256), 0, stream, a);
  } else {
    if (C2) hipLaunchKernelGGL((conv1_x3_kernel<2, 2, false>), grid, dim3(256), 0, stream, a);
    else if (C1 == 64) hipLaunchKernelGGL((conv1_x3_kernel<4, 0, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1_x3_kernel<2, 0, false>), grid, dim3(256), 0, stream, a);
  }
  return dca_launch_status();
}

}  // namespace

extern "C" int dca_conv1_x3_forward(const float* x, const float* x2, const void* wfrag, float* y, const float* scale,
                                    const float* shift, const float* res_pre, const float* res_post, float slope, int N,
                                    int C1, int C2, int Cout, int CoutTotal, int co_off, long S, hipStream_t stream) {
  return c1x_launch(x, x2, wfrag, y, scale, shift, res_pre, res_post, slope, nullptr, N, C1, C2, Cout, CoutTotal, co_off, S,
                    stream);
}

// nchunk of the statistics dca_conv1_x3_forward_stats produces (one partial per workgroup of the launch it will make)
extern "C" long dca_conv1_x3_stats_chunks(int N, long S) {
  if (N <= 0 || S <= 0) return 0;
  return c1x_blocks(N, S);
}

// y = conv(x [, x2]) (no epilogue) on channels [co_off, co_off + Cout) plus their BatchNorm batch statistics: part
// (CoutTotal * nchunk * 4 doubles, nchunk = dca_conv1_x3_stats_chunks(N, S); every channel slice of a sliced convolution
// fills its own channels) = one {K, n, sum (y - K), sum (y - K)^2} per (channel, workgroup), for dca_bn_finalize_centered
extern "C" int dca_conv1_x3_forward_stats(const float* x, const float* x2, const void* wfrag, float* y, double* stat_part,
                                          int N, int C1, int C2, int Cout, int CoutTotal, int co_off, long S,
                                          hipStream_t stream) {
  DCA_REQUIRE(stat_part);
  return c1x_launch(x, x2, wfrag, y, nullptr, nullptr, nullptr, nullptr, 1.f, stat_part, N, C1, C2, Cout, CoutTotal,
                    co_off, S, stream);
}
