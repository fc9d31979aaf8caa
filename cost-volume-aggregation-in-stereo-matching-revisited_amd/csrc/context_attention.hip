// DCA module: homogeneous-region context injection and per-pixel disparity attention (gfx950).
//
// Replaces (reference):
//  * SemanticLevelContext.forward, models/augment/semantic_level.py:96-126 -- the python loop over
//    (batch, disparity class) with boolean-mask gathers and one host sync per class.  Closed form
//    (SURVEY Appendix B.3): p = softmax_k(preds), k* = argmax_k p, w = softmax of p[k*] over the
//    pixels that share (b, k*), key = x * (1 + [k == k*] * w).  Two launches, no host sync.
//  * SelfAttentionBlock.forward core, models/augment/SelfAttention_bn.py:70-94 -- 4 heads of 8
//    channels, sequence = the n = D/8 disparity bins of one pixel, softmax(q k^T / sqrt 8) v.
//    The n x n score matrix never leaves registers (online softmax); K/V tiles of 32 pixels are
//    staged in LDS with the pixel index on the lanes (coalesced along W, conflict free).
#include "dca_common.h"
#include "../../include/dca_hip.h"

// ---------------------------------------------------------------------------------------------
// context injection, forward
// ---------------------------------------------------------------------------------------------
// Deterministic per-class sums: every wave reduces (kstar == k ? val : 0) with a fixed shuffle tree, the 4 wave
// results are added in order, each workgroup writes its partial row part[(b*nblk + blk)*n + k] and a second kernel
// adds the rows in block order.  (Float atomics here made the whole network non-reproducible at the 1e-7 level,
// which can flip an arg-max further down -- a discontinuity of the reference's algorithm itself.)
__device__ __forceinline__ void class_partials(float val, int kb, bool valid, int n, float* red /*[4][n]*/,
                                               float* __restrict__ part_row) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int k = 0; k < n; ++k) {
    const float s = wave_sum((valid && kb == k) ? val : 0.f);
    if (lane == 0) red[wv * n + k] = s;
  }
  __syncthreads();
  for (int k = tid; k < n; k += 256) part_row[k] = (red[k] + red[n + k]) + (red[2 * n + k] + red[3 * n + k]);
}

__global__ void class_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int n, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int b = i / n, k = i % n;
  float s = 0.f;
  for (int j = 0; j < nblk; ++j) s += part[((long)b * nblk + j) * n + k];
  out[i] = s;
}

// per pixel: k*, e = exp(p[k*] - 1), pm = p[k*]; part[b][blk][k] = sum of e over this block's pixels with k* = k
__global__ __launch_bounds__(256) void ctx_stats_kernel(const float* __restrict__ preds, int* __restrict__ kstar,
                                                        float* __restrict__ e_out, float* __restrict__ pm_out,
                                                        float* __restrict__ part, int n, long HW) {
  extern __shared__ float red[];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long pix = (long)blockIdx.x * 256 + tid;
  float e = 0.f;
  int kb = -1;
  if (pix < HW) {
    const float* lp = preds + (long)b * n * HW + pix;
    float m = -INFINITY;
    for (int k = 0; k < n; ++k) m = fmaxf(m, lp[k * HW]);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(lp[k * HW] - m);
    float best = -1.f;
    kb = 0;
    for (int k = 0; k < n; ++k) {
      const float p = expf(lp[k * HW] - m) / s;
      if (p > best) { best = p; kb = k; }   // first maximum on ties (argmax)
    }
    e = expf(best - 1.0f);
    kstar[(long)b * HW + pix] = kb;
    e_out[(long)b * HW + pix] = e;
    pm_out[(long)b * HW + pix] = best;
  }
  class_partials(e, kb, pix < HW, n, red, part + ((long)b * gridDim.x + blockIdx.x) * n);
}

// out[b,c,k,pix] = in[b,c,k,pix] * (1 + [k == k*] e/denom[b,k*])   (forward: in = x; backward: in = dkey)
__global__ void ctx_scale_kernel(const float* __restrict__ in, const int* __restrict__ kstar,
                                 const float* __restrict__ e, const float* __restrict__ denom,
                                 float* __restrict__ out, int C, int n, long HW, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx % HW;
    long t = idx / HW;
    const int k = t % n; t /= n;
    const long b = t / C;
    const int ks = kstar[b * HW + pix];
    float sc = 1.f;
    if (k == ks) sc += e[b * HW + pix] / denom[b * n + ks];
    out[idx] = in[idx] * sc;
  }
}

// the same with four pixels per thread (HW % 4 == 0, 16-byte aligned tensors): grid (pixel quads, C * n, B), 32-bit index
// arithmetic, 16-byte loads and stores -- the flat form above spends its time in 64-bit div / mod chains (97 us for a 200 MB
// stream at the batch-4 shape)
__global__ __launch_bounds__(256) void ctx_scale4_kernel(const float* __restrict__ in, const int* __restrict__ kstar,
                                                         const float* __restrict__ e, const float* __restrict__ denom,
                                                         float* __restrict__ out, int n, int HWq) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= HWq) return;
  const int ck = blockIdx.y, k = ck % n, b = blockIdx.z;
  const long row = ((long)b * gridDim.y + ck) * HWq + q;          // in float4 units
  const int4 ks = ((const int4*)kstar)[(long)b * HWq + q];
  float4 v = ((const float4*)in)[row];
  if (ks.x == k || ks.y == k || ks.z == k || ks.w == k) {
    const float4 ev = ((const float4*)e)[(long)b * HWq + q];
    const float* dn = denom + (long)b * n;
    if (ks.x == k) v.x *= 1.f + ev.x / dn[k];
    if (ks.y == k) v.y *= 1.f + ev.y / dn[k];
    if (ks.z == k) v.z *= 1.f + ev.z / dn[k];
    if (ks.w == k) v.w *= 1.f + ev.w / dn[k];
  }
  ((float4*)out)[row] = v;
}

// ---------------------------------------------------------------------------------------------
// context injection, backward
// ---------------------------------------------------------------------------------------------
// dw[pix] = sum_c dkey[c,k*,pix] x[c,k*,pix];  part[b][blk][k] = sum of w*dw over this block's pixels of class k
__global__ __launch_bounds__(256) void ctx_bwd_reduce_kernel(const float* __restrict__ dkey,
                                                             const float* __restrict__ x,
                                                             const int* __restrict__ kstar,
                                                             const float* __restrict__ e,
                                                             const float* __restrict__ denom,
                                                             float* __restrict__ dw, float* __restrict__ part, int C,
                                                             int n, long HW) {
  extern __shared__ float red[];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long pix = (long)blockIdx.x * 256 + tid;
  float val = 0.f;
  int ks = -1;
  if (pix < HW) {
    ks = kstar[(long)b * HW + pix];
    const long base = ((long)b * C * n + ks) * HW + pix;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dkey[base + (long)c * n * HW] * x[base + (long)c * n * HW];
    dw[(long)b * HW + pix] = s;
    val = e[(long)b * HW + pix] / denom[b * n + ks] * s;
  }
  class_partials(val, ks, pix < HW, n, red, part + ((long)b * gridDim.x + blockIdx.x) * n);
}

// dpreds[b,k,pix] = dm * pm * ([k==k*] - p_k),  dm = w (dw - T[b,k*])
__global__ void ctx_bwd_preds_kernel(const float* __restrict__ preds, const int* __restrict__ kstar,
                                     const float* __restrict__ e, const float* __restrict__ pm,
                                     const float* __restrict__ denom, const float* __restrict__ dw,
                                     const float* __restrict__ T, float* __restrict__ dpreds, int n, long HW,
                                     long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, pix = idx % HW;
    const int ks = kstar[idx];
    const float w = e[idx] / denom[b * n + ks];
    const float dm = w * (dw[idx] - T[b * n + ks]);
    const float c0 = dm * pm[idx];
    const float* lp = preds + b * n * HW + pix;
    float* gp = dpreds + b * n * HW + pix;
    float m = -INFINITY;
    for (int k = 0; k < n; ++k) m = fmaxf(m, lp[k * HW]);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(lp[k * HW] - m);
    const float inv = 1.f / s;
    for (int k = 0; k < n; ++k) {
      const float p = expf(lp[k * HW] - m) * inv;
      gp[k * HW] = c0 * ((k == ks ? 1.f : 0.f) - p);
    }
  }
}

// Staging of NT tensors' (8 channels x n bins x PW pixels) head tiles into LDS: hardware-predicated 16-byte buffer loads, six per
// tensor and thread in flight at once (HW % 4 == 0), instead of a loop of conditional 4-byte loads that waits for memory once
// per iteration -- with ONE workgroup per CU (the backward kernel's 107 KB of LDS) that loop was 24 round trips = most of the
// kernel (attn_bwd 800 us per launch, tools/attn_time.py).  scale multiplies tensor `sidx` (the pre-scaled q tile).
template <int NT, int PW>
__device__ __forceinline__ void attn_stage(const float* const (&src)[NT], float* const (&dst)[NT], long hb, int n, long HW,
                                           long pix0, int tid, int sidx, float scale) {
  constexpr int NTH = 8 * PW, PQ = PW / 4, RQ = 6;
  const int nq = 2 * n * PW;                       // quads per tensor
  if ((HW & 3) == 0 && 8L * n * HW * 4 < 0x7ffffff0L) {
    __amdgpu_buffer_rsrc_t r[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = dca_rsrc(src[t] + hb, 8L * n * HW * 4);
    for (int i0 = 0; i0 < nq; i0 += NTH * RQ) {
      float4 v[NT][RQ];
#pragma unroll
      for (int k = 0; k < RQ; ++k) {
        const int it = i0 + tid + NTH * k, p4 = it & (PQ - 1), cj = it / PQ;
        const int ok = (int)(it < nq) & (int)(pix0 + 4 * p4 < HW);
        const int off = (int)(((long)cj * HW + pix0 + 4 * p4) * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t][k] = dca_bload4(r[t], off, ok);
      }
#pragma unroll
      for (int k = 0; k < RQ; ++k) {
        const int it = i0 + tid + NTH * k;
        if (it < nq) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float4 w = v[t][k];
            if (t == sidx) { w.x *= scale; w.y *= scale; w.z *= scale; w.w *= scale; }
            *(float4*)(dst[t] + 4 * it) = w;
          }
        }
      }
    }
  } else {
    for (int it = tid; it < 8 * n * PW; it += NTH) {
      const int p = it & (PW - 1), cj = it / PW;
      const bool ok = pix0 + p < HW;
      const long g = hb + (long)cj * HW + pix0 + p;
#pragma unroll
      for (int t = 0; t < NT; ++t) dst[t][it] = ok ? src[t][g] * (t == sidx ? scale : 1.f) : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// disparity attention, forward.  grid (ceil(HW/PW), heads, B); thread = (pixel lane 0..PW-1, query group 0..7);
// a thread owns the queries qg + 8t, t < QPT = ceil(n/8) <= 8, i.e. n <= 64 disparity bins.  PW = 32 pixels per
// workgroup; the backward kernel drops to PW = 16 for n > 32 so that its five staged tiles still fit the 160 KB of LDS.
// ---------------------------------------------------------------------------------------------
template <int QPT, int PW>
__global__ __launch_bounds__(8 * PW) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out, int C,
                                                       int n, long HW) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ks = smem;               // [8][n][32]
  float* vs = smem + 8 * n * PW;  // [8][n][PW]
  const int tid = threadIdx.x, pl = tid & (PW - 1), qg = tid / PW;
  const int head = blockIdx.y, b = blockIdx.z;
  const long pix0 = (long)blockIdx.x * PW, pix = pix0 + pl;
  const long hb = ((long)b * C + head * 8) * n * HW;  // offset of (b, head*8, 0, 0)
  {
    const float* const src[2] = {k, v};
    float* const dst[2] = {ks, vs};
    attn_stage<2, PW>(src, dst, hb, n, HW, pix0, tid, -1, 1.f);
  }
  __syncthreads();
  if (pix >= HW) return;
  float qv[QPT][8], ctx[QPT][8], m[QPT], l[QPT];
#pragma unroll
  for (int t = 0; t < QPT; ++t) {
    const int i = qg + 8 * t;
    m[t] = -INFINITY;
    l[t] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      qv[t][c] = (i < n) ? q[hb + ((long)c * n + i) * HW + pix] * 0.35355339059327373f : 0.f;
      ctx[t][c] = 0.f;
    }
  }
#pragma unroll 4
  for (int j = 0; j < n; ++j) {
    float kj[8], vj[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      kj[c] = ks[(c * n + j) * PW + pl];
      vj[c] = vs[(c * n + j) * PW + pl];
    }
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) s += qv[t][c] * kj[c];
      const float mn = fmaxf(m[t], s);
      const float corr = expf(m[t] - mn), p = expf(s - mn);
      l[t] = l[t] * corr + p;
#pragma unroll
      for (int c = 0; c < 8; ++c) ctx[t][c] = ctx[t][c] * corr + p * vj[c];
      m[t] = mn;
    }
  }
#pragma unroll
  for (int t = 0; t < QPT; ++t) {
    const int i = qg + 8 * t;
    if (i >= n) continue;
    const float inv = 1.f / l[t];
#pragma unroll
    for (int c = 0; c < 8; ++c) out[hb + ((long)c * n + i) * HW + pix] = ctx[t][c] * inv;
  }
}

// backward, two phases inside one workgroup (32 pixels x one head), no atomics:
//  phase 1, thread = (pixel, queries i = qg + 8t, t < QPT): softmax statistics (m, 1/l), D_i = sum_j p_ij dP_ij and
//           dq_i; (m, 1/l, D) go to LDS next to the staged q / dout tiles.
//  phase 2, thread = (pixel, keys j = qg + 8t):  dk_j = sum_i dS_ij q_i / sqrt(8), dv_j = sum_i p_ij dout_i in registers.
// A thread's QPT queries (keys) share every k/v (q/dout) value it reads from LDS, which divides the LDS traffic by QPT
// and gives QPT independent dependency chains (the workgroup's 107 KB of LDS allow only one wave per SIMD).
template <int QPT, int PW>
__global__ __launch_bounds__(8 * PW) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ dout,
                                                       float* __restrict__ dq, float* __restrict__ dk,
                                                       float* __restrict__ dv, int C, int n, long HW) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile = 8 * n * PW;
  float* ks = smem;              // [8][n][32]
  float* vs = smem + tile;
  float* qs = smem + 2 * tile;   // pre-scaled by 1/sqrt(8)
  float* gs = smem + 3 * tile;   // dout
  float* st = smem + 4 * tile;   // [3][n][32]: m, 1/l, D
  const int tid = threadIdx.x, pl = tid & (PW - 1), qg = tid / PW;
  const int head = blockIdx.y, b = blockIdx.z;
  const long pix0 = (long)blockIdx.x * PW, pix = pix0 + pl;
  const long hb = ((long)b * C + head * 8) * n * HW;
  const float scale = 0.35355339059327373f;
  {
    const float* const src[4] = {k, v, q, dout};
    float* const dst[4] = {ks, vs, qs, gs};
    attn_stage<4, PW>(src, dst, hb, n, HW, pix0, tid, 2, scale);
  }
  __syncthreads();
  {
    float qv[QPT][8], go[QPT][8], dqv[QPT][8], m[QPT], l[QPT], dnum[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      const int i = min(qg + 8 * t, n - 1);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        qv[t][c] = qs[(c * n + i) * PW + pl];
        go[t][c] = gs[(c * n + i) * PW + pl];
        dqv[t][c] = 0.f;
      }
      m[t] = -INFINITY; l[t] = 0.f; dnum[t] = 0.f;
    }
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
      float kj[8], vj[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        kj[c] = ks[(c * n + j) * PW + pl];
        vj[c] = vs[(c * n + j) * PW + pl];
      }
#pragma unroll
      for (int t = 0; t < QPT; ++t) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          s += qv[t][c] * kj[c];
          dp += go[t][c] * vj[c];
        }
        const float mn = fmaxf(m[t], s);
        const float corr = expf(m[t] - mn), p = expf(s - mn);
        l[t] = l[t] * corr + p;
        dnum[t] = dnum[t] * corr + p * dp;
        m[t] = mn;
      }
    }
    float inv[QPT], Dsum[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      const int i = qg + 8 * t;
      inv[t] = 1.f / l[t];
      Dsum[t] = dnum[t] * inv[t];
      if (i < n) {
        st[(0 * n + i) * PW + pl] = m[t];
        st[(1 * n + i) * PW + pl] = inv[t];
        st[(2 * n + i) * PW + pl] = Dsum[t];
      }
    }
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
      float kj[8], vj[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        kj[c] = ks[(c * n + j) * PW + pl];
        vj[c] = vs[(c * n + j) * PW + pl];
      }
#pragma unroll
      for (int t = 0; t < QPT; ++t) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          s += qv[t][c] * kj[c];
          dp += go[t][c] * vj[c];
        }
        const float ds = expf(s - m[t]) * inv[t] * (dp - Dsum[t]);
#pragma unroll
        for (int c = 0; c < 8; ++c) dqv[t][c] += ds * kj[c];
      }
    }
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      const int i = qg + 8 * t;
      if (i < n && pix < HW) {
#pragma unroll
        for (int c = 0; c < 8; ++c) dq[hb + ((long)c * n + i) * HW + pix] = dqv[t][c] * scale;
      }
    }
  }
  __syncthreads();
  {
    float kj[QPT][8], vj[QPT][8], dkj[QPT][8], dvj[QPT][8];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      const int j = min(qg + 8 * t, n - 1);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        kj[t][c] = ks[(c * n + j) * PW + pl];
        vj[t][c] = vs[(c * n + j) * PW + pl];
        dkj[t][c] = 0.f;
        dvj[t][c] = 0.f;
      }
    }
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      float qv[8], go[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        qv[c] = qs[(c * n + i) * PW + pl];
        go[c] = gs[(c * n + i) * PW + pl];
      }
      const float mi = st[(0 * n + i) * PW + pl], li = st[(1 * n + i) * PW + pl], Di = st[(2 * n + i) * PW + pl];
#pragma unroll
      for (int t = 0; t < QPT; ++t) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          s += qv[c] * kj[t][c];
          dp += go[c] * vj[t][c];
        }
        const float p = expf(s - mi) * li;
        const float ds = p * (dp - Di);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          dkj[t][c] += ds * qv[c];   // qv already carries 1/sqrt(8)
          dvj[t][c] += p * go[c];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      const int j = qg + 8 * t;
      if (j < n && pix < HW) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          dk[hb + ((long)c * n + j) * HW + pix] = dkj[t][c];
          dv[hb + ((long)c * n + j) * HW + pix] = dvj[t][c];
        }
      }
    }
  }
}

static int ew_grid(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192);
}

static void ctx_scale_launch(const float* in, const int* kstar, const float* e, const float* denom, float* out, int B, int C,
                             int n, long HW, hipStream_t stream) {
  const uintptr_t al = (uintptr_t)in | (uintptr_t)kstar | (uintptr_t)e | (uintptr_t)out;
  if (HW % 4 == 0 && (al & 15) == 0 && (long)C * n <= 65535 && HW / 4 < 0x7fffffffL) {
    const int HWq = (int)(HW / 4);
    hipLaunchKernelGGL(ctx_scale4_kernel, dim3(cdiv(HWq, 256), C * n, B), dim3(256), 0, stream, in, kstar, e, denom, out, n, HWq);
    return;
  }
  const long total = (long)B * C * n * HW;
  hipLaunchKernelGGL(ctx_scale_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, in, kstar, e, denom, out, C, n, HW, total);
}

// x, key: (B,C,n,H,W); preds: (B,n,H,W); outputs kstar (B,HW) int32, e/pm (B,HW), denom (B,n).
// part: scratch of B * ceil(HW/256) * n floats.
extern "C" int dca_context_inject_fwd(const float* x, const float* preds, float* key, int* kstar, float* e, float* pm,
                                      float* denom, float* part, int B, int C, int n, long HW, hipStream_t stream) {
  DCA_REQUIRE(x && preds && key && kstar && e && pm && denom && part && B > 0 && C > 0 && n > 0 && HW > 0);
  DCA_REQUIRE(B <= 65535);
  const int nblk = cdiv(HW, 256);
  hipLaunchKernelGGL(ctx_stats_kernel, dim3(nblk, B), dim3(256), 4 * n * sizeof(float), stream, preds, kstar, e, pm,
                     part, n, HW);
  hipLaunchKernelGGL(class_sum_kernel, dim3(cdiv((long)B * n, 64)), dim3(64), 0, stream, part, denom, nblk, n, B);
  ctx_scale_launch(x, kstar, e, denom, key, B, C, n, HW, stream);
  return dca_launch_status();
}

// dkey -> dx (B,C,n,HW) and dpreds (B,n,HW); scratch dw (B,HW), T (B,n), part (B * ceil(HW/256) * n)
extern "C" int dca_context_inject_bwd(const float* dkey, const float* x, const float* preds, const int* kstar,
                                      const float* e, const float* pm, const float* denom, float* dx, float* dpreds,
                                      float* dw, float* T, float* part, int B, int C, int n, long HW,
                                      hipStream_t stream) {
  DCA_REQUIRE(dkey && x && preds && kstar && e && pm && denom && dx && dpreds && dw && T && part);
  DCA_REQUIRE(B > 0 && C > 0 && n > 0 && HW > 0 && B <= 65535);
  const int nblk = cdiv(HW, 256);
  hipLaunchKernelGGL(ctx_bwd_reduce_kernel, dim3(nblk, B), dim3(256), 4 * n * sizeof(float), stream, dkey, x, kstar, e,
                     denom, dw, part, C, n, HW);
  hipLaunchKernelGGL(class_sum_kernel, dim3(cdiv((long)B * n, 64)), dim3(64), 0, stream, part, T, nblk, n, B);
  hipLaunchKernelGGL(ctx_bwd_preds_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, preds, kstar, e, pm,
                     denom, dw, T, dpreds, n, HW, (long)B * HW);
  ctx_scale_launch(dkey, kstar, e, denom, dx, B, C, n, HW, stream);
  return dca_launch_status();
}

extern "C" int dca_disp_attention_fwd(const float* q, const float* k, const float* v, float* out, int B, int C, int n,
                                      long HW, hipStream_t stream) {
  DCA_REQUIRE(q && k && v && out && B > 0 && C > 0 && C % 8 == 0 && n > 0 && n <= 64 && HW > 0);
  DCA_REQUIRE(C / 8 <= 65535 && B <= 65535);
  const size_t lds = (size_t)2 * 8 * n * 32 * 4;
  const dim3 grid(cdiv(HW, 32), C / 8, B);
  const int qpt = (n + 7) / 8;
#define LAUNCH(Q)                                                                                                      \
  do {                                                                                                                 \
    if (lds > 64 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<Q, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds);                                                                    \
      if (e != hipSuccess) return (int)e;                                                                              \
    }                                                                                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<Q, 32>), grid, dim3(256), lds, stream, q, k, v, out, C, n, HW);                \
  } while (0)
  switch (qpt) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    default: LAUNCH(8); break;
  }
#undef LAUNCH
  return dca_launch_status();
}

extern "C" int dca_disp_attention_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq,
                                      float* dk, float* dv, int B, int C, int n, long HW, hipStream_t stream) {
  DCA_REQUIRE(q && k && v && dout && dq && dk && dv && B > 0 && C > 0 && C % 8 == 0 && n > 0 && n <= 64 && HW > 0);
  DCA_REQUIRE(C / 8 <= 65535 && B <= 65535);
  const int qpt = (n + 7) / 8;
#define LAUNCH(Q, PWV)                                                                                                 \
  do {                                                                                                                 \
    const size_t lds = (size_t)(4 * 8 + 3) * n * PWV * 4;                                                              \
    const dim3 grid(cdiv(HW, PWV), C / 8, B);                                                                          \
    if (lds > 64 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel<Q, PWV>,                                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return (int)e;                                                                              \
    }                                                                                                                  \
    hipLaunchKernelGGL((attn_bwd_kernel<Q, PWV>), grid, dim3(8 * PWV), lds, stream, q, k, v, dout, dq, dk, dv, C, n, HW); \
  } while (0)
  switch (qpt) {
    case 1: LAUNCH(1, 32); break;
    case 2: LAUNCH(2, 32); break;
    case 3: LAUNCH(3, 32); break;
    case 4: LAUNCH(4, 32); break;
    case 5: LAUNCH(5, 16); break;     // n > 32: 16 pixels per workgroup keep the five staged tiles inside 160 KB of LDS
    case 6: LAUNCH(6, 16); break;
    case 7: LAUNCH(7, 16); break;
    default: LAUNCH(8, 16); break;
  }
#undef LAUNCH
  return dca_launch_status();
}
