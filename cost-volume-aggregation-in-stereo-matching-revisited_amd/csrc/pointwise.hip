// Bandwidth-bound glue of the DCANet aggregation path: BatchNorm3d (batch statistics, affine +
// activation + residual apply, backward), AvgPool3d(3,2,1) and trilinear up-sampling.
//
// Replaces (reference): nn.BatchNorm3d inside convbn_3d (models/submodule.py:121-124) with the
// ReLU / LeakyReLU(0.1) / residual adds that follow it (models/gwcnet_dca_g.py:141-148,225,229;
// models/augment/cva.py:26-31,55; SelfAttention_bn.py:139-143), nn.AvgPool3d (cva.py:39) and
// F.interpolate(mode='trilinear', align_corners=False) (cva.py:64, gwcnet_dca_g.py:251-261).
//
// All tensors are NC[D]HW fp32: channel c owns contiguous runs of S = D*H*W floats per sample, so
// every kernel streams 16 B per lane along S.
#include "dca_common.h"
#include "../../include/dca_hip.h"

// ------------------------------------------------------------------------------------ BN statistics
// part[(c*nchunk + chunk)*2 + {0,1}] = (sum (x-K), sum (x-K)^2) over all samples and one chunk of S (double), with the
// per-channel shift K = x[0, c, 0] stored at part[C*nchunk*2 + c]: E[(x-K)^2] - E[x-K]^2 does not cancel
// catastrophically when |mean| >> std (the raw E[x^2] - mean^2 loses (mean/std)^2 * 1e-7 of the variance in fp32).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ part, int N,
                                                       int C, long S, int nchunk, long chunk_len, int vec) {
  const int c = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  const float K = x[(long)c * S];
  if (ch == 0 && tid == 0) part[(long)C * nchunk * 2 + c] = (double)K;
  float s = 0.f, ss = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* p = x + ((long)n * C + c) * S;
    if (vec) {
      for (long i = s0 + 4 * tid; i < s1; i += 1024) {
        float4 v = *(const float4*)(p + i);
        v.x -= K; v.y -= K; v.z -= K; v.w -= K;
        s += (v.x + v.y) + (v.z + v.w);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    } else {
      for (long i = s0 + tid; i < s1; i += 256) {
        const float v = p[i] - K;
        s += v;
        ss += v * v;
      }
    }
  }
  double ds = wave_sum_d((double)s), dss = wave_sum_d((double)ss);
  __shared__ double red[8];
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = ds; red[(tid >> 6) * 2 + 1] = dss; }
  __syncthreads();
  if (tid == 0) {
    part[((long)c * nchunk + ch) * 2 + 0] = red[0] + red[2] + red[4] + red[6];
    part[((long)c * nchunk + ch) * 2 + 1] = red[1] + red[3] + red[5] + red[7];
  }
}

// Scale exponent (dca_common.h) of channel c of z = act(gamma * xhat + beta), |slope| <= 1, for batch statistics over `count`
// elements: no sample lies more than sqrt(count - 1) standard deviations from the batch mean, so |z| <= |gamma| sqrt(count) +
// |beta| whatever the data -- known BEFORE z is written, which is what lets bn_apply_pack_kernel write the packed operand
// format in one pass.  The bound is loose (a Gaussian's maximum over 6e6 samples is ~5.3 sigma, the bound 2500): typical
// values land ~2^9 below the [2^14, 2^15) target, inside the 18 binades over which the two f16 terms keep all 22 bits.
// With residuals z = act(BN(y) + res_pre) + res_post the bound grows by the residual channels' maxima (their producers'
// per-channel slots, dca_common.h).  Wave-wide call (the slots are reduced over the lanes); the result is valid in lane 0.
__device__ __forceinline__ int bn_bound_exp(float gamma, float beta, double count, int c, const unsigned* rp_slots, int rp_n,
                                            const unsigned* rq_slots, int rq_n) {
  unsigned a = 0, b2 = 0;
  const int lane = threadIdx.x & 63;
  if (rp_slots) for (int i = lane; i < rp_n; i += 64) { const unsigned u = rp_slots[(long)c * DCA_AMAX_CSLOTS + i]; a = a > u ? a : u; }
  if (rq_slots) for (int i = lane; i < rq_n; i += 64) { const unsigned u = rq_slots[(long)c * DCA_AMAX_CSLOTS + i]; b2 = b2 > u ? b2 : u; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned u = (unsigned)__shfl_xor((int)a, o, 64), w = (unsigned)__shfl_xor((int)b2, o, 64);
    a = a > u ? a : u;
    b2 = b2 > w ? b2 : w;
  }
  const float b = (fabsf(gamma) * sqrtf((float)count) + fabsf(beta) + __uint_as_float(a) + __uint_as_float(b2)) * 1.0001f;
  return x2_scale_exp(__float_as_uint(b));
}

// stats[0..C) mean, [C..2C) invstd, [2C..3C) scale = gamma*invstd, [3C..4C) shift = beta - mean*scale.
// training: batch statistics (+ running-stat update, momentum m, unbiased variance); otherwise running stats.
__global__ __launch_bounds__(64) void bn_finalize_kernel(const double* __restrict__ part, int nchunk, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                   float eps, int training, float* __restrict__ stats, int* __restrict__ zexps,
                                   const unsigned* __restrict__ rp_slots, int rp_n, const unsigned* __restrict__ rq_slots,
                                   int rq_n, int C) {
  const int c = blockIdx.x, lane = threadIdx.x;   // one wave per channel; lanes stride over the chunk partials
  float mean, var;
  if (training) {
    double s = 0.0, ss = 0.0;
    for (int i = lane; i < nchunk; i += 64) {
      s += part[((long)c * nchunk + i) * 2 + 0];
      ss += part[((long)c * nchunk + i) * 2 + 1];
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    const double ms = s / count;                       // mean of (x - K)
    double v = ss / count - ms * ms;
    if (v < 0.0) v = 0.0;
    const double m = part[(long)C * nchunk * 2 + c] + ms;
    mean = (float)m;
    var = (float)v;
    if (running_mean && lane == 0) {
      const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const int ze = zexps ? bn_bound_exp(g, b, count, c, rp_slots, rp_n, rq_slots, rq_n) : 0;
  if (lane != 0) return;
  const float invstd = 1.0f / sqrtf(var + eps);
  stats[c] = mean;
  stats[C + c] = invstd;
  stats[2 * C + c] = g * invstd;
  stats[3 * C + c] = b - mean * g * invstd;
  if (zexps) zexps[c] = ze;
}

// Training-mode finalize for the statistics the convolution kernels emit themselves (dca_*_forward_stats): partial i of
// channel c is {K, n, s, q} = (its own shift, element count, sum of (y - K), sum of (y - K)^2), each partial centred on a
// value of its own data (so no partial loses its variance to a large common mean).  All partials are re-centred on the
// first one's K in double: s' = s + n dk, q' = q + 2 dk s + n dk^2 with dk = K_i - K_ref.
__global__ __launch_bounds__(64) void bn_finalize_centered_kernel(const double* __restrict__ part, int nchunk,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                   float eps, float* __restrict__ stats, int* __restrict__ zexps,
                                   const unsigned* __restrict__ rp_slots, int rp_n, const unsigned* __restrict__ rq_slots,
                                   int rq_n, int C) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const double* p = part + (long)c * nchunk * 4;
  const double kref = p[0];
  double n = 0.0, s = 0.0, q = 0.0;
  for (int i = lane; i < nchunk; i += 64) {
    const double ki = p[i * 4 + 0], ni = p[i * 4 + 1], si = p[i * 4 + 2], qi = p[i * 4 + 3];
    const double dk = ki - kref;
    n += ni;
    s += si + ni * dk;
    q += qi + 2.0 * dk * si + ni * dk * dk;
  }
  n = wave_sum_d(n);
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const int ze = zexps ? bn_bound_exp(g, b, n, c, rp_slots, rp_n, rq_slots, rq_n) : 0;
  if (lane != 0) return;
  const double ms = s / n;
  double v = q / n - ms * ms;
  if (v < 0.0) v = 0.0;
  const float mean = (float)(kref + ms), var = (float)v;
  if (running_mean) {
    const double unb = n > 1.0 ? v * n / (n - 1.0) : v;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  stats[c] = mean;
  stats[C + c] = invstd;
  stats[2 * C + c] = g * invstd;
  stats[3 * C + c] = b - mean * g * invstd;
  if (zexps) zexps[c] = ze;
}

// The f16x2 convolution kernels scale every operand CHANNEL by a power of two taken from the channel's max |.|
// (dca_common.h); the kernels that PRODUCE those operands (bn_apply: activations, bn_bwd_apply: gradients) emit the maxima on
// the way: block (chunk ch, channel c) stores its maximum into slot [c][ch] (nchunk <= DCA_AMAX_CSLOTS), or they write the
// operand in the packed px2 format directly (the *_pack kernels), scaled by exponents the finalize kernels derive from bounds.

// z = act(scale[c]*y + shift[c] + res_pre) + res_post.  Block (ch, c): chunk ch of channel c, all samples.
// zmax / ymax (may be null): per-channel slots that receive max |z| and max |y - mean| (the latter bounds |xhat| for the
// gradient's scale in the backward pass).
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                       const float* __restrict__ res_pre, const float* __restrict__ res_post,
                                                       float* __restrict__ z, int N, int C, long S, long chunk_len,
                                                       float slope, int vec, unsigned* __restrict__ zmax,
                                                       unsigned* __restrict__ ymax) {
  const int c = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  const float mean = stats[c], sc = stats[2 * C + c], sh = stats[3 * C + c];
  float am = 0.f, ym = 0.f;
  for (int n = N - 1; n >= 0; --n) {   // descending: see bn_bwd_reduce_kernel
    const long base = ((long)n * C + c) * S;
    if (vec) {
      for (long i = s0 + 4 * tid; i < s1; i += 1024) {
        float4 v = *(const float4*)(y + base + i);
        ym = fmaxf(fmaxf(ym, fmaxf(fabsf(v.x - mean), fabsf(v.y - mean))), fmaxf(fabsf(v.z - mean), fabsf(v.w - mean)));
        v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
        if (res_pre) { const float4 r = *(const float4*)(res_pre + base + i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        v.x = act_apply(v.x, slope); v.y = act_apply(v.y, slope); v.z = act_apply(v.z, slope); v.w = act_apply(v.w, slope);
        if (res_post) { const float4 r = *(const float4*)(res_post + base + i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        *(float4*)(z + base + i) = v;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    } else {
      for (long i = s0 + tid; i < s1; i += 256) {
        const float yv = y[base + i];
        ym = fmaxf(ym, fabsf(yv - mean));
        float v = yv * sc + sh;
        if (res_pre) v += res_pre[base + i];
        v = act_apply(v, slope);
        if (res_post) v += res_post[base + i];
        z[base + i] = v;
        am = fmaxf(am, fabsf(v));
      }
    }
  }
  if (zmax) dca_cmax_put(am, zmax + (long)c * DCA_AMAX_CSLOTS + ch);
  if (ymax) dca_cmax_put(ym, ymax + (long)c * DCA_AMAX_CSLOTS + ch);
}

// x 2^e = h + l (+ <= 2^-22 relative): the two f16 terms of the f16x2 kernels
__device__ __forceinline__ void px2_split(float v, int e, unsigned short& h, unsigned short& l) {
  const float u = ldexpf(v, e);       // exact
  const _Float16 hh = (_Float16)u;
  const _Float16 ll = (_Float16)(u - (float)hh);   // the residual is exact in fp32
  h = __builtin_bit_cast(unsigned short, hh);
  l = __builtin_bit_cast(unsigned short, ll);
}
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// The same BatchNorm apply (no residuals) writing z in the packed px2 operand format (dca_common.h) for the f16x2
// convolution that consumes it: block (ch, cg) = chunk ch of the 8-channel group cg, all samples; a lane takes ONE voxel and
// its 8 channels (eight 4-byte loads, each 256 contiguous bytes per wave; two 16-byte stores, 1 KB contiguous per wave).
// stats == null: identity (plain packing of an fp32 tensor, dca_pack_x2).
// res_pre / res_post / zf (may be null): the residual adds of bn_apply_kernel and a second, fp32, copy of z for readers
// other than the f16x2 convolution.
__global__ __launch_bounds__(256) void bn_apply_pack_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                            const int* __restrict__ zexps, char* __restrict__ zp, int N,
                                                            int C, long S, long chunk_len, float slope,
                                                            unsigned* __restrict__ ymax, const float* __restrict__ res_pre,
                                                            const float* __restrict__ res_post, float* __restrict__ zf,
                                                            unsigned* __restrict__ zmax) {
  const int cg = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  float mean[8], sc[8], sh[8], ym[8], zm[8];
  int ex[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mean[j] = stats ? stats[c] : 0.f;
    sc[j] = stats ? stats[2 * C + c] : 1.f;
    sh[j] = stats ? stats[3 * C + c] : 0.f;
    ex[j] = dca_coherent_loadi(zexps + c);
    ym[j] = zm[j] = 0.f;
  }
  const long tb = px2_term_bytes(C, S);
  for (int n = N - 1; n >= 0; --n) {   // descending: see bn_bwd_reduce_kernel
    const float* yb = y + ((long)n * C + cg * 8) * S;
    char* zb = zp + (long)n * 2 * tb + (long)cg * S * 16;
    const long cb = ((long)n * C + cg * 8) * S;
    for (long i = s0 + tid; i < s1; i += 256) {
      float v[8], rp[8], rq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = yb[j * S + i];
        rp[j] = res_pre ? res_pre[cb + j * S + i] : 0.f;
        rq[j] = res_post ? res_post[cb + j * S + i] : 0.f;
      }
      u16x8 hv, lv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ym[j] = fmaxf(ym[j], fabsf(v[j] - mean[j]));
        const float zz = act_apply(v[j] * sc[j] + sh[j] + rp[j], slope) + rq[j];
        if (zf) zf[cb + j * S + i] = zz;
        zm[j] = fmaxf(zm[j], fabsf(zz));
        unsigned short h, l;
        px2_split(zz, ex[j], h, l);
        hv[j] = h; lv[j] = l;
      }
      *(u16x8*)(zb + i * 16) = hv;
      *(u16x8*)(zb + tb + i * 16) = lv;
    }
  }
  if (ymax) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dca_cmax_put(ym[j], ymax + (long)(cg * 8 + j) * DCA_AMAX_CSLOTS + ch);
  }
  if (zmax) {     // per-channel max |z| for the readers of the fp32 copy (residual bounds, f16x2 operand scales)
#pragma unroll
    for (int j = 0; j < 8; ++j) dca_cmax_put(zm[j], zmax + (long)(cg * 8 + j) * DCA_AMAX_CSLOTS + ch);
  }
}

// The same with FOUR voxels per lane (S % 4 == 0, 16-byte aligned tensors): 16-byte loads per channel, the fp32 copy as
// 16-byte stores, the packed words of the four voxels as 64 contiguous bytes per term; used when residuals are read too
// (dca_bn_apply_pack).
__global__ __launch_bounds__(256) void bn_apply_pack4_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                             const int* __restrict__ zexps, char* __restrict__ zp, int N,
                                                             int C, long S, long chunk_len, float slope,
                                                             unsigned* __restrict__ ymax, const float* __restrict__ res_pre,
                                                             const float* __restrict__ res_post, float* __restrict__ zf,
                                                             unsigned* __restrict__ zmax) {
  const int cg = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  float mean[8], sc[8], sh[8], ym[8], zm[8];
  int ex[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mean[j] = stats ? stats[c] : 0.f;
    sc[j] = stats ? stats[2 * C + c] : 1.f;
    sh[j] = stats ? stats[3 * C + c] : 0.f;
    ex[j] = dca_coherent_loadi(zexps + c);
    ym[j] = zm[j] = 0.f;
  }
  const long tb = px2_term_bytes(C, S);
  for (int n = N - 1; n >= 0; --n) {   // descending: see bn_bwd_reduce_kernel
    const long cb = ((long)n * C + cg * 8) * S;
    char* zb = zp + (long)n * 2 * tb + (long)cg * S * 16;
    for (long i = s0 + 4 * tid; i < s1; i += 1024) {
      u16x8 hv[4], lv[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v4 = *(const float4*)(y + cb + j * S + i);
        float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = p4;
        if (res_pre) p4 = *(const float4*)(res_pre + cb + j * S + i);
        if (res_post) q4 = *(const float4*)(res_post + cb + j * S + i);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w}, rp[4] = {p4.x, p4.y, p4.z, p4.w}, rq[4] = {q4.x, q4.y, q4.z, q4.w};
        float zz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ym[j] = fmaxf(ym[j], fabsf(v[e] - mean[j]));
          zz[e] = act_apply(v[e] * sc[j] + sh[j] + rp[e], slope) + rq[e];
          zm[j] = fmaxf(zm[j], fabsf(zz[e]));
          unsigned short h, l;
          px2_split(zz[e], ex[j], h, l);
          hv[e][j] = h; lv[e][j] = l;
        }
        if (zf) *(float4*)(zf + cb + j * S + i) = make_float4(zz[0], zz[1], zz[2], zz[3]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *(u16x8*)(zb + (i + e) * 16) = hv[e];
        *(u16x8*)(zb + tb + (i + e) * 16) = lv[e];
      }
    }
  }
  if (ymax) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dca_cmax_put(ym[j], ymax + (long)(cg * 8 + j) * DCA_AMAX_CSLOTS + ch);
  }
  if (zmax) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dca_cmax_put(zm[j], zmax + (long)(cg * 8 + j) * DCA_AMAX_CSLOTS + ch);
  }
}

// per-channel max |x| of an fp32 tensor into slots [c][ch] (the read pass for operands whose producer emits nothing)
__global__ __launch_bounds__(256) void cmax_kernel(const float* __restrict__ x, int N, int C, long S, long chunk_len, int vec,
                                                   unsigned* __restrict__ slots) {
  const int c = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  float m = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* p = x + ((long)n * C + c) * S;
    if (vec) {
      for (long i = s0 + 4 * tid; i < s1; i += 1024) {
        const float4 v = *(const float4*)(p + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    } else {
      for (long i = s0 + tid; i < s1; i += 256) m = fmaxf(m, fabsf(p[i]));
    }
  }
  dca_cmax_put(m, slots + (long)c * DCA_AMAX_CSLOTS + ch);
}

// exps[c] = exponent that brings max over the channel's nslots slot maxima into [2^14, 2^15); one wave per channel
__global__ __launch_bounds__(64) void cmax_exps_kernel(const unsigned* __restrict__ slots, int nslots, int* __restrict__ exps) {
  const int c = blockIdx.x, lane = threadIdx.x;
  unsigned v = 0;
  for (int i = lane; i < nslots; i += 64) { const unsigned u = slots[(long)c * DCA_AMAX_CSLOTS + i]; v = v > u ? v : u; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned u = (unsigned)__shfl_xor((int)v, o, 64); v = v > u ? v : u; }
  if (lane == 0) exps[c] = x2_scale_exp(v);
}

// ------------------------------------------------------------------------------------ BN backward
// u = scale*y + shift (+res_pre); g = dz * (u > 0 ? 1 : slope); xhat = (y - mean)*invstd
// part = per-(channel, chunk) double sums of (g, g*xhat); gmax (may be null): per-channel slots [c][ch] with max |g|.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                            const float* __restrict__ res_pre,
                                                            const float* __restrict__ stats,
                                                            double* __restrict__ part, int N, int C, long S,
                                                            int nchunk, long chunk_len, float slope, int vec,
                                                            unsigned* __restrict__ gmax) {
  const int c = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  const float mean = stats[c], invstd = stats[C + c], sc = stats[2 * C + c], sh = stats[3 * C + c];
  float a0 = 0.f, a1 = 0.f, gm = 0.f;
  for (int n = N - 1; n >= 0; --n) {   // the producer of dz wrote the last sample last: its tail is still in the infinity cache
    const long base = ((long)n * C + c) * S;
    if (vec) {
      for (long i = s0 + 4 * tid; i < s1; i += 1024) {
        const float4 yv = *(const float4*)(y + base + i);
        const float4 dv = *(const float4*)(dz + base + i);
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res_pre) rv = *(const float4*)(res_pre + base + i);
        const float ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = ys[j] * sc + sh + rs[j];
          const float g = ds[j] * (u > 0.f ? 1.f : slope);
          a0 += g;
          a1 += g * (ys[j] - mean) * invstd;
          gm = fmaxf(gm, fabsf(g));
        }
      }
    } else {
      for (long i = s0 + tid; i < s1; i += 256) {
        const float yv = y[base + i];
        float u = yv * sc + sh;
        if (res_pre) u += res_pre[base + i];
        const float g = dz[base + i] * (u > 0.f ? 1.f : slope);
        a0 += g;
        a1 += g * (yv - mean) * invstd;
        gm = fmaxf(gm, fabsf(g));
      }
    }
  }
  double d0 = wave_sum_d((double)a0), d1 = wave_sum_d((double)a1);
  __shared__ double red[8];
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = d0; red[(tid >> 6) * 2 + 1] = d1; }
  __syncthreads();
  if (tid == 0) {
    part[((long)c * nchunk + ch) * 2 + 0] = red[0] + red[2] + red[4] + red[6];
    part[((long)c * nchunk + ch) * 2 + 1] = red[1] + red[3] + red[5] + red[7];
  }
  if (gmax) dca_cmax_put(gm, gmax + (long)c * DCA_AMAX_CSLOTS + ch);
}

// dgb[0..C) = dgamma, dgb[C..2C) = dbeta, dgb[2C..3C) = dbeta/count, dgb[3C..4C) = dgamma/count.
// dyexps (may be null): the scale exponent of channel c of dy = scale (g - dbeta/count - xhat dgamma/count) from the bound
// |dy| <= |scale| (max|g| + |dbeta/count| + max|xhat| |dgamma/count|), max |g| from gmax (this launch's reduce pass),
// max |xhat| = invstd * max|y - mean| from ymax (the forward apply pass; null: sqrt(count), the largest |xhat| a sample can have).
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nchunk, double count,
                                       float* __restrict__ dgb, int C, const float* __restrict__ stats, int training,
                                       const unsigned* __restrict__ gmax, const unsigned* __restrict__ ymax, int ymax_slots,
                                       int* __restrict__ dyexps) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int i = lane; i < nchunk; i += 64) {
    s0 += part[((long)c * nchunk + i) * 2 + 0];
    s1 += part[((long)c * nchunk + i) * 2 + 1];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  unsigned gmb = 0, ymb = 0;
  if (dyexps) {
    for (int i = lane; i < nchunk; i += 64) { const unsigned u = gmax[(long)c * DCA_AMAX_CSLOTS + i]; gmb = gmb > u ? gmb : u; }
    if (ymax)
      for (int i = lane; i < ymax_slots; i += 64) { const unsigned u = ymax[(long)c * DCA_AMAX_CSLOTS + i]; ymb = ymb > u ? ymb : u; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned u = (unsigned)__shfl_xor((int)gmb, o, 64), w = (unsigned)__shfl_xor((int)ymb, o, 64);
      gmb = gmb > u ? gmb : u;
      ymb = ymb > w ? ymb : w;
    }
  }
  if (lane != 0) return;
  dgb[c] = (float)s1;
  dgb[C + c] = (float)s0;
  dgb[2 * C + c] = (float)(s0 / count);
  dgb[3 * C + c] = (float)(s1 / count);
  if (dyexps) {
    const float invstd = stats[C + c], sc = fabsf(stats[2 * C + c]);
    const float xh = ymax ? __uint_as_float(ymb) * invstd : sqrtf((float)count);
    float bound = __uint_as_float(gmb);
    if (training) bound += fabsf((float)(s0 / count)) + xh * fabsf((float)(s1 / count));
    dyexps[c] = x2_scale_exp(__float_as_uint(bound * sc * 1.0001f));
  }
}

// dy = scale * (g - [training](dbeta/count + xhat*dgamma/count)); optionally g_out = g (grad of res_pre).
// Block (ch, c) as bn_apply_kernel; dmax (may be null): per-channel slots receiving max |dy|.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                    const float* __restrict__ res_pre, const float* __restrict__ stats,
                                    const float* __restrict__ dgb, float* __restrict__ dy, float* __restrict__ g_out,
                                    int N, int C, long S, long chunk_len, float slope, int training, int vec,
                                    unsigned* __restrict__ dmax) {
  const int c = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  const float mean = stats[c], invstd = stats[C + c], sc = stats[2 * C + c], sh = stats[3 * C + c];
  const float k1 = training ? dgb[2 * C + c] : 0.f, k2 = training ? dgb[3 * C + c] * invstd : 0.f;
  float am = 0.f;
  for (int n = 0; n < N; ++n) {
    const long base = ((long)n * C + c) * S;
    if (vec) {
      for (long i = s0 + 4 * tid; i < s1; i += 1024) {
        const float4 yv = *(const float4*)(y + base + i), dv = *(const float4*)(dz + base + i);
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res_pre) rv = *(const float4*)(res_pre + base + i);
        const float ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w};
        float o[4], g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = ys[j] * sc + sh + rs[j];
          g[j] = ds[j] * (u > 0.f ? 1.f : slope);
          o[j] = (g[j] - k1 - (ys[j] - mean) * k2) * sc;
        }
        *(float4*)(dy + base + i) = make_float4(o[0], o[1], o[2], o[3]);
        if (g_out) *(float4*)(g_out + base + i) = make_float4(g[0], g[1], g[2], g[3]);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
      }
    } else {
      for (long i = s0 + tid; i < s1; i += 256) {
        const float yv = y[base + i];
        float u = yv * sc + sh;
        if (res_pre) u += res_pre[base + i];
        const float g = dz[base + i] * (u > 0.f ? 1.f : slope);
        float v = g;
        if (training) v -= dgb[2 * C + c] + (yv - mean) * invstd * dgb[3 * C + c];
        dy[base + i] = v * sc;
        if (g_out) g_out[base + i] = g;
        am = fmaxf(am, fabsf(v * sc));
      }
    }
  }
  if (dmax) dca_cmax_put(am, dmax + (long)c * DCA_AMAX_CSLOTS + ch);
}

// The same (no res_pre, no g_out) writing dy in the packed px2 format with the exponents of bn_bwd_finalize_kernel: block
// (ch, cg) and lane = one voxel x 8 channels as in bn_apply_pack_kernel.
__global__ __launch_bounds__(256) void bn_bwd_apply_pack_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                    const float* __restrict__ stats, const float* __restrict__ dgb,
                                    const int* __restrict__ dyexps, char* __restrict__ dyp, int N, int C, long S,
                                    long chunk_len, float slope, int training) {
  const int cg = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const long s0 = ch * chunk_len, s1 = min(S, s0 + chunk_len);
  float mean[8], sc[8], sh[8], k1[8], k2[8];
  int ex[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mean[j] = stats[c]; sc[j] = stats[2 * C + c]; sh[j] = stats[3 * C + c];
    k1[j] = training ? dca_coherent_loadf(dgb + 2 * C + c) : 0.f;
    k2[j] = training ? dca_coherent_loadf(dgb + 3 * C + c) * stats[C + c] : 0.f;
    ex[j] = dca_coherent_loadi(dyexps + c);
  }
  const long tb = px2_term_bytes(C, S);
  for (int n = 0; n < N; ++n) {
    const long base = ((long)n * C + cg * 8) * S;
    char* ob = dyp + (long)n * 2 * tb + (long)cg * S * 16;
    for (long i = s0 + tid; i < s1; i += 256) {
      float yv[8], dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { yv[j] = y[base + j * S + i]; dv[j] = dz[base + j * S + i]; }
      u16x8 hv, lv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = yv[j] * sc[j] + sh[j];
        const float g = dv[j] * (u > 0.f ? 1.f : slope);
        const float o = (g - k1[j] - (yv[j] - mean[j]) * k2[j]) * sc[j];
        unsigned short h, l;
        px2_split(o, ex[j], h, l);
        hv[j] = h; lv[j] = l;
      }
      *(u16x8*)(ob + i * 16) = hv;
      *(u16x8*)(ob + tb + i * 16) = lv;
    }
  }
}

// ------------------------------------------------------------------------------------ AvgPool3d(3, 2, 1)
// One block row = one (channel, output depth) plane, so the channel base is wave-uniform and every tap is a
// hardware-predicated buffer load (dca_common.h): 27 independent loads in flight per output, no per-tap branches.
__global__ __launch_bounds__(256) void avgpool3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Di,
                                                            int Hi, int Wi, int Do, int Ho, int Wo) {
  const long row = blockIdx.x;
  const int od = (int)(row % Do);
  const long nc = row / Do;
  const long plane = (long)Di * Hi * Wi;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(x + nc * plane, plane * 4);
  const int HW = Ho * Wo, end = min(HW, ((int)blockIdx.y + 1) * 1024);
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < end; i += 256) {
    const int oh = i / Wo, ow = i - oh * Wo;
    float s = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int d = 2 * od - 1 + kd;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int w = 2 * ow - 1 + kw;
          const int ok = (int)((unsigned)d < (unsigned)Di) & (int)((unsigned)h < (unsigned)Hi) & (int)((unsigned)w < (unsigned)Wi);
          s += dca_bload1(xr, ((d * Hi + h) * Wi + w) * 4, ok);
        }
      }
    }
    y[row * HW + i] = s * (1.0f / 27.0f);  // count_include_pad=True: always /27
  }
}

__global__ __launch_bounds__(256) void avgpool3d_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                            const float* __restrict__ res, const float* __restrict__ res2,
                                                            int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  const long row = blockIdx.x;
  const int d = (int)(row % Di);
  const long nc = row / Di;
  const long plane = (long)Do * Ho * Wo;
  const __amdgpu_buffer_rsrc_t gr = dca_rsrc(gy + nc * plane, plane * 4);
  const int HW = Hi * Wi, end = min(HW, ((int)blockIdx.y + 1) * 1024);
  // outputs o with 2o-1+k = i, k in {0,1,2}: o in [ceil((i-1)/2), floor((i+1)/2)] = {i>>1, (i+1)>>1}
  const int d0 = d >> 1, dn = d & 1;
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < end; i += 256) {
    const int h = i / Wi, w = i - h * Wi;
    const int h0 = h >> 1, hn = h & 1, w0 = w >> 1, wn = w & 1;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int od = d0 + a, oh = h0 + b, ow = w0 + c;
          const int ok = (a <= dn) & (b <= hn) & (c <= wn) & (int)(od < Do) & (int)(oh < Ho) & (int)(ow < Wo);
          s += dca_bload1(gr, ((od * Ho + oh) * Wo + ow) * 4, ok);
        }
    float o = s * (1.0f / 27.0f) + (res ? res[row * HW + i] : 0.f);
    if (res2) o += res2[row * HW + i];
    gx[row * HW + i] = o;
  }
}

// LDS-tiled forward for aligned inputs (Wi % 4 == 0): a workgroup produces an 8 x 32 (h x w) tile of one (channel, output
// depth) plane from the 3 x 17 x 65 fine samples under it, staged once with 16-byte loads; same summation order as the
// direct kernel above (bitwise identical results).
__global__ __launch_bounds__(256) void avgpool3d_fwd_tiled_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  constexpr int TH = 8, TW = 32, FH = 2 * TH + 1, PW = 2 * TW + 2;   // LDS column j = w' - (2*ow0 - 1), 65 used
  __shared__ __attribute__((aligned(16))) float tile[3 * FH * PW];
  const int tid = threadIdx.x, ow0 = blockIdx.x * TW, oh0 = blockIdx.y * TH;
  const int od = blockIdx.z % Do;
  const long nc = blockIdx.z / Do;
  const long plane = (long)Di * Hi * Wi;
  const __amdgpu_buffer_rsrc_t xr = dca_rsrc(x + nc * plane, plane * 4);
  constexpr int NQ = 3 * FH * 16, KQ = (NQ + 255) / 256;   // 816 quads -> 4 per thread
  float4 rq[KQ];
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    const int it = tid + 256 * k, q = it & 15, row = it >> 4, a = row / FH, b = row - a * FH;
    const int d = 2 * od - 1 + a, h = 2 * oh0 - 1 + b, w = 2 * ow0 + 4 * q;
    const int ok = (int)(it < NQ) & (int)((unsigned)d < (unsigned)Di) & (int)((unsigned)h < (unsigned)Hi) & (int)(w < Wi);
    rq[k] = dca_bload4(xr, ((d * Hi + h) * Wi + w) * 4, ok);
  }
  float redge = 0.f;
  {
    const int a = tid / FH, b = tid - a * FH;   // 3*FH = 51 left-edge samples (w' = 2*ow0 - 1)
    const int d = 2 * od - 1 + a, h = 2 * oh0 - 1 + b, w = 2 * ow0 - 1;
    const int ok = (int)(tid < 3 * FH) & (int)((unsigned)d < (unsigned)Di) & (int)((unsigned)h < (unsigned)Hi) &
                   (int)((unsigned)w < (unsigned)Wi);
    redge = dca_bload1(xr, ((d * Hi + h) * Wi + w) * 4, ok);
  }
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    const int it = tid + 256 * k;
    if (it < NQ) {
      float* p = tile + (it >> 4) * PW + 1 + 4 * (it & 15);
      p[0] = rq[k].x; p[1] = rq[k].y; p[2] = rq[k].z; p[3] = rq[k].w;
    }
  }
  if (tid < 3 * FH) tile[tid * PW] = redge;
  __syncthreads();
  const int hl = tid >> 5, wl = tid & 31, oh = oh0 + hl, ow = ow0 + wl;
  if (oh >= Ho || ow >= Wo) return;
  float s = 0.f;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = tile + (kd * FH + 2 * hl + kh) * PW + 2 * wl;   // row[0..2] = fine w' = 2ow-1 .. 2ow+1
      const float2 lo = *(const float2*)row;
      s += lo.x;
      s += lo.y;
      s += row[2];
    }
  y[((nc * Do + od) * Ho + oh) * (long)Wo + ow] = s * (1.0f / 27.0f);
}

// Backward with 16-byte stores for aligned outputs (Wi % 4 == 0): a thread produces the 2 x 2 x 4 fine block
// (d in {2bd-1, 2bd}, h in {2bh-1, 2bh}, w = 4t..4t+3) from the 2 x 2 x 3 coarse gradients above it -- an odd fine index lies
// under two coarse cells, an even one under one -- 12 gather loads per four 16-byte stores (round 2: 12 per store; the kernel
// was bound by its load instructions, 478 us per batch-4 launch against a 330 us stream); per-element summation order as in
// the scalar kernel (bitwise identical results).
__global__ __launch_bounds__(256) void avgpool3d_bwd_vec_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                const float* __restrict__ res, const float* __restrict__ res2,
                                                                int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  const int BD = Di / 2 + 1, BH = Hi / 2 + 1;
  const int bd = (int)(blockIdx.x % BD);
  const long nc = blockIdx.x / BD;
  const long plane = (long)Do * Ho * Wo;
  const __amdgpu_buffer_rsrc_t gr = dca_rsrc(gy + nc * plane, plane * 4);
  const int WQ = Wi >> 2, HQ = BH * WQ, end = min(HQ, ((int)blockIdx.y + 1) * 1024);
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < end; i += 256) {
    const int bh = i / WQ, t = i - bh * WQ, c0 = 2 * t;   // fine w = 4t .. 4t+3, coarse c0 .. c0+2
    float g[2][2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int od = bd - 1 + a, oh = bh - 1 + b, ow = c0 + c;
          const int ok = (int)((unsigned)od < (unsigned)Do) & (int)((unsigned)oh < (unsigned)Ho) & (int)(ow < Wo);
          g[a][b][c] = dca_bload1(gr, ((od * Ho + oh) * Wo + ow) * 4, ok);
        }
#pragma unroll
    for (int pd = 0; pd < 2; ++pd) {        // pd = 0: fine d = 2bd - 1 (odd: coarse a = 0, 1); pd = 1: d = 2bd (even: a = 1)
      const int d = 2 * bd - 1 + pd;
      if ((unsigned)d >= (unsigned)Di) continue;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int h = 2 * bh - 1 + ph;
        if ((unsigned)h >= (unsigned)Hi) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // fine w = 4t + e: coarse {w >> 1, (w + 1) >> 1} = c0 + {e >> 1, (e + 1) >> 1}
          float s = 0.f;
#pragma unroll
          for (int a = pd; a < 2; ++a)
#pragma unroll
            for (int b = ph; b < 2; ++b) {
              s += g[a][b][e >> 1];
              s += (e & 1) ? g[a][b][(e + 1) >> 1] : 0.f;
            }
          o[e] = s * (1.0f / 27.0f);
        }
        const long off = ((nc * Di + d) * Hi + h) * (long)Wi + 4 * t;
        if (res) {   // + other gradients of the same input (ops._PoolFork): saves autograd's separate accumulation passes
          const float4 r = *(const float4*)(res + off);
          o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        if (res2) {
          const float4 r = *(const float4*)(res2 + off);
          o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        *(float4*)(gx + off) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ trilinear
// align_corners=False, integer scale s: src = (dst+0.5)/s - 0.5 clamped at 0 (ATen area_pixel_compute_source_index)
__device__ __forceinline__ void lin_src(int o, float rs, int n, int& i0, int& i1, float& l0, float& l1) {
  float src = rs * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

__global__ void trilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long NC, int Di, int Hi,
                                     int Wi, int s) {
  const int Do = Di * s, Ho = Hi * s, Wo = Wi * s;
  const float rs = 1.0f / (float)s;
  const long total = NC * Do * Ho * Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ow = idx % Wo;
    long t = idx / Wo;
    const int oh = t % Ho; t /= Ho;
    const int od = t % Do;
    const long nc = t / Do;
    int d0, d1, h0, h1, w0, w1;
    float ld0, ld1, lh0, lh1, lw0, lw1;
    lin_src(od, rs, Di, d0, d1, ld0, ld1);
    lin_src(oh, rs, Hi, h0, h1, lh0, lh1);
    lin_src(ow, rs, Wi, w0, w1, lw0, lw1);
    const float* p = x + nc * Di * Hi * Wi;
#define XV(d, h, w) p[((long)(d) * Hi + (h)) * Wi + (w)]
    const float v = ld0 * (lh0 * (lw0 * XV(d0, h0, w0) + lw1 * XV(d0, h0, w1)) +
                           lh1 * (lw0 * XV(d0, h1, w0) + lw1 * XV(d0, h1, w1))) +
                    ld1 * (lh0 * (lw0 * XV(d1, h0, w0) + lw1 * XV(d1, h0, w1)) +
                           lh1 * (lw0 * XV(d1, h1, w0) + lw1 * XV(d1, h1, w1)));
#undef XV
    y[idx] = v;
  }
}

// weight with which output o contributes to input i along one dim (0 if none)
__device__ __forceinline__ float lin_w(int o, int i, float rs, int n) {
  int i0, i1;
  float l0, l1;
  lin_src(o, rs, n, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// gather form of the trilinear backward: input i collects outputs o in [s*i - ceil(s/2), s*i + 3s/2 - 1]
// (even s exact; odd s a superset -- weights are match-tested per candidate, so a superset is safe).
__global__ void trilinear_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, long NC, int Di, int Hi,
                                     int Wi, int s) {
  const int Do = Di * s, Ho = Hi * s, Wo = Wi * s;
  const float rs = 1.0f / (float)s;
  const long total = NC * Di * Hi * Wi;
  const int lo_off = (s + 1) / 2, hi_off = (3 * s) / 2 - 1 + (s & 1);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int w = idx % Wi;
    long t = idx / Wi;
    const int h = t % Hi; t /= Hi;
    const int d = t % Di;
    const long nc = t / Di;
    const float* p = gy + nc * Do * Ho * Wo;
    const int od0 = max(0, s * d - lo_off), od1 = min(Do - 1, s * d + hi_off);
    const int oh0 = max(0, s * h - lo_off), oh1 = min(Ho - 1, s * h + hi_off);
    const int ow0 = max(0, s * w - lo_off), ow1 = min(Wo - 1, s * w + hi_off);
    float acc = 0.f;
    for (int od = od0; od <= od1; ++od) {
      const float wd = lin_w(od, d, rs, Di);
      if (wd == 0.f) continue;
      for (int oh = oh0; oh <= oh1; ++oh) {
        const float wh = lin_w(oh, h, rs, Hi);
        if (wh == 0.f) continue;
        const float* row = p + ((long)od * Ho + oh) * Wo;
        float r = 0.f;
        for (int ow = ow0; ow <= ow1; ++ow) r += lin_w(ow, w, rs, Wi) * row[ow];
        acc += wd * wh * r;
      }
    }
    gx[idx] = acc;
  }
}

// x2 up-sampling, specialised: per dim o=2m -> 0.25 x[m-1] + 0.75 x[m] (m=0: x[0]); o=2m+1 -> 0.75 x[m] + 0.25 x[m+1]
// (m=n-1: x[n-1]).  One thread per coarse cell writes its 2x2x2 outputs (two float2 stores per row pair); grid
// (W tiles of 64, H tiles of 4, NC*D) so no 64-bit div/mod chains; a wave is 64 consecutive w of one row (a 256-wide block
// per row left 136 of 256 lanes idle at W = 120: 276 -> 259 us per batch-4 launch).  Measured and withdrawn in round 3: two
// cells per thread with 16-byte stores (308 us: 36 gather loads per thread), the left / right samples from the neighbouring
// lanes instead of loads (355 us: 18 ds_bpermute + divergent edge loads).
__global__ __launch_bounds__(256) void trilinear_up2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int Di, int Hi, int Wi) {
  const int mw = blockIdx.x * 64 + (threadIdx.x & 63), mh = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int md = blockIdx.z % Di;
  const long nc = blockIdx.z / Di;
  if (mw >= Wi || mh >= Hi) return;
  const float* p = x + nc * Di * Hi * Wi;
  const int dm = max(md - 1, 0), dp = min(md + 1, Di - 1), hm = max(mh - 1, 0), hp = min(mh + 1, Hi - 1);
  const int wm = max(mw - 1, 0), wp = min(mw + 1, Wi - 1);
  // rows: (d in {dm, md, dp}) x (h in {hm, mh, hp}), each reduced along w to the two outputs (2mw, 2mw+1)
  float r0[3][3], r1[3][3];
  const int ds[3] = {dm, md, dp}, hs[3] = {hm, mh, hp};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float* row = p + ((long)ds[a] * Hi + hs[b]) * Wi;
      const float xm = row[wm], x0 = row[mw], xp = row[wp];
      r0[a][b] = 0.25f * xm + 0.75f * x0;
      r1[a][b] = 0.75f * x0 + 0.25f * xp;
    }
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  float* q = y + nc * (2L * Di) * Ho * Wo;
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      // along d: pd=0 -> .25*[dm] + .75*[md]; pd=1 -> .75*[md] + .25*[dp]; same along h
      const int a0 = pd ? 1 : 0, a1 = pd ? 2 : 1, b0 = ph ? 1 : 0, b1 = ph ? 2 : 1;
      const float wa0 = pd ? 0.75f : 0.25f, wa1 = pd ? 0.25f : 0.75f, wb0 = ph ? 0.75f : 0.25f, wb1 = ph ? 0.25f : 0.75f;
      float2 o;
      o.x = wa0 * (wb0 * r0[a0][b0] + wb1 * r0[a0][b1]) + wa1 * (wb0 * r0[a1][b0] + wb1 * r0[a1][b1]);
      o.y = wa0 * (wb0 * r1[a0][b0] + wb1 * r1[a0][b1]) + wa1 * (wb0 * r1[a1][b0] + wb1 * r1[a1][b1]);
      *(float2*)(q + ((long)(2 * md + pd) * Ho + 2 * mh + ph) * Wo + 2 * mw) = o;
    }
}

// per-dim backward weights of the x2 up-sampling: input i gets outputs 2i-1, 2i, 2i+1, 2i+2 with weights
// (.25, .75, .75, .25); the clamped ends fold the missing neighbour's weight into the edge sample.
__device__ __forceinline__ void up2_bwd_w(int i, int n, float w[4]) {
  w[0] = (i > 0) ? 0.25f : 0.f;
  w[1] = (i > 0) ? 0.75f : 1.0f;
  w[2] = (i < n - 1) ? 0.75f : 1.0f;
  w[3] = (i < n - 1) ? 0.25f : 0.f;
}

__global__ __launch_bounds__(256) void trilinear_up2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                int Di, int Hi, int Wi) {
  const int w = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  const int d = blockIdx.z % Di;
  const long nc = blockIdx.z / Di;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const long plane = (2L * Di) * Ho * Wo;
  const __amdgpu_buffer_rsrc_t gr = dca_rsrc(gy + nc * plane, plane * 4);
  const int wok = (int)(w < Wi);
  float wd[4], wh[4], ww[4];
  up2_bwd_w(d, Di, wd); up2_bwd_w(h, Hi, wh); up2_bwd_w(w, Wi, ww);
  // 16 rows x (one 8-byte + two 4-byte) predicated buffer loads, all independent; a zero weight means the sample
  // is outside the tensor, and the masked load returns 0 for it
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int od = 2 * d - 1 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int oh = 2 * h - 1 + b;
      const int ok = wok & (int)(wd[a] != 0.f) & (int)(wh[b] != 0.f);
      const int off = ((od * Ho + oh) * Wo + 2 * w) * 4;
      const float2 mid = dca_bload2(gr, off, ok);
      const float lo = dca_bload1(gr, off - 4, ok & (int)(ww[0] != 0.f));
      const float hi = dca_bload1(gr, off + 8, ok & (int)(ww[3] != 0.f));
      float r = ww[1] * mid.x + ww[2] * mid.y;
      r += ww[0] * lo;
      r += ww[3] * hi;
      acc += wd[a] * wh[b] * r;
    }
  }
  if (wok) gx[((nc * Di + d) * Hi + h) * Wi + w] = acc;
}

// LDS-tiled form for aligned inputs: a workgroup produces an 8 x 32 (h x w) coarse tile of one (channel, depth) plane
// from the 4 x 18 x 66 fine samples it depends on, staged once with 16-byte loads (the direct form above re-reads each
// fine sample ~8 times through L1 with 48 load instructions per output).  Same arithmetic, same summation order.
__global__ __launch_bounds__(256) void trilinear_up2_bwd_tiled_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                      int Di, int Hi, int Wi) {
  constexpr int TH = 8, TW = 32, FH = 2 * TH + 2, FW = 2 * TW + 2, PW = FW + 2;   // LDS row: j = w' - (2*w0 - 1)
  __shared__ __attribute__((aligned(16))) float tile[4 * FH * PW];
  const int tid = threadIdx.x, w0 = blockIdx.x * TW, h0 = blockIdx.y * TH;
  const int d = blockIdx.z % Di;
  const long nc = blockIdx.z / Di;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const long plane = (long)Do * Ho * Wo;
  const __amdgpu_buffer_rsrc_t gr = dca_rsrc(gy + nc * plane, plane * 4);
  // interior: 4 * FH rows x 16 quads (fine w' = 2*w0 + 4q .. +3); edges: two scalars per row (2*w0 - 1, 2*w0 + 64)
  constexpr int NQ = 4 * FH * 16, KQ = (NQ + 255) / 256;   // 1152 quads -> 5 per thread, all loads issued up front
  float4 rq[KQ];
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    const int it = tid + 256 * k, q = it & 15, row = it >> 4, a = row / FH, b = row - a * FH;
    const int od = 2 * d - 1 + a, oh = 2 * h0 - 1 + b, ow = 2 * w0 + 4 * q;
    const int ok = (int)(it < NQ) & (int)((unsigned)od < (unsigned)Do) & (int)((unsigned)oh < (unsigned)Ho) &
                   (int)(ow < Wo);  // Wo % 4 == 0: a quad is inside or outside as a whole
    rq[k] = dca_bload4(gr, ((od * Ho + oh) * Wo + ow) * 4, ok);
  }
  float redge = 0.f;
  {
    const int side = tid & 1, row = tid >> 1, a = row / FH, b = row - a * FH;   // 4*FH*2 = 144 edge samples
    const int od = 2 * d - 1 + a, oh = 2 * h0 - 1 + b, ow = side ? 2 * w0 + 2 * TW : 2 * w0 - 1;
    const int ok = (int)(tid < 4 * FH * 2) & (int)((unsigned)od < (unsigned)Do) & (int)((unsigned)oh < (unsigned)Ho) &
                   (int)((unsigned)ow < (unsigned)Wo);
    redge = dca_bload1(gr, ((od * Ho + oh) * Wo + ow) * 4, ok);
  }
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    const int it = tid + 256 * k;
    if (it < NQ) {
      float* p = tile + (it >> 4) * PW + 1 + 4 * (it & 15);
      p[0] = rq[k].x; p[1] = rq[k].y; p[2] = rq[k].z; p[3] = rq[k].w;
    }
  }
  if (tid < 4 * FH * 2) tile[(tid >> 1) * PW + ((tid & 1) ? FW - 1 : 0)] = redge;
  __syncthreads();
  const int hl = tid >> 5, wl = tid & 31, h = h0 + hl, w = w0 + wl;
  if (h >= Hi || w >= Wi) return;
  float wd[4], wh[4], ww[4];
  up2_bwd_w(d, Di, wd); up2_bwd_w(h, Hi, wh); up2_bwd_w(w, Wi, ww);
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float* row = tile + (a * FH + 2 * hl + b) * PW + 2 * wl;   // row[0..3] = fine w' = 2w-1 .. 2w+2
      const float2 lo = *(const float2*)row, hi = *(const float2*)(row + 2);
      float r = ww[1] * lo.y + ww[2] * hi.x;
      r += ww[0] * lo.x;
      r += ww[3] * hi.y;
      acc += wd[a] * wh[b] * r;
    }
  gx[((nc * Di + d) * Hi + h) * Wi + w] = acc;
}

static int ew_grid(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192);
}

static void chunking(long S, int C, int* nchunk, long* chunk_len) {
  // enough blocks to fill 256 CUs a few times over, chunks a multiple of 1024 floats; at most DCA_AMAX_CSLOTS chunks (a
  // chunk index is also a slot of the per-channel operand maxima)
  long want = (2048 + C - 1) / C;
  if (want > 256) want = 256;      // (<= DCA_AMAX_CSLOTS; every consumer reads all written slots)
  long len = (S + want - 1) / want;
  len = ((len + 1023) / 1024) * 1024;
  *chunk_len = len;
  *nchunk = (int)((S + len - 1) / len);
}

extern "C" int dca_bn_num_chunks(int C, long S) {
  int n; long l;
  chunking(S, C, &n, &l);
  return n;
}
// chunks (= slots per channel) of the 8-channel-group kernels that write the packed px2 format
extern "C" int dca_bn_pack_chunks(int C, long S) {
  int n; long l;
  chunking(S, (C + 7) / 8, &n, &l);
  return n;
}

extern "C" int dca_bn_stats(const float* x, double* part, int N, int C, long S, hipStream_t stream) {
  DCA_REQUIRE(x && part && N > 0 && C > 0 && S > 0 && C <= 65535);
  int nchunk; long len;
  chunking(S, C, &nchunk, &len);
  const int vec = (S % 4 == 0) && (((uintptr_t)x & 15) == 0);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C, nchunk), dim3(256), 0, stream, x, part, N, C, S, nchunk, len, vec);
  return dca_launch_status();
}

extern "C" int dca_bn_finalize(const double* part, int nchunk, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int training,
                               float* stats, int* zexps, const unsigned* rpre_slots, int rpre_n,
                               const unsigned* rpost_slots, int rpost_n, int C, hipStream_t stream) {
  DCA_REQUIRE(stats && C > 0 && (training ? (part != nullptr && nchunk > 0) : (running_mean && running_var)));
  DCA_REQUIRE(zexps == nullptr || training);      // the bound behind zexps holds for batch statistics only
  DCA_REQUIRE((rpre_slots == nullptr || (rpre_n > 0 && rpre_n <= DCA_AMAX_CSLOTS)) &&
              (rpost_slots == nullptr || (rpost_n > 0 && rpost_n <= DCA_AMAX_CSLOTS)));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, stream, part, nchunk, count, gamma, beta,
                     running_mean, running_var, momentum, eps, training, stats, zexps, rpre_slots, rpre_n, rpost_slots,
                     rpost_n, C);
  return dca_launch_status();
}

extern "C" int dca_bn_finalize_centered(const double* part, int nchunk, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float momentum, float eps, float* stats,
                                        int* zexps, const unsigned* rpre_slots, int rpre_n, const unsigned* rpost_slots,
                                        int rpost_n, int C, hipStream_t stream) {
  DCA_REQUIRE(part && nchunk > 0 && stats && C > 0 && ((running_mean == nullptr) == (running_var == nullptr)));
  DCA_REQUIRE((rpre_slots == nullptr || (rpre_n > 0 && rpre_n <= DCA_AMAX_CSLOTS)) &&
              (rpost_slots == nullptr || (rpost_n > 0 && rpost_n <= DCA_AMAX_CSLOTS)));
  hipLaunchKernelGGL(bn_finalize_centered_kernel, dim3(C), dim3(64), 0, stream, part, nchunk, gamma, beta, running_mean,
                     running_var, momentum, eps, stats, zexps, rpre_slots, rpre_n, rpost_slots, rpost_n, C);
  return dca_launch_status();
}

extern "C" int dca_bn_apply(const float* y, const float* stats, const float* res_pre, const float* res_post, float* z,
                            int N, int C, long S, float slope, unsigned* zmax, unsigned* ymax, hipStream_t stream) {
  DCA_REQUIRE(y && stats && z && N > 0 && C > 0 && S > 0 && C <= 65535);
  int nchunk; long len;
  chunking(S, C, &nchunk, &len);
  const uintptr_t al = (uintptr_t)y | (uintptr_t)z | (uintptr_t)res_pre | (uintptr_t)res_post;
  const int vec = (S % 4 == 0) && ((al & 15) == 0);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(nchunk, C), dim3(256), 0, stream, y, stats, res_pre, res_post, z, N, C, S, len,
                     slope, vec, zmax, ymax);
  return dca_launch_status();
}

extern "C" int dca_bn_apply_pack(const float* y, const float* stats, const int* zexps, void* zp, int N, int C, long S,
                                 float slope, unsigned* ymax, const float* res_pre, const float* res_post, float* zf,
                                 unsigned* zmax, hipStream_t stream) {
  DCA_REQUIRE(y && zexps && zp && N > 0 && C > 0 && C % 8 == 0 && S > 0 && C / 8 <= 65535 && ((((uintptr_t)zp) & 15) == 0));
  DCA_REQUIRE(stats != nullptr || (ymax == nullptr && slope == 1.f));
  int nchunk; long len;
  chunking(S, C / 8, &nchunk, &len);
  const uintptr_t al = (uintptr_t)y | (uintptr_t)res_pre | (uintptr_t)res_post | (uintptr_t)zf;
  // four voxels per lane pay when residual streams are read beside y (batch-4 layer shape, tools/bn_time.py: 631 vs 726 us with
  // fp32 copy + one residual); without them the one-voxel form is faster (332 vs 449 us packed only, 563 vs 584 with the copy:
  // its packed stores are contiguous per instruction, the four-voxel form's are 16 of every 64 bytes)
  const bool four = (res_pre || res_post) && S % 4 == 0 && (al & 15) == 0;       // chunks are multiples of 1024
  auto kern = four ? bn_apply_pack4_kernel : bn_apply_pack_kernel;
  hipLaunchKernelGGL(kern, dim3(nchunk, C / 8), dim3(256), 0, stream, y, stats, zexps, (char*)zp, N, C, S, len, slope, ymax,
                     res_pre, res_post, zf, zmax);
  return dca_launch_status();
}

extern "C" int dca_cmax_f32(const float* x, int N, int C, long S, unsigned* slots, hipStream_t stream) {
  DCA_REQUIRE(x && slots && N > 0 && C > 0 && S > 0 && C <= 65535);
  int nchunk; long len;
  chunking(S, C, &nchunk, &len);
  const int vec = (S % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
  hipLaunchKernelGGL(cmax_kernel, dim3(nchunk, C), dim3(256), 0, stream, x, N, C, S, len, vec, slots);
  return dca_launch_status();
}

extern "C" int dca_cmax_exps(const unsigned* slots, int nslots, int C, int* exps, hipStream_t stream) {
  DCA_REQUIRE(slots && exps && C > 0 && nslots > 0 && nslots <= DCA_AMAX_CSLOTS);
  hipLaunchKernelGGL(cmax_exps_kernel, dim3(C), dim3(64), 0, stream, slots, nslots, exps);
  return dca_launch_status();
}

extern "C" int dca_bn_backward(const float* dz, const float* y, const float* res_pre, const float* stats,
                               double* part, float* dgb, float* dy, float* g_out, int N, int C, long S, float slope,
                               int training, unsigned* dmax, hipStream_t stream) {
  DCA_REQUIRE(dz && y && stats && part && dgb && dy && N > 0 && C > 0 && S > 0 && C <= 65535);
  int nchunk; long len;
  chunking(S, C, &nchunk, &len);
  const uintptr_t al = (uintptr_t)dz | (uintptr_t)y | (uintptr_t)res_pre | (uintptr_t)dy | (uintptr_t)g_out;
  const int vec = (S % 4 == 0) && ((al & 15) == 0);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, nchunk), dim3(256), 0, stream, dz, y, res_pre, stats, part, N, C, S,
                     nchunk, len, slope, vec, (unsigned*)nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, stream, part, nchunk, (double)N * (double)S, dgb, C,
                     stats, training, (const unsigned*)nullptr, (const unsigned*)nullptr, 0, (int*)nullptr);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nchunk, C), dim3(256), 0, stream, dz, y, res_pre, stats, dgb, dy, g_out, N, C,
                     S, len, slope, training, vec, dmax);
  return dca_launch_status();
}

// BatchNorm backward (no res_pre) writing dy in the packed px2 format: dyp (N*C*S*4 bytes), dyexps (C ints, out) = the
// per-channel exponents dy was scaled by, from the bound of bn_bwd_finalize_kernel; gmax = scratch of C * DCA_AMAX_CSLOTS
// words; ymax / ymax_slots = the per-channel max |y - mean| slots the forward apply pass emitted (may be null / 0).
extern "C" int dca_bn_backward_pack(const float* dz, const float* y, const float* stats, double* part, float* dgb, void* dyp,
                                    int* dyexps, unsigned* gmax, const unsigned* ymax, int ymax_slots, int N, int C, long S,
                                    float slope, int training, hipStream_t stream) {
  DCA_REQUIRE(dz && y && stats && part && dgb && dyp && dyexps && gmax && N > 0 && C > 0 && C % 8 == 0 && S > 0 && C <= 65535);
  DCA_REQUIRE((ymax == nullptr || (ymax_slots > 0 && ymax_slots <= DCA_AMAX_CSLOTS)) && ((((uintptr_t)dyp) & 15) == 0));
  int nchunk; long len;
  chunking(S, C, &nchunk, &len);
  const uintptr_t al = (uintptr_t)dz | (uintptr_t)y;
  const int vec = (S % 4 == 0) && ((al & 15) == 0);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, nchunk), dim3(256), 0, stream, dz, y, (const float*)nullptr, stats, part,
                     N, C, S, nchunk, len, slope, vec, gmax);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, stream, part, nchunk, (double)N * (double)S, dgb, C,
                     stats, training, (const unsigned*)gmax, ymax, ymax_slots, dyexps);
  int pchunk; long plen;
  chunking(S, C / 8, &pchunk, &plen);
  hipLaunchKernelGGL(bn_bwd_apply_pack_kernel, dim3(pchunk, C / 8), dim3(256), 0, stream, dz, y, stats, dgb, dyexps,
                     (char*)dyp, N, C, S, plen, slope, training);
  return dca_launch_status();
}

extern "C" int dca_avgpool3d_fwd(const float* x, float* y, long NC, int Di, int Hi, int Wi, hipStream_t stream) {
  DCA_REQUIRE(x && y && NC > 0 && Di > 0 && Hi > 0 && Wi > 0);
  const int Do = (Di + 1) / 2, Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;  // floor((i+2-3)/2)+1
  DCA_REQUIRE((long)Di * Hi * Wi * 4 < 0x7ffffff0L && NC * Do < 0x7fffffffL);  // 32-bit offsets inside one channel
  if (Wi % 4 == 0 && (((uintptr_t)x) & 15) == 0 && NC * Do <= 65535 && cdiv(Ho, 8) <= 65535) {
    hipLaunchKernelGGL(avgpool3d_fwd_tiled_kernel, dim3(cdiv(Wo, 32), cdiv(Ho, 8), (unsigned)(NC * Do)), dim3(256), 0, stream,
                       x, y, Di, Hi, Wi, Do, Ho, Wo);
    return dca_launch_status();
  }
  hipLaunchKernelGGL(avgpool3d_fwd_kernel, dim3((unsigned)(NC * Do), cdiv((long)Ho * Wo, 1024)), dim3(256), 0, stream, x,
                     y, Di, Hi, Wi, Do, Ho, Wo);
  return dca_launch_status();
}

extern "C" int dca_avgpool3d_bwd(const float* gy, float* gx, const float* res, const float* res2, long NC, int Di, int Hi,
                                 int Wi, hipStream_t stream) {
  DCA_REQUIRE(gy && gx && NC > 0 && Di > 0 && Hi > 0 && Wi > 0 && (res || !res2));
  const int Do = (Di + 1) / 2, Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
  DCA_REQUIRE((long)Di * Hi * Wi * 4 < 0x7ffffff0L && NC * Di < 0x7fffffffL);
  if (Wi % 4 == 0 && ((((uintptr_t)gx) | ((uintptr_t)res) | ((uintptr_t)res2)) & 15) == 0) {
    hipLaunchKernelGGL(avgpool3d_bwd_vec_kernel, dim3((unsigned)(NC * (Di / 2 + 1)), cdiv((long)(Hi / 2 + 1) * (Wi / 4), 1024)),
                       dim3(256), 0, stream, gy, gx, res, res2, Di, Hi, Wi, Do, Ho, Wo);
    return dca_launch_status();
  }
  hipLaunchKernelGGL(avgpool3d_bwd_kernel, dim3((unsigned)(NC * Di), cdiv((long)Hi * Wi, 1024)), dim3(256), 0, stream, gy,
                     gx, res, res2, Di, Hi, Wi, Do, Ho, Wo);
  return dca_launch_status();
}

extern "C" int dca_trilinear_fwd(const float* x, float* y, long NC, int Di, int Hi, int Wi, int scale,
                                 hipStream_t stream) {
  DCA_REQUIRE(x && y && NC > 0 && Di > 0 && Hi > 0 && Wi > 0 && scale >= 1);
  if (scale == 2 && Hi <= 65535 && NC * Di <= 65535 && (((uintptr_t)y & 7) == 0)) {
    hipLaunchKernelGGL(trilinear_up2_fwd_kernel, dim3(cdiv(Wi, 64), cdiv(Hi, 4), (unsigned)(NC * Di)), dim3(256), 0, stream,
                       x, y, Di, Hi, Wi);
    return dca_launch_status();
  }
  const long total = NC * Di * Hi * Wi * scale * scale * scale;
  hipLaunchKernelGGL(trilinear_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, x, y, NC, Di, Hi, Wi, scale);
  return dca_launch_status();
}

extern "C" int dca_trilinear_bwd(const float* gy, float* gx, long NC, int Di, int Hi, int Wi, int scale,
                                 hipStream_t stream) {
  DCA_REQUIRE(gy && gx && NC > 0 && Di > 0 && Hi > 0 && Wi > 0 && scale >= 1);
  if (scale == 2 && cdiv(Hi, 8) <= 65535 && NC * Di <= 65535 && 32L * Di * Hi * Wi < 0x7ffffff0L && Wi % 2 == 0 &&
      (((uintptr_t)gy & 15) == 0)) {
    hipLaunchKernelGGL(trilinear_up2_bwd_tiled_kernel, dim3(cdiv(Wi, 32), cdiv(Hi, 8), (unsigned)(NC * Di)), dim3(256), 0,
                       stream, gy, gx, Di, Hi, Wi);
    return dca_launch_status();
  }
  if (scale == 2 && Hi <= 65535 && NC * Di <= 65535 && 32L * Di * Hi * Wi < 0x7ffffff0L && (((uintptr_t)gy & 7) == 0)) {
    hipLaunchKernelGGL(trilinear_up2_bwd_kernel, dim3(cdiv(Wi, 256), Hi, (unsigned)(NC * Di)), dim3(256), 0, stream, gy, gx,
                       Di, Hi, Wi);
    return dca_launch_status();
  }
  hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(ew_grid(NC * Di * Hi * Wi)), dim3(256), 0, stream, gy, gx, NC, Di, Hi,
                     Wi, scale);
  return dca_launch_status();
}
