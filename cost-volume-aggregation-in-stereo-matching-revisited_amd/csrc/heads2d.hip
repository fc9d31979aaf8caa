// The two steps right after the hot path, each as one bandwidth-bound kernel per direction:
//
//  * convex x4 up-sampling of the 1/4-res disparity -- PropgationNet_4x.forward after its conv
//    (models/submodule.py:366-373, identical copy models/gwcnet_dca_g.py:114-124):
//        nb   = F.unfold(4*disp, [3,3], padding=1).view(b,1,9,1,1,h,w)
//        mask = softmax(conv(guidance).view(b,1,9,4,4,h,w), dim=2);  up = sum(mask*nb, 2) -> pixel shuffle (b,1,4h,4w)
//    One thread per 1/4-res cell: 144 coalesced logit loads (channel = k*16 + i*4 + j), 16 softmaxes over the 9
//    neighbours, four 16-byte row stores.  The reference materialises the (b,9,4,4,h,w) softmax, the product and the
//    permuted copy: ~5 passes over 144 channels instead of one.
//
//  * stereo focal loss of one level -- StereoFocalLoss.loss_per_level with LaplaceDisp2Prob (models/loss.py:206-240,
//    60-128): log_softmax of the estimate over the disparity axis, Laplace target softmax(-|k - gt|) of the (already
//    pooled) ground truth, focal weight (1 - P)^-coef, validity masks, mean over ALL pixels.  The reference builds
//    five (B,48,h,w) temporaries per level; here a thread owns one pixel of one level, keeps its target coefficients
//    in an LDS column, and one launch covers ALL levels of that resolution; one partial sum per workgroup and level
//    (order-fixed reduction, no atomics).  The estimate handed in by GwcNet is already a
//    softmax output and is log_softmax-ed again -- reproduced, not fixed.
#include "dca_common.h"
#include "../../include/dca_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ convex up-sampling
__device__ __forceinline__ void load_nb(const float* __restrict__ d, int h, int w, int y, int x, float (&nb)[9]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool ok = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
    nb[k] = ok ? 4.f * d[(long)yy * w + xx] : 0.f;     // F.unfold zero padding of 4*disp
  }
}

__global__ __launch_bounds__(256) void convex_up4_fwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ disp, float* __restrict__ up,
                                                             int h, int w) {
  const int b = blockIdx.y, cell = blockIdx.x * 256 + threadIdx.x, hw = h * w;
  if (cell >= hw) return;
  const int y = cell / w, x = cell - y * w;
  float nb[9];
  load_nb(disp + (long)b * hw, h, w, y, x, nb);
  const float* lg = logits + (long)b * 144 * hw + cell;
  float* o = up + (long)b * 16 * hw + (long)(4 * y) * (4 * w) + 4 * x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[9], m = -INFINITY;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        v[k] = lg[(long)(k * 16 + i * 4 + j) * hw];
        m = fmaxf(m, v[k]);
      }
      float s = 0.f, acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float e = expf(v[k] - m);
        s += e;
        acc += e * nb[k];
      }
      r[j] = acc / s;
    }
    *(float4*)(o + (long)i * (4 * w)) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// glogits[k,i,j] = m_k * gup_ij * (nb_k - up_ij);  wk[k] = 4 * sum_ij gup_ij * m_k,ij  (gathered by the second kernel)
__global__ __launch_bounds__(256) void convex_up4_bwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ disp,
                                                             const float* __restrict__ gup,
                                                             float* __restrict__ glogits, float* __restrict__ wk,
                                                             int h, int w) {
  const int b = blockIdx.y, cell = blockIdx.x * 256 + threadIdx.x, hw = h * w;
  if (cell >= hw) return;
  const int y = cell / w, x = cell - y * w;
  float nb[9], wsum[9];
  load_nb(disp + (long)b * hw, h, w, y, x, nb);
#pragma unroll
  for (int k = 0; k < 9; ++k) wsum[k] = 0.f;
  const float* lg = logits + (long)b * 144 * hw + cell;
  float* gl = glogits + (long)b * 144 * hw + cell;
  const float* g = gup + (long)b * 16 * hw + (long)(4 * y) * (4 * w) + 4 * x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 g4 = *(const float4*)(g + (long)i * (4 * w));
    const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[9], m = -INFINITY;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        v[k] = lg[(long)(k * 16 + i * 4 + j) * hw];
        m = fmaxf(m, v[k]);
      }
      float s = 0.f, acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        v[k] = expf(v[k] - m);
        s += v[k];
        acc += v[k] * nb[k];
      }
      const float inv = 1.f / s, upv = acc * inv;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float mk = v[k] * inv;
        gl[(long)(k * 16 + i * 4 + j) * hw] = mk * gs[j] * (nb[k] - upv);
        wsum[k] += gs[j] * mk;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[((long)b * 9 + k) * hw + cell] = 4.f * wsum[k];
}

// gdisp[y',x'] = sum_k wk[k][y' - (k/3 - 1)][x' - (k%3 - 1)]   (cell (y,x) sees (y',x') as its neighbour k)
__global__ __launch_bounds__(256) void convex_up4_bwd_disp_kernel(const float* __restrict__ wk, float* __restrict__ gdisp,
                                                                  int h, int w) {
  const int b = blockIdx.y, cell = blockIdx.x * 256 + threadIdx.x, hw = h * w;
  if (cell >= hw) return;
  const int y = cell / w, x = cell - y * w;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y - (k / 3 - 1), xx = x - (k % 3 - 1);
    if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) s += wk[((long)b * 9 + k) * hw + (long)yy * w + xx];
  }
  gdisp[(long)b * hw + cell] = s;
}

// ------------------------------------------------------------------------------------------------ stereo focal loss
constexpr int FL_KMAX = 256;  // disparity bins per level (48 at D = 192); columns live in LDS: K * NT floats, with
                              // NT = 256 threads per workgroup up to 64 bins and 64 threads beyond

// Laplace target of one pixel (models/loss.py:87-96,124-126): P_k = softmax_k(-|k - gt*inner|) * inner + 1e-40, with
// `valid` = 0 < gt < K (loss_per_level:220-222) and `inner` = 0 < gt*valid < K-1 (Disp2Prob.getProb:87-89).
struct Target { float g, inner, valid, m, inv; };
__device__ __forceinline__ Target make_target(float gt, int K, int any_valid) {
  Target t;
  t.valid = (gt > 0.f && gt < (float)K) ? 1.f : 0.f;
  const float mg = gt * t.valid;
  t.inner = (mg > 0.f && mg < (float)(K - 1)) ? 1.f : 0.f;
  t.g = mg * t.inner;
  // softmax(-|k - g|) over k = 0..K-1: the maximum is at the integer nearest to g (clamped)
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < K; ++k) m = fmaxf(m, -fabsf((float)k - t.g));
  for (int k = 0; k < K; ++k) s += expf(-fabsf((float)k - t.g) - m);
  t.m = m;
  t.inv = 1.f / s;
  if (!any_valid) { t.inner = 0.f; t.inv = 0.f; }   // "no valid point" branch: zero target (loss.py:224-227)
  return t;
}
__device__ __forceinline__ float target_prob(const Target& t, int k, int any_valid) {
  const float p = expf(-fabsf((float)k - t.g) - t.m) * t.inv * t.inner;
  return any_valid ? p + 1e-40f : 0.f;
}
__device__ __forceinline__ float focal_weight(float p, float coef) { return powf(1.f - p, -coef); }

constexpr int FL_LMAX = 8;    // levels of equal resolution handled by one launch (DCANet: 5)
struct FocalArgs {
  const float* est[FL_LMAX];
  float* gest[FL_LMAX];
  float weight[FL_LMAX];
  int nlev;
};

// part[(lev*B + b)*nblk + block] = sum over the block's pixels of  -sum_k P_k * log_softmax(est)_k * (1-P_k)^-coef * valid.
// The target (P_k, focal weight) depends only on the ground truth; grid.z = level.
template <int NT>
__global__ __launch_bounds__(NT) void focal_fwd_kernel(FocalArgs a, const float* __restrict__ gt,
                                                        const int* __restrict__ any_valid_p, double* __restrict__ part,
                                                        int K, long HW, float coef) {
  extern __shared__ float col[];          // [K][NT] target coefficient c_k = P_k * w_k * valid of this thread's pixel
  __shared__ double red[4];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long pix = (long)blockIdx.x * NT + tid;
  const int any_valid = *any_valid_p;
  const bool live = pix < HW;
  if (live) {
    const Target t = make_target(gt[(long)b * HW + pix], K, any_valid);
    for (int k = 0; k < K; ++k) {
      const float p = target_prob(t, k, any_valid);
      col[k * NT + tid] = p * focal_weight(p, coef) * t.valid;
    }
  }
  {
    const int lev = blockIdx.z;     // one workgroup per (pixel block, batch, level): 5x the workgroups of a level loop
    float loss = 0.f;
    if (live) {
      const float* e = a.est[lev] + (long)b * K * HW + pix;
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, e[(long)k * HW]);
      float s = 0.f, acc = 0.f, csum = 0.f;
      for (int k = 0; k < K; ++k) {
        const float v = e[(long)k * HW], c = col[k * NT + tid];
        s += expf(v - m);
        acc += c * v;
        csum += c;
      }
      loss = -(acc - csum * (m + logf(s)));    // -sum_k c_k * (v_k - lse)
    }
    const double d = wave_sum_d((double)loss);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = d;
    __syncthreads();
    if (tid == 0) {
      double r = red[0];
      for (int i = 1; i < NT / 64; ++i) r += red[i];
      part[((long)lev * gridDim.y + b) * gridDim.x + blockIdx.x] = r;
    }
  }
}

// d loss_lev / d est_k = gscale_lev * -(c_k - softmax(est)_k * sum_j c_j);  gscale_lev = upstream * weight_lev / (B*HW)
template <int NT>
__global__ __launch_bounds__(NT) void focal_bwd_kernel(FocalArgs a, const float* __restrict__ gt,
                                                        const int* __restrict__ any_valid_p,
                                                        const float* __restrict__ gout, int K, long HW, float coef,
                                                        float inv_count) {
  extern __shared__ float col[];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long pix = (long)blockIdx.x * NT + tid;
  if (pix >= HW) return;
  const int any_valid = *any_valid_p;
  const Target t = make_target(gt[(long)b * HW + pix], K, any_valid);
  float csum = 0.f;
  for (int k = 0; k < K; ++k) {
    const float p = target_prob(t, k, any_valid);
    const float c = p * focal_weight(p, coef) * t.valid;
    col[k * NT + tid] = c;
    csum += c;
  }
  {
    const int lev = blockIdx.z;
    const float gs = gout[0] * a.weight[lev] * inv_count;
    const float* e = a.est[lev] + (long)b * K * HW + pix;
    float* g = a.gest[lev] + (long)b * K * HW + pix;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, e[(long)k * HW]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(e[(long)k * HW] - m);
    const float inv = 1.f / s;
    for (int k = 0; k < K; ++k) g[(long)k * HW] = -gs * (col[k * NT + tid] - expf(e[(long)k * HW] - m) * inv * csum);
  }
}

// any_valid = (count of pixels with 0 < gt < K) >= 1; one block (it is a flag, so order-free)
__global__ __launch_bounds__(256) void focal_any_valid_kernel(const float* __restrict__ gt, long total, int K,
                                                              int* __restrict__ flag) {
  int any = 0;
  for (long i = threadIdx.x; i < total; i += 256) any |= (gt[i] > 0.f && gt[i] < (float)K) ? 1 : 0;
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) *flag = any ? 1 : 0;
}

// out[lev] = mean loss of the level; out[nlev] = sum_lev weight_lev * out[lev]
__global__ __launch_bounds__(64) void focal_finalize_kernel(FocalArgs a, const double* __restrict__ part, int per_level,
                                                            double count, float* __restrict__ out) {
  double total = 0.0;
  for (int lev = 0; lev < a.nlev; ++lev) {
    double s = 0.0;
    for (int i = threadIdx.x; i < per_level; i += 64) s += part[(long)lev * per_level + i];
    s = wave_sum_d(s) / count;
    if (threadIdx.x == 0) out[lev] = (float)s;
    total += s * (double)a.weight[lev];
  }
  if (threadIdx.x == 0) out[a.nlev] = (float)total;
}

}  // namespace

extern "C" int dca_convex_up4_fwd(const float* mask_logits, const float* disp, float* up, int B, int h, int w,
                                  hipStream_t stream) {
  DCA_REQUIRE(mask_logits && disp && up && B > 0 && h > 0 && w > 0 && B <= 65535);
  DCA_REQUIRE((((uintptr_t)up) & 15) == 0);
  hipLaunchKernelGGL(convex_up4_fwd_kernel, dim3(cdiv((long)h * w, 256), B), dim3(256), 0, stream, mask_logits, disp, up,
                     h, w);
  return dca_launch_status();
}

extern "C" int dca_convex_up4_bwd(const float* mask_logits, const float* disp, const float* gup, float* glogits,
                                  float* gdisp, float* wk, int B, int h, int w, hipStream_t stream) {
  DCA_REQUIRE(mask_logits && disp && gup && glogits && gdisp && wk && B > 0 && h > 0 && w > 0 && B <= 65535);
  DCA_REQUIRE((((uintptr_t)gup) & 15) == 0);
  const dim3 grid(cdiv((long)h * w, 256), B);
  hipLaunchKernelGGL(convex_up4_bwd_kernel, grid, dim3(256), 0, stream, mask_logits, disp, gup, glogits, wk, h, w);
  hipLaunchKernelGGL(convex_up4_bwd_disp_kernel, grid, dim3(256), 0, stream, wk, gdisp, h, w);
  return dca_launch_status();
}

extern "C" long dca_focal_loss_workspace(int nlev, int B, long HW) {
  if (nlev <= 0 || B <= 0 || HW <= 0) return 0;
  return (long)nlev * B * ((HW + 63) / 64) + 1;   // doubles: per-block partials (+ one slot for the any-valid flag)
}

static int focal_args(FocalArgs& a, const float* const* ests, float* const* gests, const float* weights, int nlev) {
  if (!ests || !weights || nlev < 1 || nlev > FL_LMAX) return 0;
  a.nlev = nlev;
  for (int i = 0; i < FL_LMAX; ++i) {
    a.est[i] = i < nlev ? ests[i] : nullptr;
    a.gest[i] = (gests && i < nlev) ? gests[i] : nullptr;
    a.weight[i] = i < nlev ? weights[i] : 0.f;
    if (i < nlev && (!a.est[i] || (gests && !a.gest[i]))) return 0;
  }
  return 1;
}

extern "C" int dca_focal_loss_fwd(const float* const* ests, const float* weights, int nlev, const float* gt,
                                  double* work, float* out, int B, int K, long HW, float focal_coefficient,
                                  hipStream_t stream) {
  FocalArgs a;
  DCA_REQUIRE(focal_args(a, ests, nullptr, weights, nlev));
  DCA_REQUIRE(gt && work && out && B > 0 && B <= 65535 && K >= 2 && K <= FL_KMAX && HW > 0);
  const int nt = K <= 64 ? 256 : 64;
  const int nb = cdiv(HW, nt), per_level = B * nb;
  int* flag = (int*)(work + (long)nlev * per_level);
  hipLaunchKernelGGL(focal_any_valid_kernel, dim3(1), dim3(256), 0, stream, gt, (long)B * HW, K, flag);
  if (nt == 256)
    hipLaunchKernelGGL(focal_fwd_kernel<256>, dim3(nb, B, nlev), dim3(256), (size_t)K * 256 * sizeof(float), stream, a, gt,
                       flag, work, K, HW, focal_coefficient);
  else
    hipLaunchKernelGGL(focal_fwd_kernel<64>, dim3(nb, B, nlev), dim3(64), (size_t)K * 64 * sizeof(float), stream, a, gt, flag,
                       work, K, HW, focal_coefficient);
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, stream, a, work, per_level, (double)B * (double)HW,
                     out);
  return dca_launch_status();
}

extern "C" int dca_focal_loss_bwd(const float* const* ests, float* const* gests, const float* weights, int nlev,
                                  const float* gt, const double* work, const float* gloss, int B, int K, long HW,
                                  float focal_coefficient, hipStream_t stream) {
  FocalArgs a;
  DCA_REQUIRE(focal_args(a, ests, gests, weights, nlev));
  DCA_REQUIRE(gt && work && gloss && B > 0 && B <= 65535 && K >= 2 && K <= FL_KMAX && HW > 0);
  const int nt = K <= 64 ? 256 : 64;
  const int nb = cdiv(HW, nt), per_level = B * nb;
  const int* flag = (const int*)(work + (long)nlev * per_level);
  const float inv_count = (float)(1.0 / ((double)B * (double)HW));
  if (nt == 256)
    hipLaunchKernelGGL(focal_bwd_kernel<256>, dim3(nb, B, nlev), dim3(256), (size_t)K * 256 * sizeof(float), stream, a, gt,
                       flag, gloss, K, HW, focal_coefficient, inv_count);
  else
    hipLaunchKernelGGL(focal_bwd_kernel<64>, dim3(nb, B, nlev), dim3(64), (size_t)K * 64 * sizeof(float), stream, a, gt, flag,
                       gloss, K, HW, focal_coefficient, inv_count);
  return dca_launch_status();
}
