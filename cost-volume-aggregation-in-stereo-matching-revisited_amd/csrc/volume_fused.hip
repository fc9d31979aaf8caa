// Fused cost-volume builder: the group-wise correlation volume and (GC variant) the concat volume written straight into
// ONE (B, G + 2*Cc, D, H, W) tensor, in fp32 or in the reduced-precision storage type, from the 2D extractor's feature
// maps as they are -- the three maps l2 / l3 / l4 that the reference concatenates into `gwc_feature` are read in place.
//
// Replaces (reference):  models/gwcnet_dca_g.py:60     gwc_feature = torch.cat((l2, l3, l4), dim=1)
//                        models/submodule.py:157-167   build_gwc_volume (+ groupwise_correlation :148-154)
//                        models/submodule.py:134-145   build_concat_volume
//                        models/gwcnet_dca_g.py:217-220 volume = torch.cat((gwc_volume, concat_volume), 1)
// i.e. two full-volume copies (the two cats) and the zero-fill + 48 slice assignments per builder.
//
// gwc kernel: one workgroup per (batch, row y, group).  The group's 2 x CPG feature rows are staged in LDS with 16-byte
// loads, the right rows behind a zero pad of PAD >= D floats so that x - i < 0 reads zeros (the volume's zero
// half-plane falls out of the arithmetic).  A thread owns 4 consecutive x and walks the disparities four at a time: per
// channel two ds_read_b128 fetch the 8 right-feature values R[x0-i0-4 .. x0-i0+3] that the 4 x 4 (disparity, x) outputs
// need, so an output costs CPG FMAs plus 1/8 LDS read -- the kernel sits on the volume's HBM write, not on the VALU
// (the round-1 kernel spent ~19 instructions per output on a sliding register window and 4-byte LDS reads).
#include "dca_common.h"
#include "../../include/dca_hip.h"

namespace {

struct VolArgs {
  const float* L[3];    // channel segments of the left / right correlation features (gwc_feature = cat of them)
  const float* R[3];
  int segC[3];
  int nseg;
  const float* cL;      // concat features (B, Cc, H, W) or null
  const float* cR;
  int Cc;
  void* vol;            // (B, Gtot, D, H, W), Gtot = G + 2*Cc
  int B, C, H, W, D, G, Gtot;
  unsigned* vmax;       // optional (fp32 volume): per-channel slots [g][b * H + y] <- max |volume| of this workgroup's row
};

template <typename OT> struct Out;
// The volume is written once and read much later by another kernel: non-temporal stores keep it from displacing the
// feature rows in L2 -- measured 67.5 -> 57.4 us for the 251 MB fp32 volume at 544x960 / D=192 (4.95 -> 5.82 TB/s).
// (The 2-byte volume, 8-byte stores per lane, measures 58 us with plain stores and 61-64 with non-temporal ones; packing
// lane pairs into 16-byte stores interleaves two rows per store instruction and was slower still, 88 us: left plain.)
#ifndef VF_NT
#define VF_NT 1
#endif
typedef float vf_f32x4 __attribute__((ext_vector_type(4)));
template <> struct Out<float> {
  static __device__ __forceinline__ void store4(float* p, const float (&o)[4]) {
    const vf_f32x4 v = {o[0], o[1], o[2], o[3]};
    if (VF_NT) __builtin_nontemporal_store(v, (vf_f32x4*)p);
    else *(vf_f32x4*)p = v;
  }
};
template <typename MT> __device__ __forceinline__ unsigned vf_pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef MT mtx2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mtx2));
}
typedef unsigned vf_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned vf_u32x4 __attribute__((ext_vector_type(4)));
template <> struct Out<__bf16> {
  static __device__ __forceinline__ void store4(__bf16* p, const float (&o)[4]) {
    *(vf_u32x2*)p = vf_u32x2{vf_pack2<__bf16>(o[0], o[1]), vf_pack2<__bf16>(o[2], o[3])};
  }
};
template <> struct Out<_Float16> {
  static __device__ __forceinline__ void store4(_Float16* p, const float (&o)[4]) {
    *(vf_u32x2*)p = vf_u32x2{vf_pack2<_Float16>(o[0], o[1]), vf_pack2<_Float16>(o[2], o[3])};
  }
};

// requires W % 4 == 0, D % 4 == 0, 16-byte aligned feature maps and volume
template <int CPG, typename OT>
__global__ __launch_bounds__(256) void gwc_fused_kernel(VolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int W = a.W, D = a.D, PAD = (D + 3) & ~3, RW = W + PAD;
  float* Ls = smem;               // [CPG][W]
  float* Rs = smem + CPG * W;     // [CPG][PAD + W], the first PAD floats of a row are zero
  const int y = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  // the group's channels [g*CPG, (g+1)*CPG) lie inside one segment (segment widths are multiples of CPG)
  int seg = 0, c0 = g * CPG;
  while (seg + 1 < a.nseg && c0 >= a.segC[seg]) { c0 -= a.segC[seg]; ++seg; }
  const long HW = (long)a.H * W;
  const float* Lp = a.L[seg] + ((long)b * a.segC[seg] + c0) * HW + (long)y * W;
  const float* Rp = a.R[seg] + ((long)b * a.segC[seg] + c0) * HW + (long)y * W;
  const int WQ = W >> 2, PQ = PAD >> 2;
  for (int i = tid; i < CPG * WQ; i += 256) {
    const int c = i / WQ, q = i - c * WQ;
    *(float4*)(Ls + c * W + 4 * q) = *(const float4*)(Lp + c * HW + 4 * q);
    *(float4*)(Rs + c * RW + PAD + 4 * q) = *(const float4*)(Rp + c * HW + 4 * q);
  }
  for (int i = tid; i < CPG * PQ; i += 256) {
    const int c = i / PQ, q = i - c * PQ;
    *(float4*)(Rs + c * RW + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const float inv = 1.0f / (float)CPG;
  const int DQ = D >> 2, rows = 256 / WQ > 0 ? 256 / WQ : 1;   // disparity quads handled concurrently
  const int xq = tid % WQ, iq0 = tid / WQ;
  const bool idle = tid >= rows * WQ && WQ <= 256;
  OT* vbase = (OT*)a.vol + (((long)b * a.Gtot + g) * D) * HW + (long)y * W;
  float vm = 0.f;
  for (int xqq = idle ? WQ : xq; xqq < WQ; xqq += (WQ <= 256 ? WQ : 256)) {   // (WQ > 256: threads stride over the row)
    const int x0 = 4 * xqq;
    float l[CPG][4];
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      const float4 v = *(const float4*)(Ls + c * W + x0);
      l[c][0] = v.x; l[c][1] = v.y; l[c][2] = v.z; l[c][3] = v.w;
    }
    for (int iq = (WQ <= 256 ? iq0 : 0); iq < DQ; iq += (WQ <= 256 ? rows : 1)) {
      const int i0 = 4 * iq;
      float o[4][4];
#pragma unroll
      for (int di = 0; di < 4; ++di)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[di][j] = 0.f;
#pragma unroll
      for (int c = 0; c < CPG; ++c) {
        // r[k] = R[x0 - i0 - 4 + k]; output (di, j) needs R[x0 + j - i0 - di] = r[4 + j - di]
        const float* rp = Rs + c * RW + PAD + x0 - i0 - 4;
        const float4 ra = *(const float4*)rp, rb = *(const float4*)(rp + 4);
        const float r[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int di = 0; di < 4; ++di)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[di][j] += l[c][j] * r[4 + j - di];
      }
#pragma unroll
      for (int di = 0; di < 4; ++di) {
        const float v[4] = {o[di][0] * inv, o[di][1] * inv, o[di][2] * inv, o[di][3] * inv};
        vm = fmaxf(fmaxf(vm, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        Out<OT>::store4(vbase + (long)(i0 + di) * HW + x0, v);
      }
    }
  }
  if (a.vmax) dca_cmax_put(vm, a.vmax + (long)g * DCA_AMAX_CSLOTS + b * a.H + y);   // every thread of the workgroup arrives here
}

// concat part: channel c < Cc: L[c][x] for x >= i; channel Cc + c: R[c][x - i] for x >= i; zero for x < i.
// One thread per (b, channel, disparity, y, x quad), quads along W fastest.
template <typename OT>
__global__ __launch_bounds__(256) void concat_fused_kernel(VolArgs a) {
  const int W = a.W, WQ = W >> 2, C2 = 2 * a.Cc;
  const long HW = (long)a.H * W;
  const long total = (long)a.B * C2 * a.D * a.H * WQ;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int q = (int)(idx % WQ);
    long t = idx / WQ;
    const int y = (int)(t % a.H); t /= a.H;
    const int i = (int)(t % a.D); t /= a.D;
    const int c2 = (int)(t % C2);
    const int b = (int)(t / C2);
    const int x0 = 4 * q;
    float v[4];
    if (c2 < a.Cc) {
      const float4 s = *(const float4*)(a.cL + ((long)b * a.Cc + c2) * HW + (long)y * W + x0);
      v[0] = x0 + 0 >= i ? s.x : 0.f; v[1] = x0 + 1 >= i ? s.y : 0.f;
      v[2] = x0 + 2 >= i ? s.z : 0.f; v[3] = x0 + 3 >= i ? s.w : 0.f;
    } else {
      const float* s = a.cR + ((long)b * a.Cc + c2 - a.Cc) * HW + (long)y * W;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = x0 + j >= i ? s[x0 + j - i] : 0.f;
    }
    Out<OT>::store4((OT*)a.vol + ((((long)b * a.Gtot + a.G + c2) * a.D + i) * HW) + (long)y * W + x0, v);
  }
}

template <int CPG, typename OT>
int launch_gwc(const VolArgs& a, hipStream_t s) {
  const int PAD = (a.D + 3) & ~3;
  const size_t lds = (size_t)CPG * (2 * a.W + PAD) * 4;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gwc_fused_kernel<CPG, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gwc_fused_kernel<CPG, OT>), dim3(a.H, a.G, a.B), dim3(256), lds, s, a);
  return dca_launch_status();
}

template <typename OT>
int launch_all(const VolArgs& a, hipStream_t s) {
  int rc;
  switch (a.C / a.G) {
    case 1: rc = launch_gwc<1, OT>(a, s); break;
    case 2: rc = launch_gwc<2, OT>(a, s); break;
    case 4: rc = launch_gwc<4, OT>(a, s); break;
    case 8: rc = launch_gwc<8, OT>(a, s); break;
    case 16: rc = launch_gwc<16, OT>(a, s); break;
    default: return (int)hipErrorInvalidValue;
  }
  if (rc != 0 || a.Cc == 0) return rc;
  const long total = (long)a.B * 2 * a.Cc * a.D * a.H * (a.W / 4);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL((concat_fused_kernel<OT>), dim3((int)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, a);
  return dca_launch_status();
}

}  // namespace

extern "C" int dca_cost_volume_fwd(const float* const* refs, const float* const* tgts, const int* seg_channels, int nseg,
                                   const float* cref, const float* ctgt, int Cc, void* vol, int B, int H, int W,
                                   int maxdisp, int num_groups, int dtype, unsigned* vmax, hipStream_t stream) {
  DCA_REQUIRE(refs && tgts && seg_channels && nseg >= 1 && nseg <= 3 && vol);
  DCA_REQUIRE(vmax == nullptr || (dtype == 0 && Cc == 0 && (long)B * H <= DCA_AMAX_CSLOTS));
  DCA_REQUIRE(B > 0 && H > 0 && W > 0 && maxdisp > 0 && num_groups > 0 && Cc >= 0);
  DCA_REQUIRE((Cc == 0) == (cref == nullptr) && (Cc == 0) == (ctgt == nullptr));
  DCA_REQUIRE(dtype == 0 || dtype == DCA_BF16 || dtype == DCA_FP16);
  DCA_REQUIRE(W % 4 == 0 && maxdisp % 4 == 0 && H <= 65535 && num_groups <= 65535 && B <= 65535);
  VolArgs a;
  a.C = 0;
  for (int i = 0; i < 3; ++i) {
    a.L[i] = i < nseg ? refs[i] : nullptr;
    a.R[i] = i < nseg ? tgts[i] : nullptr;
    a.segC[i] = i < nseg ? seg_channels[i] : 0;
    if (i < nseg) {
      DCA_REQUIRE(a.L[i] && a.R[i] && a.segC[i] > 0 && (((uintptr_t)a.L[i] | (uintptr_t)a.R[i]) & 15) == 0);
      a.C += a.segC[i];
    }
  }
  DCA_REQUIRE(a.C % num_groups == 0);
  const int cpg = a.C / num_groups;
  for (int i = 0; i < nseg; ++i) DCA_REQUIRE(a.segC[i] % cpg == 0);   // no group straddles two segments
  DCA_REQUIRE((((uintptr_t)vol | (uintptr_t)cref | (uintptr_t)ctgt) & 15) == 0);
  a.nseg = nseg; a.cL = cref; a.cR = ctgt; a.Cc = Cc; a.vol = vol; a.vmax = vmax;
  a.B = B; a.H = H; a.W = W; a.D = maxdisp; a.G = num_groups; a.Gtot = num_groups + 2 * Cc;
  if (dtype == 0) return launch_all<float>(a, stream);
  if (dtype == DCA_BF16) return launch_all<__bf16>(a, stream);
  return launch_all<_Float16>(a, stream);
}
