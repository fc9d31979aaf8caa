// One launch that re-lays-out MANY convolution weights (both the fp32 wt[tap][cin][cout] images of conv3d_mfma.hip and
// the pre-split bf16x3 fragment images of conv3d_bf16x3.hip): a training step re-packs every weight twice (forward and
// backward-data layouts) right after the optimizer update, which as ~130 separate 5-us launches costs more GPU time
// than the work itself.  The descriptors live in a device table built once (ops.PrepackPlan).
#include "dca_common.h"

namespace {

struct PrepDesc {          // mirrored by ops.PrepackPlan (72 bytes)
  const float* src;
  void* dst;
  int kind;                // 0: fp32 image (dca_conv3d_prep_weight), 1: bf16x3 image (dca_conv3d_x3_prep_weight),
                           // 2: bf16x3 fragments of a 1x1x1 conv (dca_conv1_x3_prep_weight)
                           // (the f16x2 images of dca_conv3d_x2_prep_weight depend on the operand's per-channel exponents and
                           // are packed per launch: not part of a plan)
  int A, Bn, Apad, Bpad, K, src_ab, flip, Btotal, b_off, NCH, pad_;
  long total;              // elements of dst
};

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  const float r1 = v - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}


__global__ __launch_bounds__(256) void prep_many_kernel(const PrepDesc* __restrict__ table) {
  const PrepDesc d = table[blockIdx.y];
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < d.total; idx += (long)gridDim.x * 256) {
    if (d.kind == 0) {
      const int ab = d.Apad * d.Bpad;
      const int tap = (int)(idx / ab), ai = (int)((idx / d.Bpad) % d.Apad), bi = (int)(idx % d.Bpad);
      float v = 0.f;
      if (ai < d.A && bi < d.Bn) {
        const int st = d.flip ? d.K - 1 - tap : tap;
        v = d.src_ab ? d.src[((long)ai * d.Btotal + d.b_off + bi) * d.K + st]
                     : d.src[((long)(d.b_off + bi) * d.A + ai) * d.K + st];
      }
      ((float*)d.dst)[idx] = v;
    } else if (d.kind == 2) {
      const int j = idx & 7, lane = (idx >> 3) & 63, term = (int)((idx >> 9) % 3), chunk = (int)((idx >> 9) / 3);
      const int bi = lane & 31, ai = chunk * 16 + 8 * (lane >> 5) + j;
      float v = 0.f;
      if (ai < d.A && bi < d.Bn)
        v = d.src_ab ? d.src[(long)ai * d.Btotal + d.b_off + bi] : d.src[(long)(d.b_off + bi) * d.A + ai];
      __bf16 h, m, l;
      split3(v, h, m, l);
      const __bf16 o = term == 0 ? h : (term == 1 ? m : l);
      ((unsigned short*)d.dst)[idx] = __builtin_bit_cast(unsigned short, o);
    } else {
      const int j = idx & 7, lane = (idx >> 3) & 63;
      long t = idx >> 9;
      const int term = t % 3; t /= 3;
      const int tap = t % 27; t /= 27;
      const int chunk = t % d.NCH;
      const int cblk = (int)(t / d.NCH);
      const int bi = cblk * 32 + (lane & 31), ai = chunk * 16 + 8 * (lane >> 5) + j;
      float v = 0.f;
      if (ai < d.A && bi < d.Bn) {
        const int st = d.flip ? 26 - tap : tap;
        v = d.src_ab ? d.src[((long)ai * d.Bn + bi) * 27 + st] : d.src[((long)bi * d.A + ai) * 27 + st];
      }
      __bf16 h, m, l;
      split3(v, h, m, l);
      const __bf16 o = term == 0 ? h : (term == 1 ? m : l);
      ((unsigned short*)d.dst)[idx] = __builtin_bit_cast(unsigned short, o);
    }
  }
}

}  // namespace

// table: n device-resident 72-byte descriptors {src, dst, kind, A, Bn, Apad, Bpad, K, src_ab, flip, Btotal, b_off, NCH,
// pad, total} (pointers 8 bytes, ints 4, total 8) with the argument meaning of dca_conv3d_prep_weight (kind 0) /
// dca_conv3d_x3_prep_weight (kind 1: A, Bn, src_ab, flip, NCH = ceil(A/16), total = weight_bytes/2) /
// dca_conv1_x3_prep_weight (kind 2: A, Bn, src_ab, Btotal, b_off, total = weight_bytes/2).
extern "C" int dca_conv3d_prep_many(const void* table, int n, hipStream_t stream) {
  DCA_REQUIRE(table && n > 0 && n <= 65535 && (((uintptr_t)table) & 7) == 0);
  hipLaunchKernelGGL(prep_many_kernel, dim3(48, n), dim3(256), 0, stream, (const PrepDesc*)table);
  return dca_launch_status();
}
