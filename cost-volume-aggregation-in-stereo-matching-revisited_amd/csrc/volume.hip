// Cost-volume builders and soft-argmin for gfx950.
//
// Replaces (reference): build_gwc_volume / groupwise_correlation (models/submodule.py:148-167),
// build_concat_volume (submodule.py:134-145), F.softmax(dim=1) + disparity_regression
// (submodule.py:127-131; call sites models/gwcnet_dca_g.py:237-239, :248-275).
//
// gwc forward: one workgroup per (batch, row y, group g).  The 2*CPG feature rows are staged in
// LDS once; every thread owns 4 consecutive x and walks the disparities with a sliding register
// window over the right features, so each output costs 2 LDS reads and the volume rows are written
// as 16-byte-per-lane coalesced stores (the volume write is the HBM-bound part: 40*D*H*W floats).
// The zero half-plane x < i is produced by the same kernel (no memset pass).
#include "dca_common.h"
#include "../../include/dca_hip.h"

template <int CPG>
__global__ __launch_bounds__(256) void gwc_fwd_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                      float* __restrict__ vol, int B, int C, int H, int W,
                                                      int D, int G, int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ls = smem;
  float* Rs = smem + CPG * W;
  const int y = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  for (int i = tid; i < CPG * W; i += 256) {
    const int c = i / W, x = i % W;
    const long src = (((long)b * C + g * CPG + c) * H + y) * W + x;
    Ls[i] = L[src];
    Rs[i] = R[src];
  }
  __syncthreads();
  constexpr int SEG = 12;
  const int nxq = (W + 3) / 4, nseg = (D + SEG - 1) / SEG;
  const float inv = 1.0f / (float)CPG;
  for (int item = tid; item < nxq * nseg; item += 256) {
    const int xq = item % nxq, seg = item / nxq, x0 = 4 * xq;
    const int ib = seg * SEG, ie = min(D, ib + SEG);
    float l[CPG][4], win[CPG][4];
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 + j, xr = x - ib;
        l[c][j] = (x < W) ? Ls[c * W + x] : 0.f;
        win[c][j] = (xr >= 0 && xr < W) ? Rs[c * W + xr] : 0.f;
      }
    for (int i = ib; i < ie; ++i) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CPG; ++c) s += l[c][j] * win[c][j];
        o[j] = s * inv;
      }
      float* dst = vol + ((((long)b * G + g) * D + i) * H + y) * W + x0;
      if (vec) {
        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x0 + j < W) dst[j] = o[j];
      }
      const int xn = x0 - i - 1;
#pragma unroll
      for (int c = 0; c < CPG; ++c) {
        win[c][3] = win[c][2];
        win[c][2] = win[c][1];
        win[c][1] = win[c][0];
        win[c][0] = (xn >= 0) ? Rs[c * W + xn] : 0.f;
      }
    }
  }
}

// gwc backward: dL[c,x] = 1/CPG sum_{i<=x} gV[i,x] R[c,x-i];
//                            dR[c,x'] = 1/CPG sum_{i<W-x'} gV[i,x'+i] L[c,x'+i].
template <int CPG>
__global__ __launch_bounds__(256) void gwc_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ L,
                                                      const float* __restrict__ R, float* __restrict__ gL,
                                                      float* __restrict__ gR, int B, int C, int H, int W, int D,
                                                      int G, int vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gv = smem;
  float* Ls = smem + D * W;
  float* Rs = Ls + CPG * W;
  const int y = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  if (vec) {   // W % 4 == 0, 16-byte aligned tensors: 16-byte loads, four in flight per thread
    const int WQ = W >> 2;
    const float* gbase = gvol + (((long)b * G + g) * D * H + y) * W;       // + d*H*W + x
    const long dstride = (long)H * W;
#pragma unroll 4
    for (int i = tid; i < D * WQ; i += 256) {
      const int d = i / WQ, q = i - d * WQ;
      *(float4*)(gv + d * W + 4 * q) = *(const float4*)(gbase + d * dstride + 4 * q);
    }
    const long cbase = (((long)b * C + g * CPG) * H + y) * W;
    for (int i = tid; i < CPG * WQ; i += 256) {
      const int c = i / WQ, q = i - c * WQ;
      *(float4*)(Ls + c * W + 4 * q) = *(const float4*)(L + cbase + c * dstride + 4 * q);
      *(float4*)(Rs + c * W + 4 * q) = *(const float4*)(R + cbase + c * dstride + 4 * q);
    }
  } else {
    for (int i = tid; i < D * W; i += 256) {
      const int d = i / W, x = i % W;
      gv[i] = gvol[((((long)b * G + g) * D + d) * H + y) * W + x];
    }
    for (int i = tid; i < CPG * W; i += 256) {
      const int c = i / W, x = i % W;
      const long src = (((long)b * C + g * CPG + c) * H + y) * W + x;
      Ls[i] = L[src];
      Rs[i] = R[src];
    }
  }
  __syncthreads();
  const float inv = 1.0f / (float)CPG;
  for (int item = tid; item < CPG * W; item += 256) {
    const int c = item / W, x = item % W;
    float gl = 0.f, gr = 0.f;
    const int nl = min(D - 1, x), nr = min(D - 1, W - 1 - x);
    for (int i = 0; i <= nl; ++i) gl += gv[i * W + x] * Rs[c * W + x - i];
    for (int i = 0; i <= nr; ++i) gr += gv[i * W + x + i] * Ls[c * W + x + i];
    const long dst = (((long)b * C + g * CPG + c) * H + y) * W + x;
    gL[dst] = gl * inv;
    gR[dst] = gr * inv;
  }
}

// The same for aligned rows (W % 4 == 0, W <= 256, D % 4 == 0, D >= 6 CPG, 16-byte aligned tensors), register blocked and
// persistent: a lane owns four consecutive x for ALL CPG channels of the group, the four waves split the disparities in blocks
// of four and meet in LDS at the end.  With x0 and the block's first disparity both multiples of four every operand of a block
// is an aligned 16-byte LDS read (rows zero padded by D + 4 floats on the side the shifted accesses run over): 44
// ds_read_b128 per 256 FMAs, where the kernel above issues two 4-byte reads per FMA.  A workgroup walks the rows y = blockIdx.x,
// + gridDim.x, ...: the next row's operands are fetched into registers (hardware-predicated buffer loads, all in flight at
// once) while the current row is being multiplied; 8 waves = (channel half, disparity share).  History at the batch-4 shape
// (tools/gwc_bwd_time.py): 0.95 ms (kernel above); register blocking alone 0.92 -- the staging loop of conditional loads
// waited for memory once per iteration and, with two workgroups per CU, WAS the kernel; all loads of a row in flight at once
// 0.62; persistent + prefetch with 4 waves: 294 registers, one workgroup per CU, 0.78; 8 waves 0.66 (the arithmetic then
// took 0.60 of it: LDS reads right before use); operand quads one channel ahead: 0.485 ms.
template <int CPG>
__global__ __launch_bounds__(512, 1) void gwc_bwd_blocked_kernel(const float* __restrict__ gvol, const float* __restrict__ L,
                                                              const float* __restrict__ R, float* __restrict__ gL,
                                                              float* __restrict__ gR, int B, int C, int H, int W, int D,
                                                              int G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PAD = D + 4, P = W + PAD;               // gv / L rows: [W data | PAD zeros]; R rows: [PAD zeros | W data]
  float* gv = smem;                                 // [D][P]
  float* Ls = smem + D * P;                         // [CPG][P]
  float* Rs = Ls + CPG * P;                         // [CPG][P]
  const int g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3, hc = tid >> 8;
  constexpr int HC = CPG > 1 ? CPG / 2 : 1;         // (CPG = 1 is never launched) waves 0-3 take channels 0 .. HC-1, waves 4-7 the others; wv = disparity share
  const int WQ = W >> 2, PADQ = PAD >> 2;
  const long dstride = (long)H * W;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < (D + CPG) * PADQ; i += 512) {      // the zero pads: written once, never touched again
    const int r = i / PADQ, q = i - r * PADQ;
    *(float4*)(gv + r * P + W + 4 * q) = z4;               // gv rows and, behind them, the L rows
  }
  for (int i = tid; i < CPG * PADQ; i += 512) {
    const int r = i / PADQ, q = i - r * PADQ;
    *(float4*)(Rs + r * P + 4 * q) = z4;
  }
  constexpr int KG = 6, KC = 1;                     // data quads per thread: D * W/4 <= 3072 of gv, CPG * W/4 <= 512 of L and of R
  float4 vg[KG], vl[KC], vr[KC];
  const long gplane = (((long)b * G + g) * D * H) * W, cplane = (((long)b * C + g * CPG) * H) * W;
  const __amdgpu_buffer_rsrc_t gr_ = dca_rsrc(gvol + gplane, (long)D * dstride * 4);
  const __amdgpu_buffer_rsrc_t lr_ = dca_rsrc(L + cplane, (long)CPG * dstride * 4);
  const __amdgpu_buffer_rsrc_t rr_ = dca_rsrc(R + cplane, (long)CPG * dstride * 4);
  auto fetch = [&](int y) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int i = tid + 512 * k, d = i / WQ, q = i - d * WQ;
      vg[k] = dca_bload4(gr_, (int)((d * dstride + (long)y * W + 4 * q) * 4), (int)(i < D * WQ));
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = tid + 512 * k, c = i / WQ, q = i - c * WQ;
      const int off = (int)((c * dstride + (long)y * W + 4 * q) * 4), ok = (int)(i < CPG * WQ);
      vl[k] = dca_bload4(lr_, off, ok);
      vr[k] = dca_bload4(rr_, off, ok);
    }
  };
  const int x0 = 4 * lane;
  const bool active = lane < WQ;
  const float inv = 1.0f / (float)CPG;
  if ((int)blockIdx.x < H) fetch(blockIdx.x);
  for (int y = blockIdx.x; y < H; y += gridDim.x) {
    __syncthreads();                                // the previous row's partial sums have been read
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int i = tid + 512 * k, d = i / WQ, q = i - d * WQ;
      if (i < D * WQ) *(float4*)(gv + d * P + 4 * q) = vg[k];
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = tid + 512 * k, c = i / WQ, q = i - c * WQ;
      if (i < CPG * WQ) {
        *(float4*)(Ls + c * P + 4 * q) = vl[k];
        *(float4*)(Rs + c * P + PAD + 4 * q) = vr[k];
      }
    }
    __syncthreads();
    if (y + (int)gridDim.x < H) fetch(y + gridDim.x);
    float gl[HC][4], gr[HC][4];
#pragma unroll
    for (int c = 0; c < HC; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) gl[c][j] = gr[c][j] = 0.f;
    if (active) {
      // this wave's blocks of four disparities.  With two waves per SIMD nothing but the code itself hides an LDS read's
      // latency, and left alone hipcc reads an operand right before its first use (36 reads x ~150 cycles per block: 7 us per
      // row where the FMAs need 1.5): the per-channel operand quads are requested one channel ahead (a full A / B operand
      // double buffer spilled 97 registers and was slower than no pipelining at all).
      for (int ib = 4 * wv; ib < D; ib += 16) {
        const int p = PAD + x0 - ib;                 // index of position x0 - ib in the padded R row (>= 4)
        const int q = x0 + ib;                       // positions q .. q+7 (< W + PAD)
        float4 gq[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) gq[s] = *(const float4*)(gv + (ib + s) * P + x0);
        float4 lo = *(const float4*)(Rs + (hc * HC) * P + p - 4), hi = *(const float4*)(Rs + (hc * HC) * P + p);
        // dL[c][x0+j] += gv[ib+s][x0+j] R[c][x0+j-ib-s]: (lo, hi) = positions (x0 - ib) - 4 .. + 3
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          float4 nlo = lo, nhi = hi;
          if (c + 1 < HC) {
            nlo = *(const float4*)(Rs + (hc * HC + c + 1) * P + p - 4);
            nhi = *(const float4*)(Rs + (hc * HC + c + 1) * P + p);
          }
          const float rv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float gs[4] = {gq[s].x, gq[s].y, gq[s].z, gq[s].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) gl[c][j] = fmaf(gs[j], rv[4 + j - s], gl[c][j]);
          }
          lo = nlo; hi = nhi;
          __builtin_amdgcn_sched_barrier(0);
        }
        // dR[c][x0+j] += gv[ib+s][x0+j+ib+s] L[c][x0+j+ib+s]: positions (x0 + ib) .. + 7 of the four gv rows and of L
        float4 bg[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bg[s][0] = *(const float4*)(gv + (ib + s) * P + q);
          bg[s][1] = *(const float4*)(gv + (ib + s) * P + q + 4);
        }
        lo = *(const float4*)(Ls + (hc * HC) * P + q); hi = *(const float4*)(Ls + (hc * HC) * P + q + 4);
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          float4 nlo = lo, nhi = hi;
          if (c + 1 < HC) {
            nlo = *(const float4*)(Ls + (hc * HC + c + 1) * P + q);
            nhi = *(const float4*)(Ls + (hc * HC + c + 1) * P + q + 4);
          }
          const float lv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float gd[8] = {bg[s][0].x, bg[s][0].y, bg[s][0].z, bg[s][0].w, bg[s][1].x, bg[s][1].y, bg[s][1].z, bg[s][1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) gr[c][j] = fmaf(gd[s + j], lv[s + j], gr[c][j]);
          }
          lo = nlo; hi = nhi;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the four waves' partial sums meet in LDS, in wave order: row (wave - 1, dL | dR, c) of the gv image, data part only
    // (the pads stay zero); D >= 6 CPG rows
    __syncthreads();
    if (wv > 0 && active) {
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        *(float4*)(gv + (((wv - 1) * 2 + 0) * CPG + hc * HC + c) * P + x0) = make_float4(gl[c][0], gl[c][1], gl[c][2], gl[c][3]);
        *(float4*)(gv + (((wv - 1) * 2 + 1) * CPG + hc * HC + c) * P + x0) = make_float4(gr[c][0], gr[c][1], gr[c][2], gr[c][3]);
      }
    }
    __syncthreads();
    if (wv == 0 && active) {
      const long cbase = cplane + (long)y * W;
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        float4 sl = make_float4(gl[c][0], gl[c][1], gl[c][2], gl[c][3]), sr = make_float4(gr[c][0], gr[c][1], gr[c][2], gr[c][3]);
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 pl = *(const float4*)(gv + ((w * 2 + 0) * CPG + hc * HC + c) * P + x0);
          const float4 pr = *(const float4*)(gv + ((w * 2 + 1) * CPG + hc * HC + c) * P + x0);
          sl.x += pl.x; sl.y += pl.y; sl.z += pl.z; sl.w += pl.w;
          sr.x += pr.x; sr.y += pr.y; sr.z += pr.z; sr.w += pr.w;
        }
        const long dst = cbase + (hc * HC + c) * dstride + x0;
        *(float4*)(gL + dst) = make_float4(sl.x * inv, sl.y * inv, sl.z * inv, sl.w * inv);
        *(float4*)(gR + dst) = make_float4(sr.x * inv, sr.y * inv, sr.z * inv, sr.w * inv);
      }
    }
  }
}

__global__ void concat_fwd_kernel(const float* __restrict__ L, const float* __restrict__ R, float* __restrict__ vol,
                                  int B, int C, int H, int W, int D) {
  const long total = (long)B * 2 * C * D * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long t = idx / W;
    const int y = t % H; t /= H;
    const int i = t % D; t /= D;
    const int c2 = t % (2 * C);
    const int b = t / (2 * C);
    float v = 0.f;
    if (x >= i) {
      if (c2 < C) v = L[(((long)b * C + c2) * H + y) * W + x];
      else v = R[(((long)b * C + c2 - C) * H + y) * W + x - i];
    }
    vol[idx] = v;
  }
}

__global__ void concat_bwd_kernel(const float* __restrict__ gvol, float* __restrict__ gL, float* __restrict__ gR,
                                  int B, int C, int H, int W, int D) {
  const long total = (long)B * C * H * W;
  const long HW = (long)H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long t = idx / W;
    const int y = t % H; t /= H;
    const int c = t % C;
    const int b = t / C;
    const float* pl = gvol + ((long)b * 2 * C + c) * D * HW + (long)y * W + x;
    const float* pr = gvol + ((long)b * 2 * C + C + c) * D * HW + (long)y * W + x;
    float gl = 0.f, gr = 0.f;
    const int nl = min(D - 1, x), nr = min(D - 1, W - 1 - x);
    for (int i = 0; i <= nl; ++i) gl += pl[i * HW];
    for (int i = 0; i <= nr; ++i) gr += pr[i * HW + i];
    gL[idx] = gl;
    gR[idx] = gr;
  }
}

// ---------------------------------------------------------------------------------------------
// softmax over dim 1 of (B, K, HW) and soft-argmin; one thread per (b, pixel), coalesced over HW.
// mode 0: p = softmax(x)                      mode 1: disp = sum_k k * softmax(x)_k
// mode 2: disp = sum_k k * x_k  (plain disparity_regression on an arbitrary x)
// ---------------------------------------------------------------------------------------------
__global__ void softargmin_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int K, long HW,
                                      int mode) {
  const long total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx % HW;
    const float* xp = x + b * K * HW + p;
    if (mode == 2) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += xp[k * HW] * (float)k;
      out[idx] = s;
      continue;
    }
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, xp[k * HW]);
    float s = 0.f, sk = 0.f;
    for (int k = 0; k < K; ++k) {
      const float e = expf(xp[k * HW] - m);
      s += e;
      sk += e * (float)k;
    }
    if (mode == 1) {
      out[idx] = sk / s;
    } else {
      const float inv = 1.0f / s;
      float* op = out + b * K * HW + p;
      for (int k = 0; k < K; ++k) op[k * HW] = expf(xp[k * HW] - m) * inv;
    }
  }
}

// mode 0: gx_k = p_k (gp_k - sum_j gp_j p_j)   (aux = p, g = gp (B,K,HW))
// mode 1: gx_k = p_k (k - disp) g              (aux = logits x, g = gdisp (B,HW)); p recomputed
// mode 2: gx_k = k * g
__global__ void softargmin_bwd_kernel(const float* __restrict__ aux, const float* __restrict__ g,
                                      float* __restrict__ gx, int B, int K, long HW, int mode) {
  const long total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / HW, p = idx % HW;
    const float* ap = aux + b * K * HW + p;
    float* gxp = gx + b * K * HW + p;
    if (mode == 0) {
      const float* gp = g + b * K * HW + p;
      float dot = 0.f;
      for (int k = 0; k < K; ++k) dot += gp[k * HW] * ap[k * HW];
      for (int k = 0; k < K; ++k) gxp[k * HW] = ap[k * HW] * (gp[k * HW] - dot);
    } else if (mode == 1) {
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, ap[k * HW]);
      float s = 0.f, sk = 0.f;
      for (int k = 0; k < K; ++k) {
        const float e = expf(ap[k * HW] - m);
        s += e;
        sk += e * (float)k;
      }
      const float inv = 1.0f / s, disp = sk * inv, gg = g[idx];
      for (int k = 0; k < K; ++k) gxp[k * HW] = expf(ap[k * HW] - m) * inv * ((float)k - disp) * gg;
    } else {
      const float gg = g[idx];
      for (int k = 0; k < K; ++k) gxp[k * HW] = (float)k * gg;
    }
  }
}

static int ew_grid(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 4096 ? (g > 0 ? g : 1) : 4096);
}

template <int CPG>
static int gwc_fwd_launch(const float* L, const float* R, float* vol, int B, int C, int H, int W, int D, int G,
                          hipStream_t s) {
  const size_t lds = (size_t)2 * CPG * W * 4;
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)gwc_fwd_kernel<CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int vec = (W % 4 == 0) && (((uintptr_t)vol & 15) == 0);
  hipLaunchKernelGGL(gwc_fwd_kernel<CPG>, dim3(H, G, B), dim3(256), lds, s, L, R, vol, B, C, H, W, D, G, vec);
  return dca_launch_status();
}
template <int CPG>
static int gwc_bwd_launch(const float* gvol, const float* L, const float* R, float* gL, float* gR, int B, int C,
                          int H, int W, int D, int G, hipStream_t s) {
  const size_t lds = (size_t)(D + 2 * CPG) * W * 4;
  if (lds > 64 * 1024)
    hipFuncSetAttribute((const void*)gwc_bwd_kernel<CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int vec = (W % 4 == 0) && ((((uintptr_t)gvol | (uintptr_t)L | (uintptr_t)R) & 15) == 0);
  const size_t blds = (size_t)(D + 2 * CPG) * (W + D + 4) * 4;      // padded rows (the partial sums reuse 6 CPG of the D gv rows)
  if (CPG <= 8 && CPG % 2 == 0 && vec && W <= 256 && D % 4 == 0 && D >= 6 * CPG && D * (W / 4) <= 3072 && CPG * (W / 4) <= 512 &&
      ((((uintptr_t)gL | (uintptr_t)gR) & 15) == 0) && blds <= 80 * 1024 && (long)D * H * W * 4 < 0x7ffffff0L &&
      (long)CPG * H * W * 4 < 0x7ffffff0L) {
    hipFuncSetAttribute((const void*)gwc_bwd_blocked_kernel<CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
    // persistent over the rows: ~1280 workgroups of 8 waves = five rounds of one per CU
    int gx = (1280 + G * B - 1) / (G * B);
    if (gx > H) gx = H;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gwc_bwd_blocked_kernel<CPG>, dim3(gx, G, B), dim3(512), blds, s, gvol, L, R, gL, gR, B, C, H, W, D, G);
    return dca_launch_status();
  }
  hipLaunchKernelGGL(gwc_bwd_kernel<CPG>, dim3(H, G, B), dim3(256), lds, s, gvol, L, R, gL, gR, B, C, H, W, D, G, vec);
  return dca_launch_status();
}

extern "C" int dca_gwc_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W,
                                  int maxdisp, int num_groups, hipStream_t stream) {
  DCA_REQUIRE(ref && tgt && vol && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0 && num_groups > 0);
  DCA_REQUIRE(C % num_groups == 0 && H <= 65535 && num_groups <= 65535 && B <= 65535);
  DCA_REQUIRE((size_t)2 * (C / num_groups) * W * 4 <= 160 * 1024);
  switch (C / num_groups) {
    case 1: return gwc_fwd_launch<1>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 2: return gwc_fwd_launch<2>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 4: return gwc_fwd_launch<4>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 8: return gwc_fwd_launch<8>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    case 16: return gwc_fwd_launch<16>(ref, tgt, vol, B, C, H, W, maxdisp, num_groups, stream);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int dca_gwc_volume_bwd(const float* gvol, const float* ref, const float* tgt, float* gref, float* gtgt,
                                  int B, int C, int H, int W, int maxdisp, int num_groups, hipStream_t stream) {
  DCA_REQUIRE(gvol && ref && tgt && gref && gtgt && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0);
  DCA_REQUIRE(num_groups > 0 && C % num_groups == 0 && H <= 65535 && num_groups <= 65535 && B <= 65535);
  DCA_REQUIRE((size_t)(maxdisp + 2 * (C / num_groups)) * W * 4 <= 160 * 1024);
  switch (C / num_groups) {
    case 1: return gwc_bwd_launch<1>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 2: return gwc_bwd_launch<2>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 4: return gwc_bwd_launch<4>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 8: return gwc_bwd_launch<8>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    case 16: return gwc_bwd_launch<16>(gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, num_groups, stream);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int dca_concat_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W,
                                     int maxdisp, hipStream_t stream) {
  DCA_REQUIRE(ref && tgt && vol && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0);
  const long total = (long)B * 2 * C * maxdisp * H * W;
  hipLaunchKernelGGL(concat_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, ref, tgt, vol, B, C, H, W, maxdisp);
  return dca_launch_status();
}

extern "C" int dca_concat_volume_bwd(const float* gvol, float* gref, float* gtgt, int B, int C, int H, int W,
                                     int maxdisp, hipStream_t stream) {
  DCA_REQUIRE(gvol && gref && gtgt && B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0);
  hipLaunchKernelGGL(concat_bwd_kernel, dim3(ew_grid((long)B * C * H * W)), dim3(256), 0, stream, gvol, gref, gtgt, B,
                     C, H, W, maxdisp);
  return dca_launch_status();
}

extern "C" int dca_softargmin_fwd(const float* x, float* out, int B, int K, long HW, int mode, hipStream_t stream) {
  DCA_REQUIRE(x && out && B > 0 && K > 0 && HW > 0 && mode >= 0 && mode <= 2);
  hipLaunchKernelGGL(softargmin_fwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, x, out, B, K, HW, mode);
  return dca_launch_status();
}

extern "C" int dca_softargmin_bwd(const float* aux, const float* g, float* gx, int B, int K, long HW, int mode,
                                  hipStream_t stream) {
  DCA_REQUIRE(aux && g && gx && B > 0 && K > 0 && HW > 0 && mode >= 0 && mode <= 2);
  hipLaunchKernelGGL(softargmin_bwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, aux, g, gx, B, K, HW,
                     mode);
  return dca_launch_status();
}
