// Root-cause experiment for the round-2 hipGraph replay bug: a small device word that
// kernel A produces and kernel B consumes inside one captured graph, replayed several times with DIFFERENT data per replay.
//
//     graph = [zero fill of the words] -> [producer: max |x| into the words] -> [consumer: every workgroup records what it read]
//
// Variants: zero fill by {memset node, kernel}; producer by {atomicMax (what `atomicMax` compiles to on gfx950:
// global_atomic_umax without sc1 -- executed at the memory side, the line is not kept in the issuing XCD's L2),
// plain store into a slot of its own}; consumer load by {plain vector load, sc1 ("agent-coherent") vector load, scalar
// s_load through the scalar data cache}.  Each consumer workgroup also records the XCD it ran on.
//
//     hipcc --offload-arch=gfx950 -O3 tools/graph_amax_repro.hip -o gpurun_out/graph_amax_repro && gpurun_out/graph_amax_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int SLOTS = 64, NCONS = 256;

__global__ void zero_kernel(unsigned* w) { w[threadIdx.x] = 0u; }

template <int PROD>   // 0: atomicMax into slot (block & 63), 1: plain store into the block's own slot
__global__ void produce_kernel(const float* __restrict__ x, long n, unsigned* w) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (PROD == 0) atomicMax(w + (blockIdx.x & (SLOTS - 1)), __float_as_uint(m));
    else w[blockIdx.x & (SLOTS - 1)] = __float_as_uint(m);      // grid == SLOTS: one writer per slot
  }
}

template <int CONS>   // 0: plain vector load, 1: sc1 vector load, 2: scalar load of slot 0..63 (uniform addresses)
__global__ void consume_kernel(const unsigned* w, unsigned* seen, unsigned* xcc) {
  unsigned v = 0;
  if (CONS == 0) v = w[threadIdx.x & (SLOTS - 1)];
  else if (CONS == 1) v = __hip_atomic_load(w + (threadIdx.x & (SLOTS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) { const unsigned u = w[i]; v = v > u ? v : u; }   // uniform -> s_load
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned u = (unsigned)__shfl_xor((int)v, o, 64); v = v > u ? v : u; }
  if (threadIdx.x == 0) {
    seen[blockIdx.x] = v;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 15;
  }
}

int main() {
  const long n = 1 << 22;
  float* x; unsigned *w, *seen, *xcc;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&w, SLOTS * 4)); CK(hipMalloc(&seen, NCONS * 4)); CK(hipMalloc(&xcc, NCONS * 4));
  std::vector<float> hx(n);
  hipStream_t s; CK(hipStreamCreate(&s));
  const char* zn[2] = {"memset node", "zero kernel"};
  const char* pn[2] = {"atomicMax (global_atomic_umax, no sc1)", "plain store, slot per workgroup"};
  const char* cn[3] = {"plain vector load", "sc1 vector load", "scalar s_load"};
  int bad_total = 0;
  for (int z = 0; z < 2; ++z) for (int p = 0; p < 2; ++p) for (int c = 0; c < 3; ++c) {
    // capture
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    if (z == 0) CK(hipMemsetAsync(w, 0, SLOTS * 4, s)); else hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(SLOTS), 0, s, w);
    if (p == 0) hipLaunchKernelGGL(produce_kernel<0>, dim3(2048), dim3(256), 0, s, x, n, w);
    else hipLaunchKernelGGL(produce_kernel<1>, dim3(SLOTS), dim3(256), 0, s, x, n, w);
    if (c == 0) hipLaunchKernelGGL(consume_kernel<0>, dim3(NCONS), dim3(64), 0, s, w, seen, xcc);
    else if (c == 1) hipLaunchKernelGGL(consume_kernel<1>, dim3(NCONS), dim3(64), 0, s, w, seen, xcc);
    else hipLaunchKernelGGL(consume_kernel<2>, dim3(NCONS), dim3(64), 0, s, w, seen, xcc);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("zero: %-11s | producer: %-38s | consumer: %-17s :", zn[z], pn[p], cn[c]);
    int bad_variant = 0;
    for (int r = 0; r < 5; ++r) {
      // new data each replay: the maximum DEcreases, so a stale word (previous replay's larger maximum, or a stale zero)
      // is told apart from the right one
      const float mx = 1000.f - 100.f * r;
      for (long i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
      hx[(n / 3 + 977 * r) % n] = mx;
      CK(hipMemcpyAsync(x, hx.data(), n * 4, hipMemcpyHostToDevice, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      unsigned hs[NCONS], hc[NCONS], hw[SLOTS];
      CK(hipMemcpy(hs, seen, sizeof hs, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, xcc, sizeof hc, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hw, w, sizeof hw, hipMemcpyDeviceToHost));
      unsigned want; { float f = mx; memcpy(&want, &f, 4); }
      unsigned wmax = 0; for (int i = 0; i < SLOTS; ++i) wmax = wmax > hw[i] ? wmax : hw[i];
      int bad = 0, badx[16] = {0};
      unsigned example = 0;
      for (int i = 0; i < NCONS; ++i) if (hs[i] != want) { ++bad; ++badx[hc[i] & 15]; example = hs[i]; }
      printf(" r%d:%s", r, bad ? "" : "ok");
      if (bad) {
        float ef; memcpy(&ef, &example, 4);
        printf("%d/%d wrong (e.g. saw %g, want %g; words after replay %s; wrong per XCD:", bad, NCONS, ef, mx, wmax == want ? "RIGHT" : "WRONG");
        for (int k = 0; k < 8; ++k) printf(" %d", badx[k]);
        printf(")");
      }
      bad_variant += bad;
    }
    printf("\n");
    bad_total += bad_variant;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  printf("total wrong reads: %d\n", bad_total);
  return 0;
}
