#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x27000);
}
__global__ void k(const float4* src, float4* dst, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  __amdgpu_buffer_rsrc_t r = mk(src, (long)n * 16);
  // each lane fetches word (n-1-idx) -> reversed copy, invalid lanes out of range -> zeros
  for (int k = 0; k < 2; ++k) {
    const int it = tid + 256 * k;
    const int off = (it < n) ? (n - 1 - it) * 16 : 0x7ffffff0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + (256 * k + 64 * wv) * 16), 16, off, 0, 0, 0);
  }
  __syncthreads();
  for (int k = 0; k < 2; ++k) dst[tid + 256 * k] = *(float4*)(smem + (tid + 256 * k) * 16);
}
int main() {
  int n = 400;
  std::vector<float4> h(512);
  for (int i = 0; i < 512; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
  float4 *s, *d; hipMalloc(&s, 512 * 16); hipMalloc(&d, 512 * 16);
  hipMemcpy(s, h.data(), 512 * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 512 * 16, 0, s, d, n);
  std::vector<float4> o(512); hipMemcpy(o.data(), d, 512 * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; ++i) { float e = i < n ? (float)(n - 1 - i) : 0.f; if (o[i].x != e) { if (bad < 5) printf("i=%d got %f want %f\n", i, o[i].x, e); ++bad; } }
  printf("bad %d\n", bad);
}
