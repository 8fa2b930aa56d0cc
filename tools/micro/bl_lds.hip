// Micro-test: does buffer_load ... lds write ZEROS to LDS for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, int nbytes, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 64 * 4; i += 64) s[i] = -7.0f;   // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  unsigned off = (lane & 1) ? 0x7FFF0000u : lane * 16;   // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = s[i];
}
int main() {
  const int n = 64 * 4;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, n * 4, o);
  std::vector<float> r(n);
  hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
  int zeros = 0, poison = 0, good = 0, other = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    float v = r[l * 4 + j];
    if (l & 1) { if (v == 0.0f) ++zeros; else if (v == -7.0f) ++poison; else ++other; }
    else { if (v == h[l * 4 + j]) ++good; else ++other; }
  }
  printf("even lanes correct %d/128, odd lanes: zero %d poison(untouched) %d other %d\n", good, zeros, poison, other);
  return 0;
}
