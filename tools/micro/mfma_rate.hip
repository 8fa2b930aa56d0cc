// MFMA issue-rate probe (gfx950): cycles per v_mfma_f32_16x16x32_f16 for accumulator dependency patterns.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void probe(const f16x8* in, f32x4* out, unsigned long long* cyc, int iters) {
  f16x8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
  f32x4 c[12];
  for (int i = 0; i < 12; ++i) c[i] = f32x4{0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (PAT == 0) {   // 6 independent accumulators
      c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[2], 0, 0, 0);
      c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[3], 0, 0, 0);
      c[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, b1, c[4], 0, 0, 0);
      c[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, b0, c[5], 0, 0, 0);
    } else if (PAT == 1) {   // the conv pattern: acc0 acc1 accx0 accx1 accx0 accx1
      c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[2], 0, 0, 0);
      c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[3], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, b1, c[2], 0, 0, 0);
      c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, b0, c[3], 0, 0, 0);
    } else if (PAT == 2) {   // one accumulator chain
#pragma unroll
      for (int k = 0; k < 6; ++k) c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0], 0, 0, 0);
    } else {   // pairs: acc0 acc0 acc1 acc1 acc2 acc2 (each accumulator twice in a row)
      c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0], 0, 0, 0);
      c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[1], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, b1, c[2], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, b0, c[2], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = c[0];
  for (int i = 1; i < 12; ++i) s += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  f16x8* in;
  f32x4* out;
  unsigned long long* cyc;
  const int blocks = 256, iters = 2000;
  hipMalloc(&in, 1024 * sizeof(f16x8));
  hipMalloc(&out, blocks * 256 * sizeof(f32x4));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  _Float16 h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = (_Float16)((i * 2654435761u >> 20 & 1023) / 1024.0f - 0.5f);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  unsigned long long hc[blocks];
  for (int pat = 0; pat < 4; ++pat) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      if (pat == 0) probe<0><<<blocks, 256>>>(in, out, cyc, iters);
      if (pat == 1) probe<1><<<blocks, 256>>>(in, out, cyc, iters);
      if (pat == 2) probe<2><<<blocks, 256>>>(in, out, cyc, iters);
      if (pat == 3) probe<3><<<blocks, 256>>>(in, out, cyc, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
      double m = 0;
      for (int i = 0; i < blocks; ++i) m += hc[i];
      m /= blocks;
      if (rep) printf("pattern %d: %.2f s_memtime ticks per MFMA; kernel %.1f us -> %.2f ns per MFMA per SIMD\n", pat, m / (6.0 * iters), ms * 1e3, ms * 1e6 / (6.0 * iters));
    }
  }
  return 0;
}
