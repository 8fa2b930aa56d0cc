// Polarized self-attention (PSA_s) of the fork's pose_hrnet_psa model on gfx950.
//
// Replaces PSA_s.forward, deep_hrnet/lib/models/PSA.py:190-269 (spatial_pool :190-222,
// channel_pool :224-258), inserted after relu(bn1(conv1)) of every BasicBlock
// (deep_hrnet/lib/models/pose_hrnet_psa.py:37,49).  Per image, on an NHWC map x [HW][C]:
//
//   spatial_pool: q = softmax_HW(wq . x_p);  ctx = Wv (sum_p q_p x_p)        (the C/2-channel
//                 conv_v_right never has to be materialised: Wv is linear)
//                 m = sigmoid(W2 relu(LN(W1 ctx + b1)) + b2);  x1 = x * m[c]
//   channel_pool: gbar = Wg mean_p(x1) = Wg (m * mean_p x);  theta = Wt x1_p (1x1 conv, MFMA kernel)
//                 s_p = sigmoid(sum_j gbar_j softmax_HW(theta_.j)_p);  x2 = x1 * s_p
//
// Four small HBM-bound kernels (one workgroup per image; the maps are re-read from L2) + the
// existing conv kernel for theta.  All reductions in fp32.
#include <type_traits>

#include "common.h"

namespace udp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Split-fp16 storage (UDP_F16X2, csrc/conv.hip): per pixel the C hi values, then the C lo values (fp16),
// x = hi + lo * 2^-11.
struct H2 {};
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

// channel c of pixel px of a [pixels][C] map stored as T
template <typename T>
__device__ __forceinline__ float ldx(const void* base, size_t px, int C, int c) {
  if constexpr (std::is_same<T, H2>::value) {
    const _Float16* q = reinterpret_cast<const _Float16*>(base) + px * 2 * C + c;
    return (float)q[0] + (float)q[C] * kLoInv;
  } else {
    return (float)reinterpret_cast<const T*>(base)[px * C + c];
  }
}
template <typename T>
__device__ __forceinline__ void stx(void* base, size_t px, int C, int c, float v) {
  if constexpr (std::is_same<T, H2>::value) {
    _Float16* q = reinterpret_cast<_Float16*>(base) + px * 2 * C + c;
    const _Float16 hi = (_Float16)v;
    h2_range_check(__builtin_fabsf(v));
    q[0] = hi;
    q[C] = (_Float16)((v - (float)hi) * kLoScale);
  } else {
    reinterpret_cast<T*>(base)[px * C + c] = (T)v;
  }
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(v, off);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = is_max ? fmaxf(r, red[k]) : r + red[k];
  return r;
}

// PSA fp32 parameter block: wq[C] | Wv[C/2][C] | W1[C/8][C/2] | b1[C/8] | ln_g[C/8] | ln_b[C/8] |
// W2[C][C/8] | b2[C] | Wg[C/2][C]
struct PsaW {
  const float *wq, *wv, *w1, *b1, *lg, *lb, *w2, *b2, *wg;
  __device__ PsaW(const float* p, int C) {
    wq = p;
    wv = wq + C;
    w1 = wv + (C / 2) * C;
    b1 = w1 + (C / 8) * (C / 2);
    lg = b1 + C / 8;
    lb = lg + C / 8;
    w2 = lb + C / 8;
    b2 = w2 + C * (C / 8);
    wg = b2 + C;
  }
};

// out[n] = { xbar[C] = sum_p softmax(q)_p x_p , xmean[C] }   (fp32)
template <typename T>
__global__ __launch_bounds__(256) void psa_pool_kernel(const ConvParams p) {
  extern __shared__ float sm[];   // e[HW] | part[256 * 2]
  __shared__ float red[4];
  const int C = p.Cin, HW = p.Hin * p.Win;
  const int n = blockIdx.x;
  const size_t px0 = (size_t)n * HW;
  const PsaW w(reinterpret_cast<const float*>(p.wgt), C);
  float* e = sm;
  float* part = sm + HW;
  float lmax = -INFINITY;
  for (int px = threadIdx.x; px < HW; px += 256) {
    float q = 0.f;
    for (int c = 0; c < C; ++c) q = fmaf(ldx<T>(p.in, px0 + px, C, c), w.wq[c], q);
    e[px] = q;
    lmax = fmaxf(lmax, q);
  }
  const float gmax = block_reduce(lmax, true, red);
  float lsum = 0.f;
  for (int px = threadIdx.x; px < HW; px += 256) {
    const float v = expf(e[px] - gmax);
    e[px] = v;
    lsum += v;
  }
  const float gsum = block_reduce(lsum, false, red);
  // channel c = tid % C, pixel group = tid / C (256 % C == 0)
  const int c = threadIdx.x % C, grp = threadIdx.x / C, ngrp = 256 / C;
  float a = 0.f, m = 0.f;
  for (int px = grp; px < HW; px += ngrp) {
    const float v = ldx<T>(p.in, px0 + px, C, c);
    a = fmaf(e[px], v, a);
    m += v;
  }
  part[threadIdx.x] = a;
  part[256 + threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x < C) {
    float sa = 0.f, smn = 0.f;
    for (int g = 0; g < ngrp; ++g) {
      sa += part[g * C + threadIdx.x];
      smn += part[256 + g * C + threadIdx.x];
    }
    float* o = reinterpret_cast<float*>(p.out) + (size_t)n * 2 * C;
    o[threadIdx.x] = sa / gsum;
    o[C + threadIdx.x] = smn / (float)HW;
  }
}

// out[n] = { m[C] = sigmoid(W2 relu(LN(W1 Wv xbar + b1)) + b2) , gbar[C/2] = Wg (m * xmean) }
__global__ __launch_bounds__(256) void psa_mlp_kernel(const ConvParams p) {
  __shared__ float xb[256], xm[256], ctx[128], h[32], msk[256];
  __shared__ float stat[2];
  const int C = p.Cin, C2 = C / 2, C8 = C / 8;
  const int n = blockIdx.x, t = threadIdx.x;
  const PsaW w(reinterpret_cast<const float*>(p.wgt), C);
  const float* s = reinterpret_cast<const float*>(p.in) + (size_t)n * 2 * C;
  if (t < C) {
    xb[t] = s[t];
    xm[t] = s[C + t];
  }
  __syncthreads();
  if (t < C2) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(w.wv[t * C + c], xb[c], a);
    ctx[t] = a;
  }
  __syncthreads();
  if (t < C8) {
    float a = w.b1[t];
    for (int j = 0; j < C2; ++j) a = fmaf(w.w1[t * C2 + j], ctx[j], a);
    h[t] = a;
  }
  __syncthreads();
  if (t == 0) {
    float mu = 0.f;
    for (int k = 0; k < C8; ++k) mu += h[k];
    mu /= (float)C8;
    float var = 0.f;
    for (int k = 0; k < C8; ++k) var += (h[k] - mu) * (h[k] - mu);
    stat[0] = mu;
    stat[1] = rsqrtf(var / (float)C8 + 1e-5f);
  }
  __syncthreads();
  if (t < C8) h[t] = fmaxf((h[t] - stat[0]) * stat[1] * w.lg[t] + w.lb[t], 0.f);
  __syncthreads();
  float* o = reinterpret_cast<float*>(p.out) + (size_t)n * (C + C2);
  if (t < C) {
    float a = w.b2[t];
    for (int k = 0; k < C8; ++k) a = fmaf(w.w2[t * C8 + k], h[k], a);
    const float m = 1.f / (1.f + expf(-a));
    msk[t] = m * xm[t];
    o[t] = m;
  }
  __syncthreads();
  if (t < C2) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(w.wg[t * C + c], msk[c], a);
    o[C + t] = a;
  }
}

// out[n,p,c] = in[n,p,c] * mask[n][c]     (mask: fp32 rows of `res`, pitch = res_pitch floats)
template <typename T>
__global__ __launch_bounds__(256) void psa_scale_kernel(const ConvParams p) {
  const int C = p.Cin, HW = p.Hin * p.Win;
  const long total = (long)p.N * HW * C;
  const float* mask = reinterpret_cast<const float*>(p.res);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = i % C;
    const long n = i / ((long)HW * C);
    const size_t px = (size_t)(i / C);
    stx<T>(p.out, px, C, c, ldx<T>(p.in, px, C, c) * mask[n * p.res_pitch + c]);
  }
}

// in = theta [N][HW][C/2], res = x1 [N][HW][C], up[0] = {m, gbar} fp32 rows; out = x1 * sigmoid(ctx_p)
template <typename T>
__global__ __launch_bounds__(256) void psa_sp_kernel(const ConvParams p) {
  extern __shared__ float sm[];   // msp[HW] | part[256] | M[C2] | S[C2] | g[C2]
  const int C = p.Cout, C2 = p.Cin, HW = p.Hin * p.Win;
  const int n = blockIdx.x, t = threadIdx.x;
  const size_t px0 = (size_t)n * HW;
  float* msp = sm;
  float* part = sm + HW;
  float* M = part + 256;
  float* S = M + C2;
  float* g = S + C2;
  const int j = t % C2, grp = t / C2, ngrp = 256 / C2;
  float lm = -INFINITY;
  for (int px = grp; px < HW; px += ngrp) lm = fmaxf(lm, ldx<T>(p.in, px0 + px, C2, j));
  part[t] = lm;
  __syncthreads();
  if (t < C2) {
    float m = part[t];
    for (int k = 1; k < ngrp; ++k) m = fmaxf(m, part[k * C2 + t]);
    M[t] = m;
    g[t] = reinterpret_cast<const float*>(p.up[0])[(size_t)n * (C + C2) + C + t];
  }
  __syncthreads();
  float ls = 0.f;
  for (int px = grp; px < HW; px += ngrp) ls += expf(ldx<T>(p.in, px0 + px, C2, j) - M[j]);
  part[t] = ls;
  __syncthreads();
  if (t < C2) {
    float s = part[t];
    for (int k = 1; k < ngrp; ++k) s += part[k * C2 + t];
    S[t] = s;
  }
  __syncthreads();
  for (int px = t; px < HW; px += 256) {
    float ctx = 0.f;
    for (int k = 0; k < C2; ++k) ctx = fmaf(g[k], expf(ldx<T>(p.in, px0 + px, C2, k) - M[k]) / S[k], ctx);
    msp[px] = 1.f / (1.f + expf(-ctx));
  }
  __syncthreads();
  for (int i = t; i < HW * C; i += 256) stx<T>(p.out, px0 + i / C, C, i % C, ldx<T>(p.res, px0 + i / C, C, i % C) * msp[i / C]);
}

static int check_psa_c(int C) {
  if (C < 16 || C > 256 || (256 % C) != 0 || (C % 16) != 0)
    return fail(UDP_ERR_UNSUPPORTED, "PSA: C=%d must divide 256 and be a multiple of 16", C);
  return UDP_OK;
}

int describe_psa(const ConvParams& p, int dtype, int kind, Launch* out) {
  const int HW = p.Hin * p.Win;
  out->block = dim3(256);
  out->p = p;
  out->lds = 0;
  if (dtype != UDP_F32 && dtype != UDP_BF16 && dtype != UDP_F16X2) return fail(UDP_ERR_ARG, "polarized self-attention ops: dtype %d", dtype);
  auto pick = [&](const void* kf, const void* kb, const void* kh) { return dtype == UDP_F32 ? kf : dtype == UDP_BF16 ? kb : kh; };
  switch (kind) {
    case UDP_OP_PSA_POOL: {
      const int rc = check_psa_c(p.Cin);
      if (rc) return rc;
      out->fn = pick(reinterpret_cast<const void*>(&psa_pool_kernel<float>), reinterpret_cast<const void*>(&psa_pool_kernel<__bf16>),
                     reinterpret_cast<const void*>(&psa_pool_kernel<H2>));
      out->grid = dim3(p.N);
      out->lds = (unsigned)((HW + 512) * sizeof(float));
      return UDP_OK;
    }
    case UDP_OP_PSA_MLP: {
      const int rc = check_psa_c(p.Cin);
      if (rc) return rc;
      out->fn = reinterpret_cast<const void*>(&psa_mlp_kernel);
      out->grid = dim3(p.N);
      return UDP_OK;
    }
    case UDP_OP_PSA_SCALE: {
      out->fn = pick(reinterpret_cast<const void*>(&psa_scale_kernel<float>), reinterpret_cast<const void*>(&psa_scale_kernel<__bf16>),
                     reinterpret_cast<const void*>(&psa_scale_kernel<H2>));
      long blocks = ((long)p.N * HW * p.Cin + 255) / 256;
      out->grid = dim3((unsigned)(blocks > 8192 ? 8192 : blocks));
      return UDP_OK;
    }
    case UDP_OP_PSA_SP: {
      const int rc = check_psa_c(p.Cout);
      if (rc) return rc;
      out->fn = pick(reinterpret_cast<const void*>(&psa_sp_kernel<float>), reinterpret_cast<const void*>(&psa_sp_kernel<__bf16>),
                     reinterpret_cast<const void*>(&psa_sp_kernel<H2>));
      out->grid = dim3(p.N);
      out->lds = (unsigned)((HW + 256 + 3 * p.Cin) * sizeof(float));
      return UDP_OK;
    }
  }
  return fail(UDP_ERR_ARG, "describe_psa: kind %d", kind);
}

int psa_h2_overflow(hipStream_t s, int reset, int* flag) { return h2_overflow_fetch(s, reset, flag); }

}  // namespace udp
