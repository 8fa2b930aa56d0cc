// Fused convolution kernels of the HRNet forward for gfx950 (MI355X, CDNA4).
//
// Replaces the cuDNN conv + BatchNorm + ReLU (+ residual add, + nearest-upsample
// fuse sum) call chains of deep_hrnet/lib/models/pose_hrnet.py:43-59 (BasicBlock),
// :80-100 (Bottleneck), :260-273 with :189-255 (exchange unit) and :344-383
// (transitions).  Activations are NHWC; BatchNorm(eval) is folded into weights and
// bias on the host, so one launch computes
//     out = act( conv(in) + bias [+ res] [+ sum_k nearest_up(up_k)] ).
//
// conv_mfma_kernel: implicit GEMM on the matrix cores.  A workgroup (4 waves)
// owns a tile of G images x R rows x TW columns of output pixels (M <= 256) and a
// block of BN = 16*NB output channels.  Per chunk of CK input channels it stages
// the input halo tile and the [tap][cout][cin-chunk] weights in LDS (80-byte rows:
// 64 B payload + 16 B pad, conflict-free for the fragment reads below), then every
// wave runs taps x k-steps of MFMA with  A = weights (rows = cout),
// B = pixels (cols = pixel), so each lane ends up with 4 consecutive output
// channels of one pixel -> one 8/16-byte NHWC store per accumulator tile.
//   fp32 : v_mfma_f32_16x16x4_f32  (exact fp32 fma chain), ds_read_b64 feeds 2 MFMAs
//   bf16 : v_mfma_f32_16x16x32_bf16 (fp32 accumulate),     ds_read_b128 feeds 1 MFMA
#include <type_traits>

#include "common.h"

namespace udp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 80;  // LDS row pitch in bytes (pixel row / weight row)

template <typename T>
struct Tr;
template <>
struct Tr<float> {
  static constexpr int CK = 16;  // input channels per LDS chunk (64 B)
};
template <>
struct Tr<__bf16> {
  static constexpr int CK = 32;
};

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load4<__bf16>(const __bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}
template <>
__device__ __forceinline__ void store4<__bf16>(__bf16* p, f32x4 v) {
  bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

template <typename T, int KS, int STRIDE, int NB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
  constexpr int CK = Tr<T>::CK;
  constexpr int BN = NB * 16;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 15;
  const int kg = lane >> 4;

  int t = blockIdx.x;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  t /= p.tiles_y;
  const int n0 = t * p.G;
  const int y0 = ty * p.R;
  const int x0 = tx * p.TW;
  const int cb = blockIdx.y;

  const int IH = p.IH, IW = p.IW;
  const int npix_in = p.G * IH * IW;
  unsigned char* in_lds = smem;
  unsigned char* w_lds = smem + (size_t)npix_in * ROWB;

  const int RT = p.R * p.TW;
  const int M = p.G * RT;
  const int nmb = (M + 15) >> 4;

  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = (wave + 4 * i) * 16 + li;
    m = m < M ? m : M - 1;
    const int g = m / RT;
    const int rem = m - g * RT;
    const int r = rem / p.TW;
    const int x = rem - r * p.TW;
    a_off[i] = ((g * IH + r * STRIDE) * IW + x * STRIDE) * ROWB;
  }

  f32x4 acc[4][NB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* in_base = reinterpret_cast<const char*>(p.in);
  const char* w_base = reinterpret_cast<const char*>(p.wgt);

  for (int c0 = 0; c0 < p.Cin; c0 += CK) {
    if (c0) __syncthreads();
    // ---- stage the input halo tile: 4 x 16-byte pieces per pixel
    for (int piece = tid; piece < npix_in * 4; piece += 256) {
      const int pix = piece >> 2, part = piece & 3;
      const int ix = pix % IW;
      const int tmp = pix / IW;
      const int iy = tmp % IH;
      const int g = tmp / IH;
      const int n = n0 + g;
      const int gy = y0 * STRIDE - PAD + iy;
      const int gx = x0 * STRIDE - PAD + ix;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < p.N && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) {
        const size_t e = ((size_t)(n * p.Hin + gy) * p.Win + gx) * p.Cin + c0;
        v = *reinterpret_cast<const uint4*>(in_base + e * sizeof(T) + part * 16);
      }
      *reinterpret_cast<uint4*>(in_lds + pix * ROWB + part * 16) = v;
    }
    // ---- stage the weights of this cout block / cin chunk: rows = (tap, cout)
    for (int piece = tid; piece < TAPS * BN * 4; piece += 256) {
      const int row = piece >> 2, part = piece & 3;
      const int tap = row / BN;
      const int co = row - tap * BN;
      const size_t e = ((size_t)(tap * p.CoutPad + cb * BN + co)) * p.Cin + c0;
      const uint4 v = *reinterpret_cast<const uint4*>(w_base + e * sizeof(T) + part * 16);
      *reinterpret_cast<uint4*>(w_lds + row * ROWB + part * 16) = v;
    }
    __syncthreads();

    // ---- MFMA over taps x k-steps
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap % KS;
      const int tap_off = (ky * IW + kx) * ROWB;
      const unsigned char* wrow = w_lds + (tap * BN + li) * ROWB;
      if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int ks = 0; ks < CK / 8; ++ks) {
          const int koff = ks * 32 + kg * 8;
          f32x2 wf[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            wf[nb] = *reinterpret_cast<const f32x2*>(wrow + nb * 16 * ROWB + koff);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (wave + 4 * i < nmb) {
              const f32x2 pf = *reinterpret_cast<const f32x2*>(in_lds + a_off[i] + tap_off + koff);
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb][0], pf[0], acc[i][nb], 0, 0, 0);
                acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb][1], pf[1], acc[i][nb], 0, 0, 0);
              }
            }
          }
        }
      } else {
        const int koff = kg * 16;
        bf16x8 wf[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          wf[nb] = *reinterpret_cast<const bf16x8*>(wrow + nb * 16 * ROWB + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (wave + 4 * i < nmb) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(in_lds + a_off[i] + tap_off + koff);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], pf, acc[i][nb], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: lane holds couts cbase+0..3 of pixel m
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = (wave + 4 * i) * 16 + li;
    if (m >= M) continue;
    const int g = m / RT;
    const int rem = m - g * RT;
    const int r = rem / p.TW;
    const int xo = x0 + rem - r * p.TW;
    const int n = n0 + g;
    const int y = y0 + r;
    if (n >= p.N || y >= p.Hout || xo >= p.Wout) continue;
    const size_t pix = ((size_t)(n * p.Hout + y) * p.Wout + xo);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int c = cb * BN + nb * 16 + kg * 4;
      f32x4 v = acc[i][nb];
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + c);
      v += b;
      if (p.out_nchw_f32) {
        float* o = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c + q < p.Cout) {
            float f = v[q];
            if (p.relu) f = f > 0.f ? f : 0.f;
            o[((size_t)(n * p.Cout + c + q) * p.Hout + y) * p.Wout + xo] = f;
          }
        continue;
      }
      if (c >= p.Cout) continue;
      if (p.res) v += load4<T>(reinterpret_cast<const T*>(p.res) + pix * p.Cout + c);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (u < p.nup) {
          const int s = p.up_shift[u];
          const size_t up_pix = ((size_t)(n * (p.Hout >> s) + (y >> s)) * (p.Wout >> s) + (xo >> s));
          v += load4<T>(reinterpret_cast<const T*>(p.up[u]) + up_pix * p.Cout + c);
        }
      }
      if (p.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
      }
      store4<T>(reinterpret_cast<T*>(p.out) + pix * p.Cout + c, v);
    }
  }
}

// Stem conv1: 3x3 stride-2 conv on the NCHW fp32 network input (Cin = 3), direct
// VALU form (27 taps), + folded BN + ReLU, NHWC output.  pose_hrnet.py:290-292,
// :437-439.  Images n >= flip_from read image n - flip_from mirrored along W
// (flip-test second pass, function.py:154-156).
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) float w_s[27 * 64];
  __shared__ __attribute__((aligned(16))) float b_s[64];
  const int tid = threadIdx.x;
  const float* wg = reinterpret_cast<const float*>(p.wgt);
  for (int i = tid; i < 27 * 64; i += 256) w_s[i] = wg[i];
  if (tid < 64) b_s[tid] = p.bias[tid];
  __syncthreads();
  const int cg = tid & 3;
  const long pix = (long)blockIdx.x * 64 + (tid >> 2);
  const long total = (long)p.N * p.Hout * p.Wout;
  if (pix >= total) return;
  const int xo = pix % p.Wout;
  const long t2 = pix / p.Wout;
  const int yo = t2 % p.Hout;
  const int n = t2 / p.Hout;
  const bool mirror = n >= p.flip_from;
  const int ns = mirror ? n - p.flip_from : n;
  const float* in = reinterpret_cast<const float*>(p.in) + (size_t)ns * 3 * p.Hin * p.Win;
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = b_s[cg * 16 + q];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int gy = yo * 2 - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int gx = xo * 2 - 1 + kx;
      const bool ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
      const int sx = mirror ? p.Win - 1 - gx : gx;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = ok ? in[((size_t)ci * p.Hin + gy) * p.Win + sx] : 0.f;
        const float* wr = &w_s[((ky * 3 + kx) * 3 + ci) * 64 + cg * 16];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + q);
          acc[q + 0] = fmaf(v, w4[0], acc[q + 0]);
          acc[q + 1] = fmaf(v, w4[1], acc[q + 1]);
          acc[q + 2] = fmaf(v, w4[2], acc[q + 2]);
          acc[q + 3] = fmaf(v, w4[3], acc[q + 3]);
        }
      }
    }
  }
  T* o = reinterpret_cast<T*>(p.out) + (size_t)pix * 64 + cg * 16;
#pragma unroll
  for (int q = 0; q < 16; q += 4) {
    f32x4 v = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
    if (p.relu) {
#pragma unroll
      for (int z = 0; z < 4; ++z) v[z] = v[z] > 0.f ? v[z] : 0.f;
    }
    store4<T>(o + q, v);
  }
}

// Exchange-unit output for the highest-resolution branch when it has no conv
// term: out = relu(x_i + sum_k nearest_up(T_ik)), pose_hrnet.py:267-272 with the
// identity f_ii of :222-223.
template <typename T>
__global__ __launch_bounds__(256) void fuse_sum_kernel(const ConvParams p) {
  const int C4 = p.Cout >> 2;
  const long total = (long)p.N * p.Hout * p.Wout * C4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (idx % C4) * 4;
    const long pix = idx / C4;
    const int xo = pix % p.Wout;
    const long t2 = pix / p.Wout;
    const int y = t2 % p.Hout;
    const int n = t2 / p.Hout;
    f32x4 v = load4<T>(reinterpret_cast<const T*>(p.in) + pix * p.Cout + c);
    for (int u = 0; u < p.nup; ++u) {
      const int s = p.up_shift[u];
      const size_t up_pix = ((size_t)(n * (p.Hout >> s) + (y >> s)) * (p.Wout >> s) + (xo >> s));
      v += load4<T>(reinterpret_cast<const T*>(p.up[u]) + up_pix * p.Cout + c);
    }
    if (p.relu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
    }
    store4<T>(reinterpret_cast<T*>(p.out) + pix * p.Cout + c, v);
  }
}

// ---------------------------------------------------------------------------
// Host side: tile selection + dispatch
// ---------------------------------------------------------------------------
static int largest_divisor_leq(int n, int lim) {
  for (int d = lim < n ? lim : n; d >= 1; --d)
    if (n % d == 0) return d;
  return 1;
}

// Picks the (G, R, TW) tile and NB for one conv; returns LDS bytes.
size_t conv_choose_tile(ConvParams& p, int ks, int stride, int* nb_out) {
  const int kMaxM = 256;
  int TW = p.Wout;
  while (TW > 64) TW = (TW + 1) / 2;
  int maxR = kMaxM / TW;
  if (maxR < 1) maxR = 1;
  if (maxR > p.Hout) maxR = p.Hout;
  int R = largest_divisor_leq(p.Hout, maxR);
  if (R * 2 <= maxR) R = maxR;
  int G = 1;
  if (R == p.Hout && TW == p.Wout) {
    G = kMaxM / (R * TW);
    if (G < 1) G = 1;
    if (G > p.N) G = p.N;
  }
  int NB = p.CoutPad >= 64 ? 4 : 2;
  auto lds = [&](int g, int r) {
    const int ih = (r - 1) * stride + ks, iw = (TW - 1) * stride + ks;
    return (size_t)(g * ih * iw + ks * ks * NB * 16) * ROWB;
  };
  const size_t kLimit = 72 * 1024;
  while (lds(G, R) > kLimit && G > 1) --G;
  while (lds(G, R) > kLimit && R > 1) R = (R + 1) / 2;
  p.G = G;
  p.R = R;
  p.TW = TW;
  p.IH = (R - 1) * stride + ks;
  p.IW = (TW - 1) * stride + ks;
  p.tiles_x = ceil_div(p.Wout, TW);
  p.tiles_y = ceil_div(p.Hout, R);
  *nb_out = NB;
  return lds(G, R);
}

template <typename T, int KS, int STRIDE, int NB>
static int launch_one(const ConvParams& p, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  auto kern = conv_mfma_kernel<T, KS, STRIDE, NB>;
  if (!attr_set) {
    UDP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const dim3 grid(ceil_div(p.N, p.G) * p.tiles_y * p.tiles_x, p.CoutPad / (NB * 16));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

template <typename T>
static int launch_conv_t(const ConvParams& p, int ks, int stride, int nb, size_t lds, hipStream_t s) {
#define UDP_CASE(K, S, B) \
  if (ks == K && stride == S && nb == B) return launch_one<T, K, S, B>(p, lds, s);
  UDP_CASE(3, 1, 2) UDP_CASE(3, 1, 4) UDP_CASE(3, 2, 2) UDP_CASE(3, 2, 4)
  UDP_CASE(1, 1, 2) UDP_CASE(1, 1, 4)
#undef UDP_CASE
  return fail(UDP_ERR_UNSUPPORTED, "conv ks=%d stride=%d nb=%d has no kernel", ks, stride, nb);
}

int launch_conv(const ConvParams& p, int dtype, int ks, int stride, int nb, size_t lds, hipStream_t s) {
  if (p.Cin % 32 != 0) return fail(UDP_ERR_UNSUPPORTED, "conv Cin=%d is not a multiple of 32", p.Cin);
  if (p.CoutPad % (nb * 16) != 0)
    return fail(UDP_ERR_UNSUPPORTED, "conv CoutPad=%d is not a multiple of %d", p.CoutPad, nb * 16);
  if (!p.out_nchw_f32 && p.Cout % 16 != 0)
    return fail(UDP_ERR_UNSUPPORTED, "NHWC conv Cout=%d is not a multiple of 16", p.Cout);
  if (dtype == UDP_F32) return launch_conv_t<float>(p, ks, stride, nb, lds, s);
  return launch_conv_t<__bf16>(p, ks, stride, nb, lds, s);
}

int launch_stem(const ConvParams& p, int dtype, hipStream_t s) {
  if (p.Cout != 64) return fail(UDP_ERR_UNSUPPORTED, "stem conv expects 64 output channels, got %d", p.Cout);
  const long total = (long)p.N * p.Hout * p.Wout;
  const dim3 grid((unsigned)((total + 63) / 64));
  if (dtype == UDP_F32)
    hipLaunchKernelGGL(stem_conv_kernel<float>, grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(stem_conv_kernel<__bf16>, grid, dim3(256), 0, s, p);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

int launch_fuse(const ConvParams& p, int dtype, hipStream_t s) {
  if (p.Cout % 4 != 0) return fail(UDP_ERR_UNSUPPORTED, "fuse Cout=%d is not a multiple of 4", p.Cout);
  const long total = (long)p.N * p.Hout * p.Wout * (p.Cout / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == UDP_F32)
    hipLaunchKernelGGL(fuse_sum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(fuse_sum_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

}  // namespace udp
