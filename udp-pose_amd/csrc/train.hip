// Training-step kernels for gfx950: what loss.backward() / optimizer.step() of
// deep_hrnet/lib/core/function.py:73-76 run for the HRNet graph of lib/models/pose_hrnet.py.
//
//   conv forward / input gradient : the implicit-GEMM kernel of conv.hip (udp_conv2d_fused) on weights
//                                   re-packed here each step (udp_pack_conv_weights): the input gradient
//                                   of a stride-1 conv is a stride-1 conv of dy with the taps mirrored and
//                                   Cin/Cout swapped; stride 2 goes through udp_zero_stuff2 first.
//   conv weight gradient          : udp_conv2d_wgrad, MFMA fp32 16x16x4, K = pixels (split-K partials +
//                                   one deterministic reduce).
//   BatchNorm2d (train mode)      : batch statistics in fp64, running-stat update, fused (+res)(+ReLU)
//                                   apply; backward with the ReLU mask folded in.
//   nearest-upsample / sum nodes  : udp_ew_accumulate, udp_upsample_bwd, udp_relu_bwd.
//   Adam                          : one launch over the flat parameter / gradient buffers.
//
// Tensors are NHWC `dtype` (fp32, or bf16 storage with fp32 arithmetic); parameters, gradients and
// optimizer state are fp32 in the reference's own layouts ([cout][cin][kh][kw]).
#include <type_traits>

#include "common.h"

namespace udp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float tof(T v) {
  return (float)v;
}
template <typename T>
__device__ __forceinline__ T fromf(float v) {
  return (T)v;
}

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }
static inline unsigned nblocks(long total, int per, long cap = 1 << 20) {
  long b = (total + per - 1) / per;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

// ---------------------------------------------------------------------------------------------
// weight re-pack: reference [cout][cin][ks][ks] fp32 ->
//   fwd   [tap][cout_pad][cin_k]            (cin_k = cin rounded up to 16, cout_pad to 32; zero filled)
//   dgrad [tap'][cin_pad][cout_k], tap' = mirrored tap, value W[co][ci][ks-1-ky'][ks-1-kx']
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int ks, int cout_pad, int cin_k,
                                    int cin_pad, int cout_k, T* __restrict__ fwd, T* __restrict__ dg) {
  const int taps = ks * ks;
  const long nf = (long)taps * cout_pad * cin_k;
  const long nd = dg ? (long)taps * cin_pad * cout_k : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (long)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = i % cin_k, co = (i / cin_k) % cout_pad, t = i / ((long)cin_k * cout_pad);
      fwd[i] = fromf<T>((co < cout && ci < cin) ? w[((long)co * cin + ci) * taps + t] : 0.f);
    } else {
      const long k = i - nf;
      const int co = k % cout_k, ci = (k / cout_k) % cin_pad, t = k / ((long)cout_k * cin_pad);
      dg[k] = fromf<T>((co < cout && ci < cin) ? w[((long)co * cin + ci) * taps + (taps - 1 - t)] : 0.f);
    }
  }
}

// all layers of a model in one launch: blockIdx.y = layer (descriptor table in device memory)
template <typename T>
__global__ void pack_weights_batch_kernel(const udp_pack_desc* __restrict__ descs) {
  const udp_pack_desc d = descs[blockIdx.y];
  const int taps = d.ks * d.ks, cout = d.cout, cin = d.cin;
  const int cout_pad = (cout + 31) / 32 * 32, cin_k = (cin + 15) / 16 * 16, cin_pad = (cin + 31) / 32 * 32, cout_k = (cout + 15) / 16 * 16;
  const float* w = d.w;
  T* fwd = reinterpret_cast<T*>(d.w_fwd);
  T* dg = reinterpret_cast<T*>(d.w_dgrad);
  const long nf = (long)taps * cout_pad * cin_k;
  const long nd = dg ? (long)taps * cin_pad * cout_k : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (long)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = i % cin_k, co = (i / cin_k) % cout_pad, t = i / ((long)cin_k * cout_pad);
      fwd[i] = fromf<T>((co < cout && ci < cin) ? w[((long)co * cin + ci) * taps + t] : 0.f);
    } else {
      const long k = i - nf;
      const int co = k % cout_k, ci = (k / cout_k) % cin_pad, t = k / ((long)cout_k * cin_pad);
      dg[k] = fromf<T>((co < cout && ci < cin) ? w[((long)co * cin + ci) * taps + (taps - 1 - t)] : 0.f);
    }
  }
}

// out[n][2y][2x][c] = in[n][y][x][c], zeros elsewhere
template <typename T>
__global__ void zero_stuff2_kernel(const T* __restrict__ in, long total_out, int h, int w, int c, T* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_out; i += (long)gridDim.x * blockDim.x) {
    const int ch = i % c;
    long r = i / c;
    const int x = r % (2 * w);
    r /= 2 * w;
    const int y = r % (2 * h);
    const long n = r / (2 * h);
    out[i] = ((x | y) & 1) ? fromf<T>(0.f) : in[((n * h + (y >> 1)) * w + (x >> 1)) * c + ch];
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  dW[co][ci][ky][kx] = sum_{n,y,x} dy[n,y,x,co] * x[n, y*s+ky-pad, x*s+kx-pad, ci]
// Workgroup = 32 co x 32 ci x all taps over `upw` pixel units; unit = TH x TW output pixels of one
// image staged in LDS with the matching input patch.  MFMA 16x16x4 fp32: A[m=co][k=pixel],
// B[k=pixel][n=ci]; wave w owns the 16x16 tile (w&1, w>>1) for every tap.
// Partials go to part[split][tap][co][ci]; wgrad_reduce_kernel sums the splits.
// ---------------------------------------------------------------------------------------------
struct WgradParams {
  const void* x;
  const void* dy;
  float* part;
  int N, Hin, Win, CinK, Hout, Wout, CoutK, stride;
  int TH, TW, units_y, units_x, units, upw;   // units per image / total / per workgroup
  int XH, XW, tiles_ci;
  int cout, cin;
};

constexpr int kWgPitch = 36;   // floats per staged pixel (32 channels + 4: 16-byte aligned, bank spread)

template <typename T, int KS>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  extern __shared__ float lds[];
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  const int PX = p.TH * p.TW, PXP = (PX + 3) & ~3;
  float* sDy = lds;                       // [PXP][pitch]
  float* sX = lds + PXP * kWgPitch;       // [XH*XW][pitch]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ci0 = (blockIdx.x % p.tiles_ci) * 32, co0 = (blockIdx.x / p.tiles_ci) * 32;
  const int mt = wave & 1, nt = wave >> 1;
  const T* gx = reinterpret_cast<const T*>(p.x);
  const T* gdy = reinterpret_cast<const T*>(p.dy);
  f32x4 acc[TAPS];
#pragma unroll
  for (int i = 0; i < TAPS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int u0 = blockIdx.y * p.upw, u1 = min(u0 + p.upw, p.units);
  const int upi = p.units_y * p.units_x;
  for (int u = u0; u < u1; ++u) {
    const int n = u / upi, ur = u % upi;
    const int oy0 = (ur / p.units_x) * p.TH, ox0 = (ur % p.units_x) * p.TW;
    __syncthreads();   // previous unit's MFMA reads are done
    // dy tile: pixel q -> (oy0 + q / TW, ox0 + q % TW); 8 threads x 4 channels per pixel
    for (int i = t; i < PXP * 8; i += 256) {
      const int q = i >> 3, c4 = (i & 7) * 4;
      const int oy = oy0 + q / p.TW, ox = ox0 + q % p.TW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < PX && oy < p.Hout && ox < p.Wout && co0 + c4 < p.CoutK) {
        const T* s = gdy + (((long)n * p.Hout + oy) * p.Wout + ox) * p.CoutK + co0 + c4;
        v = f32x4{tof(s[0]), tof(s[1]), tof(s[2]), tof(s[3])};
      }
      *reinterpret_cast<f32x4*>(sDy + q * kWgPitch + c4) = v;
    }
    const int iy0 = oy0 * p.stride - PAD, ix0 = ox0 * p.stride - PAD;
    for (int i = t; i < p.XH * p.XW * 8; i += 256) {
      const int q = i >> 3, c4 = (i & 7) * 4;
      const int iy = iy0 + q / p.XW, ix = ix0 + q % p.XW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && ci0 + c4 < p.CinK) {
        const T* s = gx + (((long)n * p.Hin + iy) * p.Win + ix) * p.CinK + ci0 + c4;
        v = f32x4{tof(s[0]), tof(s[1]), tof(s[2]), tof(s[3])};
      }
      *reinterpret_cast<f32x4*>(sX + q * kWgPitch + c4) = v;
    }
    __syncthreads();
    for (int k0 = 0; k0 < PXP; k0 += 4) {
      int q = k0 + (lane >> 4);
      const float a = sDy[q * kWgPitch + mt * 16 + (lane & 15)];
      q = min(q, PX - 1);               // padded pixels carry dy == 0; keep the x address inside the tile
      const int ty = q / p.TW, tx = q % p.TW;
      const float* bx = sX + ((ty * p.stride) * p.XW + tx * p.stride) * kWgPitch + nt * 16 + (lane & 15);
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
          acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx[(ky * p.XW + kx) * kWgPitch], acc[ky * KS + kx], 0, 0, 0);
    }
  }
  // D[m = 4*(lane/16)+r][n = lane%16]
  const int ci = ci0 + nt * 16 + (lane & 15);
  float* dst = p.part + (long)blockIdx.y * TAPS * p.cout * p.cin;
  if (ci < p.cin) {
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mt * 16 + 4 * (lane >> 4) + r;
        if (co < p.cout) dst[((long)tp * p.cout + co) * p.cin + ci] = acc[tp][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient on the bf16 matrix pipe.  Same tiling / partial layout as wgrad_kernel; the
// LDS images stay in the natural NHWC form [pixel][32 channels] (bf16, 80-byte rows) and both MFMA
// operands -- A[m=co][k=pixel], B[k=pixel][n=ci], K-major per lane -- come out of them through
// ds_read_b64_tr_b16 (gfx950 transposed LDS read: a 16-lane group reads 4 pixel rows x 16 channels and
// each lane receives one channel's 4 pixels), so no transposed copy is ever written.
// v_mfma_f32_32x32x16_bf16, fp32 accumulate.  The 4 waves take different 16-pixel k-steps of a unit
// (all 9 taps each) and are summed through LDS at the end; units are register-prefetched one ahead.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWbPitch = 80;      // bytes per staged pixel row (32 bf16 channels + 16 B)
constexpr int kWbItems = 8;       // 16-byte staging items per thread (rows*4/256 <= 8: up to 512 staged rows)

__device__ __forceinline__ bf16x8 tr_read8(const char* lds_lo, const char* lds_hi) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_hi));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int KS>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgradParams p) {
  extern __shared__ char ldsb[];
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  const int PX = p.TH * p.TW, PXP = (PX + 15) & ~15;
  const int XR = p.XH * p.XW, ROWS = PXP + XR;
  char* sDy = ldsb;                       // [PXP] rows
  char* sX = ldsb + PXP * kWbPitch;       // [XH*XW] rows
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ci0 = (blockIdx.x % p.tiles_ci) * 32, co0 = (blockIdx.x / p.tiles_ci) * 32;
  const __bf16* gx = reinterpret_cast<const __bf16*>(p.x);
  const __bf16* gdy = reinterpret_cast<const __bf16*>(p.dy);

  // staging plan of this thread (the tile geometry is the same for every unit)
  int it_row[kWbItems], it_a[kWbItems], it_b[kWbItems];      // row in LDS; (ty,tx) or (xy,xx)
#pragma unroll
  for (int k = 0; k < kWbItems; ++k) {
    const int item = t + k * 256, r = item >> 2;
    it_row[k] = r < ROWS ? r : -1;
    if (r < PXP) {
      it_a[k] = r / p.TW;
      it_b[k] = r % p.TW;
    } else {
      it_a[k] = (r - PXP) / p.XW;
      it_b[k] = (r - PXP) % p.XW;
    }
  }
  const int seg = (t & 3) * 8;            // first of the 8 channels of this thread's 16-byte items

  // operand addresses of this lane (one k-step per wave and unit when PXP <= 64; more via the loop)
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, pp = li & 3;
  const int colb = (16 * (g & 1) + 4 * pp) * 2;              // byte offset of the 4 channels this lane addresses

  f32x16 acc[TAPS];
#pragma unroll
  for (int i = 0; i < TAPS; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int u0 = blockIdx.y * p.upw, u1 = min(u0 + p.upw, p.units);
  const int upi = p.units_y * p.units_x;
  u32x4 regs[kWbItems];

  auto fetch = [&](int u) {
    const int n = u / upi, ur = u % upi;
    const int oy0 = (ur / p.units_x) * p.TH, ox0 = (ur % p.units_x) * p.TW;
    const int iy0 = oy0 * p.stride - PAD, ix0 = ox0 * p.stride - PAD;
#pragma unroll
    for (int k = 0; k < kWbItems; ++k) {
      u32x4 v = {0u, 0u, 0u, 0u};
      const int r = it_row[k];
      if (r >= 0) {
        if (r < PXP) {
          const int oy = oy0 + it_a[k], ox = ox0 + it_b[k];
          if (r < PX && oy < p.Hout && ox < p.Wout && co0 + seg < p.CoutK)
            v = *reinterpret_cast<const u32x4*>(gdy + (((long)n * p.Hout + oy) * p.Wout + ox) * p.CoutK + co0 + seg);
        } else {
          const int iy = iy0 + it_a[k], ix = ix0 + it_b[k];
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && ci0 + seg < p.CinK)
            v = *reinterpret_cast<const u32x4*>(gx + (((long)n * p.Hin + iy) * p.Win + ix) * p.CinK + ci0 + seg);
        }
      }
      regs[k] = v;
    }
  };

  if (u0 < u1) fetch(u0);
  for (int u = u0; u < u1; ++u) {
    __syncthreads();                      // previous unit's operand reads are done
#pragma unroll
    for (int k = 0; k < kWbItems; ++k)
      if (it_row[k] >= 0) *reinterpret_cast<u32x4*>(ldsb + it_row[k] * kWbPitch + seg * 2) = regs[k];
    __syncthreads();
    if (u + 1 < u1) fetch(u + 1);         // next unit's global loads fly under the MFMAs
    for (int k0 = wave * 16; k0 < PXP; k0 += 64) {
      const int r0 = k0 + 8 * (g >> 1) + q4;                 // pixel rows this lane addresses: r0 and r0 + 4
      const bf16x8 a = tr_read8(sDy + r0 * kWbPitch + colb, sDy + (r0 + 4) * kWbPitch + colb);
      const int qa = min(r0, PX - 1), qb = min(r0 + 4, PX - 1);   // padded pixels carry dy == 0
      const char* xa = sX + (((qa / p.TW) * p.stride) * p.XW + (qa % p.TW) * p.stride) * kWbPitch + colb;
      const char* xb = sX + (((qb / p.TW) * p.stride) * p.XW + (qb % p.TW) * p.stride) * kWbPitch + colb;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int off = (ky * p.XW + kx) * kWbPitch;
          const bf16x8 b = tr_read8(xa + off, xb + off);
          acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ky * KS + kx], 0, 0, 0);
        }
    }
  }
  // sum the 4 waves through LDS, tap by tap; D[m = 8*(i/4) + 4*(lane/32) + i%4][n = lane%32]
  float* red = reinterpret_cast<float*>(ldsb);               // [4][1024]
  float* dst = p.part + (long)blockIdx.y * TAPS * p.cout * p.cin;
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
      red[wave * 1024 + m * 32 + (lane & 31)] = acc[tp][i];
    }
    __syncthreads();
    for (int e = t; e < 1024; e += 256) {
      const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
      const int co = co0 + (e >> 5), ci = ci0 + (e & 31);
      if (co < p.cout && ci < p.cin) dst[((long)tp * p.cout + co) * p.cin + ci] = v;
    }
  }
}

// dw[co][ci][tap] (+)= sum_s part[s][tap][co][ci].  Block = 16 element quads x 16 split lanes
// (fixed summation order: deterministic).
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int splits, int taps, int cout,
                                                  int cin, int accumulate, float* __restrict__ dw, unsigned bx) {
  __shared__ f32x4 red[16][17];
  const long per = (long)taps * cout * cin;          // multiple of 4? not necessarily: tail handled scalar
  const int q = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long i4 = ((long)bx * 16 + q) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 + 3 < per && (per & 3) == 0) {
    for (int k = sl; k < splits; k += 16) s += *reinterpret_cast<const f32x4*>(part + k * per + i4);
  } else {
    for (int e = 0; e < 4; ++e)
      if (i4 + e < per)
        for (int k = sl; k < splits; k += 16) s[e] += part[k * per + i4 + e];
  }
  red[sl][q] = s;
  __syncthreads();
  if (sl == 0) {
    for (int k = 1; k < 16; ++k) s += red[k][q];
    for (int e = 0; e < 4; ++e) {
      const long i = i4 + e;
      if (i >= per) break;
      const int ci = i % cin, co = (i / cin) % cout, tp = i / ((long)cin * cout);
      float* d = dw + ((long)co * cin + ci) * taps + tp;
      *d = accumulate ? *d + s[e] : s[e];
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int taps, int cout,
                                                           int cin, int accumulate, float* __restrict__ dw) {
  wgrad_reduce_body(part, splits, taps, cout, cin, accumulate, dw, blockIdx.x);
}
// the reduces of up to 4 weight gradients (the same-depth convs of the HRNet branches) in one launch
struct WgradReduceMulti {
  const float* part[4];
  float* dw[4];
  int splits[4], taps[4], cout[4], cin[4], accumulate[4];
  unsigned start[5];
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi(const WgradReduceMulti a) {
  const unsigned b = blockIdx.x;
  const int j = __builtin_amdgcn_readfirstlane((b >= a.start[1]) + (b >= a.start[2]) + (b >= a.start[3]));
  wgrad_reduce_body(a.part[j], a.splits[j], a.taps[j], a.cout[j], a.cin[j], a.accumulate[j], a.dw[j], b - a.start[j]);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d, train mode (nn.BatchNorm2d.forward with self.training; pose_hrnet.py:36-57 etc.)
// ---------------------------------------------------------------------------------------------
// Per-channel partial sums of two row functions, fp64, one row of ws per block: ws[block][2C].
//   MODE 0: (x, x*x)                       forward statistics
//   MODE 1: (g, g*xhat), g = dy*(y>0)      backward sums
// Thread = 4 consecutive channels (16-byte loads) x one row lane; C % 4 == 0.
constexpr int kBnMaxBlocks = 4096;   // partial rows: the sums kernel uses <= 512, a conv epilogue one per tile

typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) {
  if constexpr (std::is_same<T, float>::value) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    const bf16x4v v = *reinterpret_cast<const bf16x4v*>(p);       // one 8-byte load
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, f32x4 v) {
  if constexpr (std::is_same<T, float>::value) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    *reinterpret_cast<bf16x4v*>(p) = bf16x4v{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  }
}

// (bx, nbx) = the workgroup's index / the number of workgroups working on THIS tensor: the single-tensor kernels
// pass blockIdx.x / gridDim.x, the multi-tensor kernels (bn_multi_*) the position inside the tensor's block range
template <typename T, int MODE>
__device__ __forceinline__ void bn_sums_body(const T* __restrict__ x, const T* __restrict__ dy,
                                             const T* __restrict__ y, const float* __restrict__ mean,
                                             const float* __restrict__ invstd, long m, int C,
                                             double* __restrict__ ws, unsigned bx, unsigned nbx) {
  __shared__ double red[256][8];
  const int t = threadIdx.x;
  const int CG = C >> 2;                       // channel quads
  const int CT = CG < 256 ? CG : 256, RG = 256 / CT;
  const long rows_per_block = (m + nbx - 1) / nbx;
  const long r0 = (long)bx * rows_per_block, r1 = min(r0 + rows_per_block, m);
  double* out = ws + (long)bx * 2 * C;
  for (int cb = 0; cb < CG; cb += CT) {
    const int cg = cb + t % CT;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (t < RG * CT && cg < CG) {
      f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
      if (MODE == 1) {
        mu = *reinterpret_cast<const f32x4*>(mean + cg * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + cg * 4);
      }
#pragma unroll 4
      for (long r = r0 + t / CT; r < r1; r += RG) {
        const f32x4 xv = ld4(x + r * C + cg * 4);
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0[e] += xv[e];
            s1[e] += (double)xv[e] * xv[e];
          }
        } else {
          f32x4 g = ld4(dy + r * C + cg * 4);
          if (y) {
            const f32x4 yv = ld4(y + r * C + cg * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!(yv[e] > 0.f)) g[e] = 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0[e] += g[e];
            s1[e] += (double)g[e] * ((xv[e] - mu[e]) * is[e]);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[t][e] = s0[e];
      red[t][4 + e] = s1[e];
    }
    __syncthreads();
    if (t < CT && cb + t < CG) {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += red[g * CT + t][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        out[(cb + t) * 4 + e] = a[e];
        out[C + (cb + t) * 4 + e] = a[4 + e];
      }
    }
    __syncthreads();
  }
}
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_sums_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                      const T* __restrict__ y, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, long m, int C,
                                                      double* __restrict__ ws) {
  bn_sums_body<T, MODE>(x, dy, y, mean, invstd, m, C, ws, blockIdx.x, gridDim.x);
}

// Sum of the per-block partial rows for 4 channels per workgroup: 64 row lanes per channel, then a
// fixed-order LDS reduction (deterministic).  Valid in threads with kl == 0.
__device__ __forceinline__ void bn_collect(const double* ws, int nblocks, int C, int c, double& a, double& b) {
  __shared__ double red[2][64][5];
  const int cl = threadIdx.x & 3, kl = threadIdx.x >> 2;
  a = 0.0;
  b = 0.0;
  if (c < C)
    for (int k = kl; k < nblocks; k += 64) {
      a += ws[(long)k * 2 * C + c];
      b += ws[(long)k * 2 * C + C + c];
    }
  red[0][kl][cl] = a;
  red[1][kl][cl] = b;
  __syncthreads();
  if (kl == 0) {
    for (int k = 1; k < 64; ++k) {
      a += red[0][k][cl];
      b += red[1][k][cl];
    }
  }
}

__device__ __forceinline__ void bn_fwd_finalize_body(double* __restrict__ ws, int nblocks, long m, int C, float eps, float momentum,
                                                     float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ smean,
                                                     float* __restrict__ sinvstd, unsigned bx) {
  const int c = bx * 4 + (threadIdx.x & 3);
  double s0, s1;
  bn_collect(ws, nblocks, C, c, s0, s1);
  if (c >= C || (threadIdx.x >> 2)) return;
  const double mu = s0 / (double)m;
  double var = s1 / (double)m - mu * mu;
  if (var < 0.0) var = 0.0;
  smean[c] = (float)mu;
  sinvstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(m > 1 ? var * (double)m / (double)(m - 1) : var);
}
__global__ void bn_fwd_finalize_kernel(double* __restrict__ ws, int nblocks, long m, int C, float eps, float momentum,
                                       float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ smean,
                                       float* __restrict__ sinvstd) {
  bn_fwd_finalize_body(ws, nblocks, m, C, eps, momentum, rmean, rvar, smean, sinvstd, blockIdx.x);
}

// y = [relu]((x - mean) * invstd * gamma + beta [+ res]); 4 channels per thread
template <typename T>
__device__ __forceinline__ void bn_apply_body(const T* __restrict__ x, const T* __restrict__ res,
                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              long total4, int C, int relu, T* __restrict__ y, unsigned bx, unsigned nbx) {
  for (long i = (long)bx * 256 + threadIdx.x; i < total4; i += (long)nbx * 256) {
    const int c = (int)((i * 4) % C);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 v = (ld4(x + i * 4) - mu) * is * ga + be;
    if (res) v += ld4(res + i * 4);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    st4(y + i * 4, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       long total4, int C, int relu, T* __restrict__ y) {
  bn_apply_body<T>(x, res, mean, invstd, gamma, beta, total4, C, relu, y, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void bn_bwd_finalize_body(double* __restrict__ ws, int nblocks, int C, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ sums, unsigned bx) {
  const int c = bx * 4 + (threadIdx.x & 3);
  double s0, s1;
  bn_collect(ws, nblocks, C, c, s0, s1);
  if (c >= C || (threadIdx.x >> 2)) return;
  dbeta[c] = (float)s0;
  dgamma[c] = (float)s1;
  sums[c] = (float)s0;
  sums[C + c] = (float)s1;
}
__global__ void bn_bwd_finalize_kernel(double* __restrict__ ws, int nblocks, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ sums) {
  bn_bwd_finalize_body(ws, nblocks, C, dgamma, dbeta, sums, blockIdx.x);
}

// g = dy*(y>0);  dx = gamma*invstd*(g - dbeta/m - xhat*dgamma/m);  optional g_out = g
template <typename T>
__device__ __forceinline__ void bn_bwd_apply_body(const T* __restrict__ x, const T* __restrict__ dy,
                                                  const T* __restrict__ y, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ sums, long total4, int C, float inv_m,
                                                  T* __restrict__ dx, T* __restrict__ g_out, unsigned bx, unsigned nbx) {
  for (long i = (long)bx * 256 + threadIdx.x; i < total4; i += (long)nbx * 256) {
    const int c = (int)((i * 4) % C);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 db = *reinterpret_cast<const f32x4*>(sums + c), dg = *reinterpret_cast<const f32x4*>(sums + C + c);
    f32x4 g = ld4(dy + i * 4);
    if (y) {
      const f32x4 yv = ld4(y + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(yv[e] > 0.f)) g[e] = 0.f;
    }
    const f32x4 xh = (ld4(x + i * 4) - mu) * is;
    st4(dx + i * 4, ga * is * (g - db * inv_m - xh * dg * inv_m));
    if (g_out) st4(g_out + i * 4, g);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, long total4, int C, float inv_m,
                                                           T* __restrict__ dx, T* __restrict__ g_out) {
  bn_bwd_apply_body<T>(x, dy, y, mean, invstd, gamma, sums, total4, C, inv_m, dx, g_out, blockIdx.x, gridDim.x);
}

// ---- the same five passes over up to 4 tensors per launch (the BatchNorms of one block depth of all HRNet
// branches are independent: train.py runs the branches in lock step).  A workgroup finds its tensor from the
// block-range table; per-tensor arithmetic and block partition are those of the single-tensor kernels, so the
// results are bit-identical.
struct BnMulti {
  udp_bn_item it[4];
  unsigned start[5];       // first block of tensor j in this launch; start[n..4] = total
  float eps, momentum;
};
#define UDP_BN_PICK                                                                              \
  const unsigned b = blockIdx.x;                                                                   \
  const int j = __builtin_amdgcn_readfirstlane((b >= a.start[1]) + (b >= a.start[2]) + (b >= a.start[3])); \
  const udp_bn_item& q = a.it[j];                                                                  \
  const unsigned bx = b - a.start[j], nbx = a.start[j + 1] - a.start[j];                           \
  (void)nbx;
template <typename T>
__global__ __launch_bounds__(256) void bn_multi_sums_bwd(const BnMulti a) {
  UDP_BN_PICK
  bn_sums_body<T, 1>((const T*)q.x, (const T*)q.dy, (const T*)q.y_relu, q.save_mean, q.save_invstd, q.m, q.c, q.ws, bx, nbx);
}
__global__ void bn_multi_fin_fwd(const BnMulti a) {
  UDP_BN_PICK
  bn_fwd_finalize_body(q.ws, q.rows, q.m, q.c, a.eps, a.momentum, q.running_mean, q.running_var, q.save_mean, q.save_invstd, bx);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_multi_apply_fwd(const BnMulti a) {
  UDP_BN_PICK
  bn_apply_body<T>((const T*)q.x, (const T*)q.res, q.save_mean, q.save_invstd, q.gamma, q.beta, q.m * q.c / 4, q.c, q.relu, (T*)q.y, bx, nbx);
}
__global__ void bn_multi_fin_bwd(const BnMulti a) {
  UDP_BN_PICK
  bn_bwd_finalize_body(q.ws, q.rows, q.c, q.dgamma, q.dbeta, reinterpret_cast<float*>(q.ws + (size_t)kBnMaxBlocks * 2 * q.c), bx);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_multi_apply_bwd(const BnMulti a) {
  UDP_BN_PICK
  bn_bwd_apply_body<T>((const T*)q.x, (const T*)q.dy, (const T*)q.y_relu, q.save_mean, q.save_invstd, q.gamma,
                       reinterpret_cast<const float*>(q.ws + (size_t)kBnMaxBlocks * 2 * q.c), q.m * q.c / 4, q.c, 1.f / (float)q.m,
                       (T*)q.dx, (T*)q.g_out, bx, nbx);
}
#undef UDP_BN_PICK

// ---------------------------------------------------------------------------------------------
// element-wise nodes
// ---------------------------------------------------------------------------------------------
// acc[n,y,x,c] (init ? = : +=) src[n, y>>s, x>>s, c];  optional ReLU on the result
template <typename T>
__global__ __launch_bounds__(256) void ew_accumulate_kernel(T* __restrict__ acc, const T* __restrict__ src, long total,
                                                            int h, int w, int c, int shift, int init, int relu) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long j = i;
    if (shift) {
      const int ch = i % c;
      long r = i / c;
      const int x = r % w;
      r /= w;
      const int y = r % h;
      const long n = r / h;
      j = ((n * (h >> shift) + (y >> shift)) * (w >> shift) + (x >> shift)) * c + ch;
    }
    float v = tof(src[j]);
    if (!init) v += tof(acc[i]);
    if (relu) v = fmaxf(v, 0.f);
    acc[i] = fromf<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, long total,
                                                       T* __restrict__ g) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    g[i] = tof(y[i]) > 0.f ? dy[i] : fromf<T>(0.f);
}

// du[n,y,x,c] (+)= sum over the 2^s x 2^s block of g   (backward of nearest upsampling)
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ g, long total_out, int h, int w, int c,
                                                           int shift, int accumulate, T* __restrict__ du) {
  const int hs = h >> shift, ws_ = w >> shift, f = 1 << shift;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_out; i += (long)gridDim.x * 256) {
    const int ch = i % c;
    long r = i / c;
    const int x = r % ws_;
    r /= ws_;
    const int y = r % hs;
    const long n = r / hs;
    float s = 0.f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) s += tof(g[((n * h + (y * f + dy)) * w + (x * f + dx)) * c + ch]);
    if (accumulate) s += tof(du[i]);
    du[i] = fromf<T>(s);
  }
}

// db[c] = sum_rows g[row*pitch + c]
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ g, long m, int pitch, int C,
                                                        float* __restrict__ db) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (long r = threadIdx.x; r < m; r += 256) s += tof(g[r * pitch + c]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && c < C) db[c] = (float)red[0];
}

// NCHW fp32 [n,c,h,w] -> NHWC [n,h,w,c_pad] (zero filled channels c..c_pad)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, long total, int c, int hw,
                                                           int c_pad, T* __restrict__ dst) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = i % c_pad;
    const long r = i / c_pad;
    const long n = r / hw, px = r % hw;
    dst[i] = fromf<T>(ch < c ? src[(n * c + ch) * hw + px] : 0.f);
  }
}

// torch.optim.Adam single-tensor rule (betas, eps; no weight decay, no amsgrad): lib/utils/utils.py:70-74
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long count, float b1, float b2, float eps,
                                                   float step_size, float inv_sqrt_bc2, float grad_scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const float gi = grad_scale == 1.f ? g[i] : g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

// The same with the two step-dependent scalars read from device memory, so that a captured launch (hipGraph
// replay of the whole training step) follows the step count: coef = {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)}
__global__ __launch_bounds__(256) void adam_kernel_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long count, float b1, float b2, float eps,
                                                       const float* __restrict__ coef, float grad_scale) {
  const float step_size = coef[0], inv_sqrt_bc2 = coef[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const float gi = grad_scale == 1.f ? g[i] : g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

#define UDP_DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
  do {                                             \
    if ((dtype) == UDP_F32) {                      \
      CALL_F32;                                    \
    } else {                                       \
      CALL_BF16;                                   \
    }                                              \
  } while (0)

static int check_dtype(int dtype, const char* who) {
  if (dtype != UDP_F32 && dtype != UDP_BF16) return fail(UDP_ERR_ARG, "%s: dtype %d", who, dtype);
  return UDP_OK;
}
// blocks of the BatchNorm partial-sum pass: >= 8 rows per row lane, at most kBnMaxBlocks
static unsigned bn_blocks(long m, int c) {
  const int cg = c / 4, ct = cg < 256 ? cg : 256, rg = 256 / ct;
  long nb = m / ((long)rg * 8);
  if (nb < 1) nb = 1;
  return (unsigned)(nb > kBnMaxBlocks ? kBnMaxBlocks : nb);
}
static int launched(const char* who) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UDP_ERR_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
  return UDP_OK;
}

}  // namespace udp

using namespace udp;

extern "C" int udp_pack_conv_weights(const float* w, int cout, int cin, int ks, int dtype, void* w_fwd, void* w_dgrad,
                                     void* stream) {
  if (!w || !w_fwd) return fail(UDP_ERR_ARG, "udp_pack_conv_weights: null pointer");
  if (cout <= 0 || cin <= 0 || (ks != 1 && ks != 3)) return fail(UDP_ERR_ARG, "udp_pack_conv_weights: shape");
  if (check_dtype(dtype, "udp_pack_conv_weights")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int cout_pad = rup(cout, 32), cin_k = rup(cin, 16), cin_pad = rup(cin, 32), cout_k = rup(cout, 16);
  const long total = (long)ks * ks * ((long)cout_pad * cin_k + (w_dgrad ? (long)cin_pad * cout_k : 0));
  UDP_DISPATCH_T(dtype,
                 (pack_weights_kernel<float><<<nblocks(total, 256, 4096), 256, 0, s>>>(
                     w, cout, cin, ks, cout_pad, cin_k, cin_pad, cout_k, (float*)w_fwd, (float*)w_dgrad)),
                 (pack_weights_kernel<__bf16><<<nblocks(total, 256, 4096), 256, 0, s>>>(
                     w, cout, cin, ks, cout_pad, cin_k, cin_pad, cout_k, (__bf16*)w_fwd, (__bf16*)w_dgrad)));
  return launched("udp_pack_conv_weights");
}

extern "C" int udp_pack_conv_weights_batch(const udp_pack_desc* descs_dev, int n, int dtype, void* stream) {
  if (!descs_dev || n <= 0) return fail(UDP_ERR_ARG, "udp_pack_conv_weights_batch: argument");
  if (check_dtype(dtype, "udp_pack_conv_weights_batch")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(64, n);
  UDP_DISPATCH_T(dtype, (pack_weights_batch_kernel<float><<<grid, 256, 0, s>>>(descs_dev)),
                 (pack_weights_batch_kernel<__bf16><<<grid, 256, 0, s>>>(descs_dev)));
  return launched("udp_pack_conv_weights_batch");
}

extern "C" int udp_zero_stuff2(const void* dy, int n, int h, int w, int c, int dtype, void* out, void* stream) {
  if (!dy || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0) return fail(UDP_ERR_ARG, "udp_zero_stuff2: argument");
  if (check_dtype(dtype, "udp_zero_stuff2")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)n * 4 * h * w * c;
  UDP_DISPATCH_T(dtype, (zero_stuff2_kernel<float><<<nblocks(total, 256 * 4), 256, 0, s>>>((const float*)dy, total, h, w, c, (float*)out)),
                 (zero_stuff2_kernel<__bf16><<<nblocks(total, 256 * 4), 256, 0, s>>>((const __bf16*)dy, total, h, w, c, (__bf16*)out)));
  return launched("udp_zero_stuff2");
}

extern "C" size_t udp_conv2d_wgrad_workspace_bytes(int cout, int cin, int ks) {
  // enough for 64 splits of the smallest layers, never less than 4 splits of the largest
  const size_t per = (size_t)ks * ks * cout * cin * sizeof(float);
  size_t want = per * 64;
  const size_t cap = (size_t)64 << 20;
  if (want > cap) want = cap;
  if (want < per * 4) want = per * 4;
  return want;
}

static int wgrad_partials(const void* x, const void* dy, int n, int hin, int win, int cin_k, int hout, int wout,
                          int cout_k, int ks, int stride, int cout, int cin, int dtype, float* dw,
                          void* workspace, size_t workspace_bytes, void* stream, int* splits_out) {
  if (!x || !dy || !dw || !workspace) return fail(UDP_ERR_ARG, "udp_conv2d_wgrad: null pointer");
  if (check_dtype(dtype, "udp_conv2d_wgrad")) return UDP_ERR_ARG;
  if (n <= 0 || (ks != 1 && ks != 3) || (stride != 1 && stride != 2) || cout <= 0 || cin <= 0 || cout > cout_k ||
      cin > cin_k || (cin_k & 3) || (cout_k & 3))
    return fail(UDP_ERR_ARG, "udp_conv2d_wgrad: shape");
  const int pad = ks / 2;
  if (hout != (hin + 2 * pad - ks) / stride + 1 || wout != (win + 2 * pad - ks) / stride + 1)
    return fail(UDP_ERR_ARG, "udp_conv2d_wgrad: output size does not match input/stride");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.dy = dy;
  p.part = reinterpret_cast<float*>(workspace);
  p.N = n; p.Hin = hin; p.Win = win; p.CinK = cin_k; p.Hout = hout; p.Wout = wout; p.CoutK = cout_k; p.stride = stride;
  p.cout = cout; p.cin = cin;
  const bool bf = dtype == UDP_BF16 && getenv("UDP_POSE_WGRAD_F32MFMA") == nullptr;
  // fp32 kernel: <= 64 pixels per unit; bf16 kernel: up to 256 pixels / 512 staged rows per unit (fewer, larger
  // global round trips: its MFMA time per unit is far below one HBM latency)
  const int maxpx = bf ? 256 : (stride == 1 ? 64 : 32);
  const int nseg = (wout + (bf ? 63 : maxpx - 1)) / (bf ? 64 : maxpx);
  p.TW = (wout + nseg - 1) / nseg;
  p.TH = 1;
  for (int th = maxpx / p.TW; th > 1; --th) {
    const int rows = ((th * p.TW + 15) & ~15) + ((th - 1) * stride + ks) * ((p.TW - 1) * stride + ks);
    if (hout % th == 0 && (!bf || rows <= kWbItems * 64)) {
      p.TH = th;
      break;
    }
  }
  p.units_x = (wout + p.TW - 1) / p.TW;
  p.units_y = (hout + p.TH - 1) / p.TH;
  p.units = n * p.units_x * p.units_y;
  p.XH = (p.TH - 1) * stride + ks;
  p.XW = (p.TW - 1) * stride + ks;
  p.tiles_ci = (cin + 31) / 32;
  const int tiles = p.tiles_ci * ((cout + 31) / 32);
  const size_t per = (size_t)ks * ks * cout * cin * sizeof(float);
  // fp32 kernel: 4 workgroups per CU; bf16 kernel (one resident workgroup per CU, 144 accumulator AGPRs): 1 per CU
  static const int env_wgs = getenv("UDP_POSE_WGRAD_WGS") ? atoi(getenv("UDP_POSE_WGRAD_WGS")) : 0;
  const int target_wgs = env_wgs > 0 ? env_wgs : (bf ? 256 : 1024);
  int splits = (target_wgs + tiles - 1) / tiles;
  if (splits > p.units) splits = p.units;
  if ((size_t)splits * per > workspace_bytes) splits = (int)(workspace_bytes / per);
  if (splits < 1) return fail(UDP_ERR_ARG, "udp_conv2d_wgrad: workspace of %zu bytes is smaller than one partial (%zu)", workspace_bytes, per);
  p.upw = (p.units + splits - 1) / splits;
  splits = (p.units + p.upw - 1) / p.upw;
  const int pxp = bf ? (p.TH * p.TW + 15) & ~15 : (p.TH * p.TW + 3) & ~3;
  unsigned lds = (unsigned)((pxp + p.XH * p.XW) * kWgPitch * sizeof(float));
  if (bf) {
    const int rows = pxp + p.XH * p.XW;
    if (rows * 4 > kWbItems * 256) return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_wgrad: tile of %d rows exceeds the staging plan", rows);
    if ((cin_k & 7) || (cout_k & 7)) return fail(UDP_ERR_ARG, "udp_conv2d_wgrad: bf16 channel pitches must be multiples of 8");
    lds = (unsigned)(rows * kWbPitch);
    if (lds < 4 * 1024 * sizeof(float)) lds = 4 * 1024 * sizeof(float);     // the cross-wave reduction buffer
  }
  if (lds > 160 * 1024) return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_wgrad: tile needs %u bytes of LDS", lds);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(tiles, splits);
  const void* fn;
  if (dtype == UDP_F32)
    fn = ks == 3 ? reinterpret_cast<const void*>(&wgrad_kernel<float, 3>) : reinterpret_cast<const void*>(&wgrad_kernel<float, 1>);
  else if (bf)
    fn = ks == 3 ? reinterpret_cast<const void*>(&wgrad_bf16_kernel<3>) : reinterpret_cast<const void*>(&wgrad_bf16_kernel<1>);
  else
    fn = ks == 3 ? reinterpret_cast<const void*>(&wgrad_kernel<__bf16, 3>) : reinterpret_cast<const void*>(&wgrad_kernel<__bf16, 1>);
  if (lds > 64 * 1024) UDP_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  void* args[] = {&p};
  UDP_HIP_CHECK(hipLaunchKernel(fn, grid, dim3(256), args, lds, s));
  *splits_out = splits;
  return UDP_OK;
}

extern "C" int udp_conv2d_wgrad(const void* x, const void* dy, int n, int hin, int win, int cin_k, int hout, int wout,
                                int cout_k, int ks, int stride, int cout, int cin, int dtype, float* dw, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream) {
  int splits = 0;
  const int rc = wgrad_partials(x, dy, n, hin, win, cin_k, hout, wout, cout_k, ks, stride, cout, cin, dtype, dw, workspace,
                                workspace_bytes, stream, &splits);
  if (rc) return rc;
  const long total = (long)ks * ks * cout * cin;
  wgrad_reduce_kernel<<<nblocks(total, 64), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
      reinterpret_cast<const float*>(workspace), splits, ks * ks, cout, cin, accumulate, dw);
  return launched("udp_conv2d_wgrad");
}

extern "C" int udp_conv2d_wgrad_group(const udp_wgrad_item* items, int n_items, int dtype, void* stream) {
  if (!items || n_items < 1 || n_items > 4) return fail(UDP_ERR_ARG, "udp_conv2d_wgrad_group: %d members (1..4)", n_items);
  WgradReduceMulti a;
  memset(&a, 0, sizeof(a));
  unsigned tot = 0;
  for (int j = 0; j < n_items; ++j) {
    const udp_wgrad_item& q = items[j];
    for (int k = 0; k < j; ++k)
      if (items[k].workspace == q.workspace) return fail(UDP_ERR_ARG, "udp_conv2d_wgrad_group: members %d and %d share a workspace", k, j);
    int splits = 0;
    const int rc = wgrad_partials(q.x, q.dy, q.n, q.hin, q.win, q.cin_k, q.hout, q.wout, q.cout_k, q.ks, q.stride, q.cout, q.cin,
                                  dtype, q.dw, q.workspace, q.workspace_bytes, stream, &splits);
    if (rc) return rc;
    a.part[j] = reinterpret_cast<const float*>(q.workspace);
    a.dw[j] = q.dw;
    a.splits[j] = splits;
    a.taps[j] = q.ks * q.ks;
    a.cout[j] = q.cout;
    a.cin[j] = q.cin;
    a.accumulate[j] = q.accumulate;
    a.start[j] = tot;
    tot += nblocks((long)q.ks * q.ks * q.cout * q.cin, 64);
  }
  for (int j = n_items; j < 5; ++j) a.start[j] = j < 4 ? 0xFFFFFFFFu : tot;
  a.start[4] = tot;
  wgrad_reduce_multi<<<tot, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(a);
  return launched("udp_conv2d_wgrad_group");
}

extern "C" size_t udp_bn_workspace_doubles(int c) { return c > 0 ? (size_t)(kBnMaxBlocks + 1) * 2 * c : 0; }

extern "C" int udp_bn_train_fwd(const void* x, int64_t m, int c, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var, float* save_mean,
                                float* save_invstd, const void* res, int relu, void* y, int dtype, double* ws,
                                void* stream) {
  if (!x || !gamma || !beta || !save_mean || !save_invstd || !y || !ws) return fail(UDP_ERR_ARG, "udp_bn_train_fwd: null pointer");
  if (m <= 0 || c <= 0 || (c & 3)) return fail(UDP_ERR_ARG, "udp_bn_train_fwd: m=%lld c=%d (c must be a multiple of 4)", (long long)m, c);
  if (check_dtype(dtype, "udp_bn_train_fwd")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = bn_blocks(m, c);
  UDP_DISPATCH_T(dtype,
                 (bn_sums_kernel<float, 0><<<nb, 256, 0, s>>>((const float*)x, nullptr, nullptr, nullptr, nullptr, m, c, ws)),
                 (bn_sums_kernel<__bf16, 0><<<nb, 256, 0, s>>>((const __bf16*)x, nullptr, nullptr, nullptr, nullptr, m, c, ws)));
  bn_fwd_finalize_kernel<<<(c + 3) / 4, 256, 0, s>>>(ws, (int)nb, m, c, eps, momentum, running_mean, running_var, save_mean, save_invstd);
  const long total4 = m * c / 4;
  UDP_DISPATCH_T(dtype,
                 (bn_apply_kernel<float><<<nblocks(total4, 256 * 2), 256, 0, s>>>((const float*)x, (const float*)res, save_mean, save_invstd, gamma, beta, total4, c, relu, (float*)y)),
                 (bn_apply_kernel<__bf16><<<nblocks(total4, 256 * 2), 256, 0, s>>>((const __bf16*)x, (const __bf16*)res, save_mean, save_invstd, gamma, beta, total4, c, relu, (__bf16*)y)));
  return launched("udp_bn_train_fwd");
}

extern "C" int udp_bn_rows_max(void) { return kBnMaxBlocks; }

extern "C" int udp_bn_train_fwd_from_sums(const void* x, int64_t m, int c, const float* gamma, const float* beta, float eps,
                                          float momentum, float* running_mean, float* running_var, float* save_mean,
                                          float* save_invstd, const void* res, int relu, void* y, int dtype, double* ws,
                                          int rows, void* stream) {
  if (!x || !gamma || !beta || !save_mean || !save_invstd || !y || !ws) return fail(UDP_ERR_ARG, "udp_bn_train_fwd_from_sums: null pointer");
  if (m <= 0 || c <= 0 || (c & 3)) return fail(UDP_ERR_ARG, "udp_bn_train_fwd_from_sums: m=%lld c=%d (c must be a multiple of 4)", (long long)m, c);
  if (rows <= 0 || rows > kBnMaxBlocks) return fail(UDP_ERR_ARG, "udp_bn_train_fwd_from_sums: %d partial rows (1..%d)", rows, kBnMaxBlocks);
  if (check_dtype(dtype, "udp_bn_train_fwd_from_sums")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  bn_fwd_finalize_kernel<<<(c + 3) / 4, 256, 0, s>>>(ws, rows, m, c, eps, momentum, running_mean, running_var, save_mean, save_invstd);
  const long total4 = m * c / 4;
  UDP_DISPATCH_T(dtype,
                 (bn_apply_kernel<float><<<nblocks(total4, 256 * 2), 256, 0, s>>>((const float*)x, (const float*)res, save_mean, save_invstd, gamma, beta, total4, c, relu, (float*)y)),
                 (bn_apply_kernel<__bf16><<<nblocks(total4, 256 * 2), 256, 0, s>>>((const __bf16*)x, (const __bf16*)res, save_mean, save_invstd, gamma, beta, total4, c, relu, (__bf16*)y)));
  return launched("udp_bn_train_fwd_from_sums");
}

extern "C" int udp_bn_train_bwd(const void* x, const void* dy, const void* y_relu, int64_t m, int c, const float* gamma,
                                const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* dx,
                                void* g_out, int dtype, double* ws, void* stream) {
  if (!x || !dy || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !dx || !ws)
    return fail(UDP_ERR_ARG, "udp_bn_train_bwd: null pointer");
  if (m <= 0 || c <= 0 || (c & 3)) return fail(UDP_ERR_ARG, "udp_bn_train_bwd: m=%lld c=%d (c must be a multiple of 4)", (long long)m, c);
  if (check_dtype(dtype, "udp_bn_train_bwd")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = bn_blocks(m, c);
  UDP_DISPATCH_T(dtype,
                 (bn_sums_kernel<float, 1><<<nb, 256, 0, s>>>((const float*)x, (const float*)dy, (const float*)y_relu, save_mean, save_invstd, m, c, ws)),
                 (bn_sums_kernel<__bf16, 1><<<nb, 256, 0, s>>>((const __bf16*)x, (const __bf16*)dy, (const __bf16*)y_relu, save_mean, save_invstd, m, c, ws)));
  float* sums = reinterpret_cast<float*>(ws + (size_t)kBnMaxBlocks * 2 * c);      // fp32 {dbeta, dgamma} after the partial rows
  bn_bwd_finalize_kernel<<<(c + 3) / 4, 256, 0, s>>>(ws, (int)nb, c, dgamma, dbeta, sums);
  const long total4 = m * c / 4;
  const float inv_m = 1.f / (float)m;
  UDP_DISPATCH_T(dtype,
                 (bn_bwd_apply_kernel<float><<<nblocks(total4, 256 * 2), 256, 0, s>>>((const float*)x, (const float*)dy, (const float*)y_relu, save_mean, save_invstd, gamma, sums, total4, c, inv_m, (float*)dx, (float*)g_out)),
                 (bn_bwd_apply_kernel<__bf16><<<nblocks(total4, 256 * 2), 256, 0, s>>>((const __bf16*)x, (const __bf16*)dy, (const __bf16*)y_relu, save_mean, save_invstd, gamma, sums, total4, c, inv_m, (__bf16*)dx, (__bf16*)g_out)));
  return launched("udp_bn_train_bwd");
}

static int bn_multi_check(const udp_bn_item* items, int n, int dtype, const char* who) {
  if (!items || n < 1 || n > 4) return fail(UDP_ERR_ARG, "%s: %d tensors (1..4)", who, n);
  if (check_dtype(dtype, who)) return UDP_ERR_ARG;
  for (int j = 0; j < n; ++j) {
    const udp_bn_item& q = items[j];
    if (!q.x || !q.gamma || !q.save_mean || !q.save_invstd || !q.ws) return fail(UDP_ERR_ARG, "%s: tensor %d: null pointer", who, j);
    if (q.m <= 0 || q.c <= 0 || (q.c & 3)) return fail(UDP_ERR_ARG, "%s: tensor %d: m=%lld c=%d (c must be a multiple of 4)", who, j, (long long)q.m, q.c);
    if (q.rows < 0 || q.rows > kBnMaxBlocks) return fail(UDP_ERR_ARG, "%s: tensor %d: %d partial rows (0..%d)", who, j, q.rows, kBnMaxBlocks);
  }
  return UDP_OK;
}

extern "C" int udp_bn_train_fwd_multi(const udp_bn_item* items, int n, float eps, float momentum, int dtype, void* stream) {
  if (bn_multi_check(items, n, dtype, "udp_bn_train_fwd_multi")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnMulti a{};
  a.eps = eps;
  a.momentum = momentum;
  for (int j = 0; j < n; ++j) {
    a.it[j] = items[j];
    if (!a.it[j].beta || !a.it[j].y) return fail(UDP_ERR_ARG, "udp_bn_train_fwd_multi: tensor %d: null pointer", j);
    if (a.it[j].rows == 0) {       // no conv-epilogue statistics for this one: its own partial-sum pass
      const unsigned nb = bn_blocks(a.it[j].m, a.it[j].c);
      UDP_DISPATCH_T(dtype,
                     (bn_sums_kernel<float, 0><<<nb, 256, 0, s>>>((const float*)a.it[j].x, nullptr, nullptr, nullptr, nullptr, a.it[j].m, a.it[j].c, a.it[j].ws)),
                     (bn_sums_kernel<__bf16, 0><<<nb, 256, 0, s>>>((const __bf16*)a.it[j].x, nullptr, nullptr, nullptr, nullptr, a.it[j].m, a.it[j].c, a.it[j].ws)));
      a.it[j].rows = (int)nb;
    }
  }
  unsigned tot = 0;
  for (int j = 0; j < 5; ++j) {
    a.start[j] = tot;
    if (j < n) tot += (unsigned)(a.it[j].c + 3) / 4;
  }
  bn_multi_fin_fwd<<<tot, 256, 0, s>>>(a);
  tot = 0;
  for (int j = 0; j < 5; ++j) {
    a.start[j] = tot;
    if (j < n) tot += nblocks(a.it[j].m * a.it[j].c / 4, 256 * 2);
  }
  UDP_DISPATCH_T(dtype, (bn_multi_apply_fwd<float><<<tot, 256, 0, s>>>(a)), (bn_multi_apply_fwd<__bf16><<<tot, 256, 0, s>>>(a)));
  return launched("udp_bn_train_fwd_multi");
}

extern "C" int udp_bn_train_bwd_multi(const udp_bn_item* items, int n, int dtype, void* stream) {
  if (bn_multi_check(items, n, dtype, "udp_bn_train_bwd_multi")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnMulti a{};
  unsigned tot = 0;
  for (int j = 0; j < n; ++j) {
    a.it[j] = items[j];
    if (!a.it[j].dy || !a.it[j].dgamma || !a.it[j].dbeta || !a.it[j].dx) return fail(UDP_ERR_ARG, "udp_bn_train_bwd_multi: tensor %d: null pointer", j);
    a.it[j].rows = (int)bn_blocks(a.it[j].m, a.it[j].c);
  }
  for (int j = 0; j < 5; ++j) {
    a.start[j] = tot;
    if (j < n) tot += (unsigned)a.it[j].rows;
  }
  UDP_DISPATCH_T(dtype, (bn_multi_sums_bwd<float><<<tot, 256, 0, s>>>(a)), (bn_multi_sums_bwd<__bf16><<<tot, 256, 0, s>>>(a)));
  tot = 0;
  for (int j = 0; j < 5; ++j) {
    a.start[j] = tot;
    if (j < n) tot += (unsigned)(a.it[j].c + 3) / 4;
  }
  bn_multi_fin_bwd<<<tot, 256, 0, s>>>(a);
  tot = 0;
  for (int j = 0; j < 5; ++j) {
    a.start[j] = tot;
    if (j < n) tot += nblocks(a.it[j].m * a.it[j].c / 4, 256 * 2);
  }
  UDP_DISPATCH_T(dtype, (bn_multi_apply_bwd<float><<<tot, 256, 0, s>>>(a)), (bn_multi_apply_bwd<__bf16><<<tot, 256, 0, s>>>(a)));
  return launched("udp_bn_train_bwd_multi");
}

extern "C" int udp_ew_accumulate(void* acc, const void* src, int n, int h, int w, int c, int shift, int init, int relu,
                                 int dtype, void* stream) {
  if (!acc || !src || n <= 0 || h <= 0 || w <= 0 || c <= 0) return fail(UDP_ERR_ARG, "udp_ew_accumulate: argument");
  if (shift < 0 || shift > 5 || (h & ((1 << shift) - 1)) || (w & ((1 << shift) - 1)))
    return fail(UDP_ERR_ARG, "udp_ew_accumulate: shift %d does not divide %dx%d", shift, h, w);
  if (check_dtype(dtype, "udp_ew_accumulate")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)n * h * w * c;
  UDP_DISPATCH_T(dtype, (ew_accumulate_kernel<float><<<nblocks(total, 256 * 4), 256, 0, s>>>((float*)acc, (const float*)src, total, h, w, c, shift, init, relu)),
                 (ew_accumulate_kernel<__bf16><<<nblocks(total, 256 * 4), 256, 0, s>>>((__bf16*)acc, (const __bf16*)src, total, h, w, c, shift, init, relu)));
  return launched("udp_ew_accumulate");
}

extern "C" int udp_relu_bwd(const void* dy, const void* y, void* g, int64_t count, int dtype, void* stream) {
  if (!dy || !y || !g || count <= 0) return fail(UDP_ERR_ARG, "udp_relu_bwd: argument");
  if (check_dtype(dtype, "udp_relu_bwd")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UDP_DISPATCH_T(dtype, (relu_bwd_kernel<float><<<nblocks(count, 256 * 4), 256, 0, s>>>((const float*)dy, (const float*)y, count, (float*)g)),
                 (relu_bwd_kernel<__bf16><<<nblocks(count, 256 * 4), 256, 0, s>>>((const __bf16*)dy, (const __bf16*)y, count, (__bf16*)g)));
  return launched("udp_relu_bwd");
}

extern "C" int udp_upsample_bwd(const void* g, int n, int h, int w, int c, int shift, void* du, int accumulate, int dtype,
                                void* stream) {
  if (!g || !du || n <= 0 || h <= 0 || w <= 0 || c <= 0) return fail(UDP_ERR_ARG, "udp_upsample_bwd: argument");
  if (shift < 1 || shift > 5 || (h & ((1 << shift) - 1)) || (w & ((1 << shift) - 1)))
    return fail(UDP_ERR_ARG, "udp_upsample_bwd: shift %d does not divide %dx%d", shift, h, w);
  if (check_dtype(dtype, "udp_upsample_bwd")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)n * (h >> shift) * (w >> shift) * c;
  UDP_DISPATCH_T(dtype, (upsample_bwd_kernel<float><<<nblocks(total, 256), 256, 0, s>>>((const float*)g, total, h, w, c, shift, accumulate, (float*)du)),
                 (upsample_bwd_kernel<__bf16><<<nblocks(total, 256), 256, 0, s>>>((const __bf16*)g, total, h, w, c, shift, accumulate, (__bf16*)du)));
  return launched("udp_upsample_bwd");
}

extern "C" int udp_bias_grad(const void* g, int64_t m, int c_pitch, int c, float* db, int dtype, void* stream) {
  if (!g || !db || m <= 0 || c <= 0 || c > c_pitch) return fail(UDP_ERR_ARG, "udp_bias_grad: argument");
  if (check_dtype(dtype, "udp_bias_grad")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UDP_DISPATCH_T(dtype, (bias_grad_kernel<float><<<c, 256, 0, s>>>((const float*)g, m, c_pitch, c, db)),
                 (bias_grad_kernel<__bf16><<<c, 256, 0, s>>>((const __bf16*)g, m, c_pitch, c, db)));
  return launched("udp_bias_grad");
}

extern "C" int udp_nchw_to_nhwc(const float* src, int n, int c, int h, int w, int c_pad, void* dst, int dtype, void* stream) {
  if (!src || !dst || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_pad < c) return fail(UDP_ERR_ARG, "udp_nchw_to_nhwc: argument");
  if (check_dtype(dtype, "udp_nchw_to_nhwc")) return UDP_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)n * h * w * c_pad;
  UDP_DISPATCH_T(dtype, (nchw_to_nhwc_kernel<float><<<nblocks(total, 256 * 4), 256, 0, s>>>(src, total, c, h * w, c_pad, (float*)dst)),
                 (nchw_to_nhwc_kernel<__bf16><<<nblocks(total, 256 * 4), 256, 0, s>>>(src, total, c, h * w, c_pad, (__bf16*)dst)));
  return launched("udp_nchw_to_nhwc");
}

extern "C" int udp_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || count <= 0 || step < 1) return fail(UDP_ERR_ARG, "udp_adam_step: argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  adam_kernel<<<nblocks(count, 256 * 4), 256, 0, s>>>(p, g, m, v, count, beta1, beta2, eps, (float)((double)lr / bc1),
                                                       (float)(1.0 / sqrt(bc2)), grad_scale);
  return launched("udp_adam_step");
}

extern "C" int udp_adam_coefficients(float lr, float beta1, float beta2, int step, float* coef_host) {
  if (!coef_host || step < 1) return fail(UDP_ERR_ARG, "udp_adam_coefficients: argument");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  coef_host[0] = (float)((double)lr / bc1);
  coef_host[1] = (float)(1.0 / sqrt(bc2));
  return UDP_OK;
}

extern "C" int udp_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t count, float beta1, float beta2,
                                 float eps, const float* coef_dev, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !coef_dev || count <= 0) return fail(UDP_ERR_ARG, "udp_adam_step_dev: argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  adam_kernel_dev<<<nblocks(count, 256 * 4), 256, 0, s>>>(p, g, m, v, count, beta1, beta2, eps, coef_dev, grad_scale);
  return launched("udp_adam_step_dev");
}
