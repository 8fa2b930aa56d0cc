// Device-side helpers shared by the conv translation units (conv.hip, conv_ws.hip): vector types, the split-fp16
// ("f16x2") storage form, the LDS swizzle, buffer-descriptor loads / stores, diagnostic stamps.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace udp {


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Split-fp16 storage ("f16x2", UDP_F16X2): a value x is kept as two fp16 numbers, x ~= hi + lo * 2^-11 with
// hi = fp16(x) and lo = fp16((x - hi) * 2^11): 22 significant bits over fp16's whole normal range (the
// scaled lo never goes subnormal before hi does).  A product of two such numbers runs as three fp16 MFMAs
// (hi*hi into the main accumulator; hi*lo and lo*hi into a second one that is folded in with 2^-11 at the
// end; lo*lo ~ 2^-22 is dropped), i.e. fp32-grade results on the 2.5 PF fp16 matrix pipe instead of the
// 157 TF fp32 one.  Memory layout: per pixel (or weight row) the C hi values, then the C lo values.
// cache-policy bits (aux operand of the buffer builtins: 1 = sc0, 2 = nt, 16 = sc1) of the split-fp16 activation stores /
// residual loads / tile DMA -- diagnostic builds only (-DUDP_H2_STORE_AUX=2 ...); 0 = default policy
#ifndef UDP_H2_STORE_AUX
#define UDP_H2_STORE_AUX 0
#endif
#ifndef UDP_H2_RES_AUX
#define UDP_H2_RES_AUX 0
#endif
#ifndef UDP_H2_DMA_AUX
#define UDP_H2_DMA_AUX 0
#endif
struct H2 {};
#ifndef UDP_WS_AD
#define UDP_WS_AD 3     // A-fragment ring depth of the weight-stationary kernels (4 / 5: 245 / 256 registers, -0.6 / -1.5 %)
#endif
#ifndef UDP_WS_DBG
#define UDP_WS_DBG 0   // ablation bits for diagnostic builds (tools/ablate_ws.sh): 1 no A prefetch, 2 no B reads, 4 no DMA after chunk 0, 16 one MFMA per block
#endif
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

constexpr int ROWB = 64;   // LDS row: one pixel's (or one weight row's) 64-byte channel chunk
constexpr int MAXG = 10;   // 16-row staging groups per wave for the input tile (<= 640 rows)

// 16-byte part p of LDS row r is stored at part position p ^ swz<T>(r).  The swizzle makes the MFMA operand
// reads of 16 CONSECUTIVE rows (any start row: the taps shift the window) bank-conflict free:
//   fp32 (ds_read_b64, lane groups = the 32-lane halves: one part of 16 rows): the 4 rows with equal row&3
//        need 4 different positions -> a permutation of (row>>2)&3;
//   bf16 (ds_read_b128, lane groups {0-3,12-15,20-27},...: rows 0-3 and 12-15 with part a, rows 4-11 with
//        part a^1): positions {f(q), f(q+1)^1, f(q+2)^1, f(q+3)} must differ for EVERY q -> f = (0,2,0,2).
//        (The permutation (0,3,2,1) is conflict free only for windows starting at a multiple of 4 rows; PMC
//        SQ_LDS_BANK_CONFLICT showed ~16 % of the conv kernels' cycles with it.)
template <typename T>
__device__ __forceinline__ int swz(int row) {
#ifdef UDP_OLD_SWZ   // A/B build only
  return (-(row >> 2)) & 3;
#else
  if constexpr (std::is_same<T, float>::value) {
    return (-(row >> 2)) & 3;
  } else {
    return (row >> 1) & 2;
  }
#endif
}

#ifdef UDP_STAMPS
// Diagnostic build (libudp_pose_hip_stamps.so, tools/stamp_conv.py): s_memtime stamps per wave into a
// buffer nothing else reads.  Never compiled into the product library.
static __device__ unsigned long long* g_stamps;     // one copy per translation unit (udp_debug_set_stamps sets all)
__device__ __forceinline__ void stamp(int k) {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  if ((threadIdx.x & 63) == 0 && g_stamps) {
    unsigned long long* q = g_stamps + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 4 + (threadIdx.x >> 6)) * 16;
    q[k] = t;
    if (k == 0)   // where the wave runs: HW_REG_XCC_ID (20) in the high word, HW_REG_HW_ID (4: CU_ID[11:8], SE_ID[15:13]) in the low
      q[8] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
  }
}
#define UDP_STAMP(k) stamp(k)
// a value (not a time) into stamp slot k of the wave's record
__device__ __forceinline__ void stamp_val(int k, unsigned long long v) {
  if ((threadIdx.x & 63) == 0 && g_stamps)
    g_stamps[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 4 + (threadIdx.x >> 6)) * 16 + k] = v;
}
#define UDP_STAMP_VAL(k, v) stamp_val(k, v)
#ifdef UDP_STAMPS_WS_ONLY          // tools/stamp_multi.py: only the weight-stationary kernels leave stamps
#define UDP_STAMP_MFMA(k)
#else
#define UDP_STAMP_MFMA(k) stamp(k)
#endif
#else
#define UDP_STAMP(k)
#define UDP_STAMP_VAL(k, v)
#define UDP_STAMP_MFMA(k)
#endif

template <typename T>
struct Tr;
template <>
struct Tr<float> {
  static constexpr int CK = 16;  // input channels per LDS chunk (64 B)
  static constexpr int ESZ = 4;  // bytes per element of one plane
  static constexpr int PL = 1;   // planes per element (H2: hi, lo)
};
template <>
struct Tr<__bf16> {
  static constexpr int CK = 32;
  static constexpr int ESZ = 2;
  static constexpr int PL = 1;
};
template <>
struct Tr<H2> {
  static constexpr int CK = 32;
  static constexpr int ESZ = 2;
  static constexpr int PL = 2;
};

// 4 consecutive channels c..c+3 of pixel `pix` of an NHWC tensor with `pitch` channels per pixel
template <typename T>
__device__ __forceinline__ f32x4 ld4(const void* base, size_t pix, int pitch, int c) {
  if constexpr (std::is_same<T, float>::value) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + pix * pitch + c);
  } else if constexpr (std::is_same<T, __bf16>::value) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + pix * pitch + c);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  } else {
    const _Float16* q = reinterpret_cast<const _Float16*>(base) + pix * (2 * (size_t)pitch) + c;
    const f16x4 hi = *reinterpret_cast<const f16x4*>(q), lo = *reinterpret_cast<const f16x4*>(q + pitch);
    return f32x4{(float)hi[0] + (float)lo[0] * kLoInv, (float)hi[1] + (float)lo[1] * kLoInv,
                 (float)hi[2] + (float)lo[2] * kLoInv, (float)hi[3] + (float)lo[3] * kLoInv};
  }
}
template <typename T>
__device__ __forceinline__ void st4(void* base, size_t pix, int pitch, int c, f32x4 v) {
  if constexpr (std::is_same<T, float>::value) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + pix * pitch + c) = v;
  } else if constexpr (std::is_same<T, __bf16>::value) {
    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + pix * pitch + c) = o;
  } else {
    _Float16* q = reinterpret_cast<_Float16*>(base) + pix * (2 * (size_t)pitch) + c;
    f16x4 hi, lo;
    h2_range_check(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                                   __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (_Float16)v[k];
      lo[k] = (_Float16)((v[k] - (float)hi[k]) * kLoScale);
    }
    *reinterpret_cast<f16x4*>(q) = hi;
    *reinterpret_cast<f16x4*>(q + pitch) = lo;
  }
}

// split / join of the H2 form for 8 consecutive channels held as two accumulator quads
__device__ __forceinline__ void h2_split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
  h2_range_check(__builtin_fmaxf(
      __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1])), __builtin_fmaxf(__builtin_fabsf(a[2]), __builtin_fabsf(a[3]))),
      __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b[0]), __builtin_fabsf(b[1])), __builtin_fmaxf(__builtin_fabsf(b[2]), __builtin_fabsf(b[3])))));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hi[q] = (_Float16)a[q];
    hi[4 + q] = (_Float16)b[q];
    lo[q] = (_Float16)((a[q] - (float)hi[q]) * kLoScale);
    lo[4 + q] = (_Float16)((b[q] - (float)hi[4 + q]) * kLoScale);
  }
}
__device__ __forceinline__ void h2_add8(f32x4& a, f32x4& b, const f16x8 hi, const f16x8 lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q] += (float)hi[q] + (float)lo[q] * kLoInv;
    b[q] += (float)hi[4 + q] + (float)lo[4 + q] * kLoInv;
  }
}

// 4*NB consecutive channels <-> NB accumulator tiles of one lane.  `p` points at the lane's first
// channel (bytes); H2: the lo plane starts lo_off bytes further
template <typename T, int NB>
__device__ __forceinline__ void add_vec(f32x4 (&v)[NB], const unsigned char* p, unsigned lo_off) {
  if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) v[nb] += *reinterpret_cast<const f32x4*>(p + 16 * nb);
  } else if constexpr (std::is_same<T, __bf16>::value) {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(p + 16 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * h][q] += (float)x[q];
        v[2 * h + 1][q] += (float)x[4 + q];
      }
    }
  } else {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h)
      h2_add8(v[2 * h], v[2 * h + 1], *reinterpret_cast<const f16x8*>(p + 16 * h), *reinterpret_cast<const f16x8*>(p + lo_off + 16 * h));
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOff = 0x7FFF0000u;   // voffset of a masked lane: beyond every buffer (< 2 GiB), no wrap with + small

// x / d for x*d < 2^20 with m = ceil(2^20 / d): one full-rate 24-bit multiply and a shift
__device__ __forceinline__ int fdiv20(int x, unsigned m) { return (int)(__umul24((unsigned)x, m) >> 20); }

// LDS-DMA through a buffer descriptor: SGPR base + 32-bit per-lane offset; lanes whose offset is out
// of range get ZEROS written to LDS (checked on MI355X: tools/micro/bl_lds.hip) -> conv zero padding
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, UDP_H2_DMA_AUX);
}

// 4*NB consecutive channels of one pixel <-> the NB accumulator tiles of a lane, through a buffer
// descriptor: masked lanes pass kOobOff (loads return 0, stores are dropped) -> no branches
template <typename T, int NB>
__device__ __forceinline__ void add_vec_buf(f32x4 (&v)[NB], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned lo_off) {
  if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      v[nb] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * nb, 0, 0));
  } else if constexpr (std::is_same<T, H2>::value) {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * h, 0, 0);
      const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff + lo_off + 16 * h, 0, 0);
      h2_add8(v[2 * h], v[2 * h + 1], __builtin_bit_cast(f16x8, hi), __builtin_bit_cast(f16x8, lo));
    }
  } else {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * h, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned wlo = x[q >> 1], whi = x[2 + (q >> 1)];
        v[2 * h][q] += __builtin_bit_cast(float, (q & 1) ? (wlo & 0xFFFF0000u) : (wlo << 16));
        v[2 * h + 1][q] += __builtin_bit_cast(float, (q & 1) ? (whi & 0xFFFF0000u) : (whi << 16));
      }
    }
  }
}
// v += s * (the 8 consecutive split-fp16 channels at voff), s a power of two (exact): the residual of a conv whose
// accumulator carries conv * s
template <int NB>
__device__ __forceinline__ void add_vec_buf_scaled(f32x4 (&v)[NB], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned lo_off, float s) {
  const float sl = s * kLoInv;
#pragma unroll
  for (int h = 0; h < NB / 2; ++h) {
    const f16x8 hi = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * h, 0, 0));
    const f16x8 lo = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff + lo_off + 16 * h, 0, 0));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[2 * h][q] = __builtin_fmaf((float)lo[q], sl, __builtin_fmaf((float)hi[q], s, v[2 * h][q]));
      v[2 * h + 1][q] = __builtin_fmaf((float)lo[4 + q], sl, __builtin_fmaf((float)hi[4 + q], s, v[2 * h + 1][q]));
    }
  }
}
// split form of add_vec_buf: issue the loads early, add later (persistent kernel: the residual of tile t
// flies under tile t's MFMAs)
template <typename T, int NB>
struct ResRegs {
  u32x4 r[std::is_same<T, float>::value ? NB : NB / 2];
};
template <typename T, int NB>
__device__ __forceinline__ void load_res_buf(ResRegs<T, NB>& o, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  constexpr int N = std::is_same<T, float>::value ? NB : NB / 2;
#pragma unroll
  for (int h = 0; h < N; ++h) o.r[h] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * h, 0, 0);
}
template <typename T, int NB>
__device__ __forceinline__ void add_res_regs(f32x4 (&v)[NB], const ResRegs<T, NB>& o) {
  if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) v[nb] += __builtin_bit_cast(f32x4, o.r[nb]);
  } else {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      const u32x4 x = o.r[h];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned wlo = x[q >> 1], whi = x[2 + (q >> 1)];
        v[2 * h][q] += __builtin_bit_cast(float, (q & 1) ? (wlo & 0xFFFF0000u) : (wlo << 16));
        v[2 * h + 1][q] += __builtin_bit_cast(float, (q & 1) ? (whi & 0xFFFF0000u) : (whi << 16));
      }
    }
  }
}
template <typename T, int NB>
__device__ __forceinline__ void store_vec_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned lo_off, const f32x4 (&v)[NB]) {
  if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[nb]), r, voff + 16 * nb, 0, 0);
  } else if constexpr (std::is_same<T, H2>::value) {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      f16x8 hi, lo;
      h2_split8(v[2 * h], v[2 * h + 1], hi, lo);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), r, voff + 16 * h, 0, UDP_H2_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), r, voff + lo_off + 16 * h, 0, UDP_H2_STORE_AUX);
    }
  } else {
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[q] = (__bf16)v[2 * h][q];
        o[4 + q] = (__bf16)v[2 * h + 1][q];
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r, voff + 16 * h, 0, 0);
    }
  }
}

static inline int largest_divisor_leq(int n, int lim) {
  for (int d = lim < n ? lim : n; d >= 1; --d)
    if (n % d == 0) return d;
  return 1;
}

}  // namespace udp
