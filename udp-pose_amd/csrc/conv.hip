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
// the input halo tile and the [tap][cout][cin-chunk] weights in LDS with LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, asynchronous) into 64-byte rows whose
// four 16-byte parts are XOR-swizzled by the row index (conflict-free fragment
// reads; the swizzle is applied on the per-lane SOURCE address because the DMA
// destination is lane-linear).  Halo rows outside the image read a 64-byte zero
// row.  Chunks are double-buffered: chunk c+1 streams in while chunk c feeds the
// MFMAs (one barrier per chunk).  Every wave runs taps x k-steps of MFMA with
// A = weights (rows = cout), B = pixels (cols = pixel); weight rows are permuted
// at staging time so that a lane ends up with 4*NB CONSECUTIVE output channels of
// one pixel -> 16-byte NHWC stores / residual loads in the epilogue.
//   fp32 : v_mfma_f32_16x16x4_f32  (exact fp32 fma chain), ds_read_b64 feeds 2 MFMAs
//   bf16 : v_mfma_f32_16x16x32_bf16 (fp32 accumulate),     ds_read_b128 feeds 1 MFMA
#include "conv_dev.h"

namespace udp {
// One chunk of MFMA work for a wave: all taps x k-steps of its MBW pixel blocks x NB cout blocks.
// Software-pipelined by hand: the fragments of step s+1 are read from LDS before the MFMAs of step s
// are issued, so LDS latency hides behind matrix work even at one or two waves per SIMD (hipcc does not
// hoist the reads across the unrolled taps by itself).
template <typename T, int KS, int NB, int MBW>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[MBW][NB], const unsigned char* sb, const unsigned char* wrow,
                                           const int (&prow)[MBW], int IW, int kg, int wswz) {
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr int BN = NB * 16;
  constexpr int NSTEP = KS * KS * (F32 ? 2 : 1);
  using Frag = typename std::conditional<F32, f32x2, bf16x8>::type;
  Frag wf[2][NB], pf[2][MBW];
  auto load = [&](int s, Frag (&w)[NB], Frag (&x)[MBW]) {
    const int tap = F32 ? s >> 1 : s;
    const int ks = F32 ? (s & 1) : 0;
    const int part = F32 ? 2 * ks + (kg >> 1) : kg;
    const int sub = F32 ? (kg & 1) * 8 : 0;
    const int tap_rows = (tap / KS) * IW + tap % KS;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      w[nb] = *reinterpret_cast<const Frag*>(wrow + (tap * BN + nb * 16) * ROWB + ((part ^ wswz) << 4) + sub);
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
      const int row = prow[i] + tap_rows;
      x[i] = *reinterpret_cast<const Frag*>(sb + row * ROWB + ((part ^ swz<T>(row)) << 4) + sub);
    }
  };
  load(0, wf[0], pf[0]);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if (s + 1 < NSTEP) load(s + 1, wf[(s + 1) & 1], pf[(s + 1) & 1]);
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if constexpr (F32) {
          acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s & 1][nb][0], pf[s & 1][i][0], acc[i][nb], 0, 0, 0);
          acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s & 1][nb][1], pf[s & 1][i][1], acc[i][nb], 0, 0, 0);
        } else {
          acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s & 1][nb], pf[s & 1][i], acc[i][nb], 0, 0, 0);
        }
      }
  }
}

// The same for split-fp16 operands: the hi / lo images of the input tile lie in_lo bytes apart in LDS, those
// of the weights w_lo bytes; per (pixel block, cout block) and tap three v_mfma_f32_16x16x32_f16:
//   acc  += Whi * Xhi            accx += Whi * Xlo + Wlo * Xhi      (result = acc + 2^-11 accx)
template <int KS, int NB, int MBW>
__device__ __forceinline__ void mfma_chunk_h2(f32x4 (&acc)[MBW][NB], f32x4 (&accx)[MBW][NB], const unsigned char* sb,
                                              const unsigned char* wrow, int in_lo, int w_lo, const int (&prow)[MBW],
                                              int IW, int kg, int wswz) {
  constexpr int BN = NB * 16;
  constexpr int NSTEP = KS * KS;
  f16x8 wh[2][NB], wl[2][NB], xh[2][MBW], xl[2][MBW];
  auto load = [&](int tap, f16x8 (&a)[NB], f16x8 (&b)[NB], f16x8 (&c)[MBW], f16x8 (&d)[MBW]) {
    const int tap_rows = (tap / KS) * IW + tap % KS;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const unsigned char* q = wrow + (tap * BN + nb * 16) * ROWB + ((kg ^ wswz) << 4);
      a[nb] = *reinterpret_cast<const f16x8*>(q);
      b[nb] = *reinterpret_cast<const f16x8*>(q + w_lo);
    }
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
      const int row = prow[i] + tap_rows;
      const unsigned char* q = sb + row * ROWB + ((kg ^ swz<H2>(row)) << 4);
      c[i] = *reinterpret_cast<const f16x8*>(q);
      d[i] = *reinterpret_cast<const f16x8*>(q + in_lo);
    }
  };
  load(0, wh[0], wl[0], xh[0], xl[0]);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if (s + 1 < NSTEP) load(s + 1, wh[(s + 1) & 1], wl[(s + 1) & 1], xh[(s + 1) & 1], xl[(s + 1) & 1]);
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s & 1][nb], xh[s & 1][i], acc[i][nb], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) accx[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s & 1][nb], xl[s & 1][i], accx[i][nb], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) accx[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s & 1][nb], xh[s & 1][i], accx[i][nb], 0, 0, 0);
  }
}

// MBW = 16-pixel blocks per wave (ceil(M/64)); NCHW = epilogue writes the fp32 NCHW network output.
template <typename T, int KS, int STRIDE, int NB, int MBW, bool NCHW, int NW>
__device__ __forceinline__ void conv_mfma_body(const ConvParams& p, const int tile_id, const int cb) {
  constexpr int CK = Tr<T>::CK;
  constexpr int ESZ = Tr<T>::ESZ;
  constexpr int PL = Tr<T>::PL;
  constexpr int BN = NB * 16;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int WGROUPS = TAPS * BN / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  UDP_STAMP_MFMA(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kg = lane >> 4;

  int t = tile_id;   // wave-uniform tile decode
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n0 = (t / p.tiles_y) * p.G;
  const int y0 = ty * p.R;
  const int x0 = tx * p.TW;

  const int IH = p.IH, IW = p.IW;
  const int npix_in = p.G * IH * IW;
  const int in_groups = (npix_in + 15) >> 4;
  const int in_bytes = in_groups * 16 * ROWB;
  constexpr int W_BYTES = TAPS * BN * ROWB;           // one plane of the weight block
  const int stage_bytes = PL * (in_bytes + W_BYTES);  // [input planes][weight planes]
  const int nchunks = (p.Cin + CK - 1) / CK;
  const bool ragged = (p.Cin % CK) != 0;   // last chunk is partly past Cin: those 16-byte parts read zeros
  const int RT = p.R * p.TW;
  const int M = p.G * RT;
  const unsigned cinb = (unsigned)p.Cin * ESZ * PL;          // weight row (H2: hi plane, then lo plane)
  const unsigned inpb = (unsigned)p.in_pitch * ESZ * PL;     // input pixel (the conv may read a channel slice)
  const unsigned outpb = (unsigned)p.out_pitch * ESZ * PL, respb = (unsigned)p.res_pitch * ESZ * PL;
  const unsigned w_lo = (unsigned)p.Cin * ESZ, in_lo = (unsigned)p.in_pitch * ESZ;   // lo plane offsets (H2)
  const unsigned out_lo = (unsigned)p.out_pitch * ESZ, res_lo = (unsigned)p.res_pitch * ESZ;

  const unsigned out_pix = (unsigned)p.N * p.Hout * p.Wout;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.in), 0, (unsigned)p.N * (p.Hin >> p.in_stuff2) * (p.Win >> p.in_stuff2) * inpb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.wgt), 0, (unsigned)TAPS * p.CoutPad * cinb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, NCHW ? out_pix * p.Cout * 4 : out_pix * outpb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.res), 0, p.res ? out_pix * respb : 0, 0x00020000);

  // ---- per-lane DMA offsets of the input halo tile (chunk 0); masked rows -> kOobOff -> zeros
  const int srow = lane >> 2;  // row inside a 16-row group
  const int spart = lane & 3;  // stored 16-byte part position
  const int gy0 = y0 * STRIDE - PAD, gx0 = x0 * STRIDE - PAD;
  unsigned src_off[MAXG];
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    unsigned off = kOobOff;
    if ((wave + NW * i) * 16 < npix_in) {   // wave-uniform
      const int row = (wave + NW * i) * 16 + srow;
      const int tmp = fdiv20(row, p.mIW);
      const int ix = row - (int)__umul24(tmp, IW);
      const int g = fdiv20(tmp, p.mIH);
      const int iy = tmp - (int)__umul24(g, IH);
      const int n = n0 + g, gy = gy0 + iy, gx = gx0 + ix;
      bool ok = row < npix_in && n < p.N && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
      unsigned pix = __umul24(__umul24(n, p.Hin) + gy, p.Win) + gx;
      if (p.in_stuff2) {      // zero-stuffed view of a half-resolution tensor: odd rows / columns are zeros (wave-uniform flag)
        ok = ok && !((gy | gx) & 1);
        pix = __umul24(__umul24(n, p.Hin >> 1) + (gy >> 1), p.Win >> 1) + (gx >> 1);
      }
      off = ok ? pix * inpb + p.in_coff * ESZ + ((spart ^ swz<T>(row)) << 4) : kOobOff;
    }
    src_off[i] = off;
  }
  // weights: LDS row (tap, nb*16 + r) holds cout 4*NB*(r>>2) + 4*nb + (r&3) of this block
  auto stage = [&](int c, unsigned char* sb) {
    const unsigned coff = (unsigned)c * (CK * ESZ);
    const bool cut = ragged && c == nchunks - 1;           // wave-uniform
    const int parts_left = (p.Cin - c * CK) / (16 / ESZ);  // 16-byte parts of this chunk that exist
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int gidx = wave + NW * i;
      if (gidx < in_groups) {
        unsigned off = src_off[i] + coff;
        if (cut && (spart ^ swz<T>(gidx * 16 + srow)) >= parts_left) off = kOobOff;
        blds16(r_in, off, sb + gidx * (16 * ROWB));
        if constexpr (PL == 2) blds16(r_in, off + in_lo, sb + in_bytes + gidx * (16 * ROWB));
      }
    }
    for (int gidx = wave; gidx < WGROUPS; gidx += NW) {
      const int wr = gidx * 16 + srow;
      const int tap = wr / BN;   // BN is a power of two
      const int rho = wr & (BN - 1);
      const int co = (4 * NB) * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3);
      const int lp = spart ^ swz<T>(wr);
      unsigned e = __umul24(__umul24(tap, p.CoutPad) + cb * BN + co, cinb) + coff + (lp << 4);
      if (cut && lp >= parts_left) e = kOobOff;
      blds16(r_w, e, sb + PL * in_bytes + gidx * (16 * ROWB));
      if constexpr (PL == 2) blds16(r_w, e + w_lo, sb + PL * in_bytes + W_BYTES + gidx * (16 * ROWB));
    }
  };

  // ---- the lane's MBW output pixels
  int prow[MBW];        // LDS row of the pixel's (0,0) tap
  int opix[MBW];        // output pixel index, -1 = masked lane
  int ocrd[MBW];        // y | x << 10 | n << 20 (for the up-sampled addends / the NCHW form)
  const int cbase = cb * BN + 4 * NB * kg;
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    const int m0 = (wave + NW * i) * 16 + li;
    const int m = m0 < M ? m0 : M - 1;
    const int g = fdiv20(m, p.mRT);
    const int rem = m - (int)__umul24(g, RT);
    const int r = fdiv20(rem, p.mTW);
    const int x = rem - (int)__umul24(r, p.TW);
    prow[i] = (int)__umul24(__umul24(g, IH) + r * STRIDE, IW) + x * STRIDE;
    const int n = n0 + g, y = y0 + r, xo = x0 + x;
    const bool ok = m0 < M && n < p.N && y < p.Hout && xo < p.Wout && (NCHW || cbase < p.Cout);
    const unsigned pix = __umul24(__umul24(n, p.Hout) + y, p.Wout) + xo;
    opix[i] = ok ? (int)pix : -1;
    ocrd[i] = ok ? (y | (xo << 10) | (n << 20)) : -1;
  }
  const int wswz = swz<T>(li);  // weight rows: (tap*BN + nb*16) is a multiple of 16 -> swizzle depends on li only

  f32x4 acc[MBW][NB];
  f32x4 accx[PL == 2 ? MBW : 1][NB];   // H2: the 2^11-scaled cross terms
#pragma unroll
  for (int i = 0; i < (PL == 2 ? MBW : 1); ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) accx[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    f32x4 bias[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias[nb] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4 * nb);
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = bias[nb];
  }

  UDP_STAMP_MFMA(1);
  stage(0, smem);
  // The residual is added into the accumulators up front (acc = bias + res, then += conv): its loads fly
  // behind chunk 0's DMA and land under the same wait, instead of a second exposed round trip in the
  // epilogue.  Same sum, different fp32 order.
  if constexpr (!NCHW) {
    if (p.res) {
#pragma unroll
      for (int i = 0; i < MBW; ++i)
        add_vec_buf<T, NB>(acc[i], r_res, opix[i] >= 0 ? (unsigned)opix[i] * respb + (p.res_coff + cbase) * ESZ : kOobOff, res_lo);
    }
  }
  UDP_STAMP_MFMA(2);
  for (int c = 0; c < nchunks; ++c) {
    // chunk c has landed (explicit wait: the compiler is not obliged to track LDS-DMA) and every
    // wave is done with chunk c-1
    if (p.sbuf && c > 0) {   // one stage buffer (two workgroups per CU hide each other's DMA): refill it once all waves left chunk c-1
      __syncthreads();
      stage(c, smem);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c == 0) UDP_STAMP_MFMA(3);
    __syncthreads();
    if (c == 0) UDP_STAMP_MFMA(4);
    if (!p.sbuf && c + 1 < nchunks) stage(c + 1, smem + ((c + 1) & 1) * stage_bytes);
    const unsigned char* sb = p.sbuf ? smem : smem + (c & 1) * stage_bytes;
    if constexpr (PL == 2)
      mfma_chunk_h2<KS, NB, MBW>(acc, accx, sb, sb + PL * in_bytes + li * ROWB, in_bytes, W_BYTES, prow, IW, kg, wswz);
    else
      mfma_chunk<T, KS, NB, MBW>(acc, sb, sb + in_bytes + li * ROWB, prow, IW, kg, wswz);
  }
  if constexpr (PL == 2) {
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] += accx[i][nb] * kLoInv;
  }

  UDP_STAMP_MFMA(5);
  // BatchNorm statistics of the output (training, p.bn_ws): per-lane sums of x and x*x over its pixels, of the
  // values AS STORED (rounded to T) -- what the normalisation pass will read
  // (fp64 like the separate statistics pass: var = E[x^2] - mean^2 cancels for channels with |mean| >> std)
  double bsum[NB][4], bsq[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) bsum[nb][q] = bsq[nb][q] = 0.0;
  // ---- epilogue (acc = conv + bias): lane holds couts cbase .. cbase + 4*NB - 1 of pixel i
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    f32x4 v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) v[nb] = acc[i][nb];
    if constexpr (NCHW) {
      const int crd = ocrd[i];
      const int y = crd & 1023, xo = (crd >> 10) & 1023, n = crd >> 20;
      const unsigned hw = __umul24(p.Hout, p.Wout);
      const unsigned base = __umul24(__umul24(__umul24(n, p.Cout), p.Hout) + y, p.Wout) + xo;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cc = cbase + 4 * nb + q;
          float f = v[nb][q];
          if (p.relu) f = f > 0.f ? f : 0.f;
          const unsigned off = (crd >= 0 && cc < p.Cout) ? (base + cc * hw) * 4u : kOobOff;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f), r_out, off, 0, 0);
        }
    } else {
      const unsigned ooff = opix[i] >= 0 ? (unsigned)opix[i] * outpb + (p.out_coff + cbase) * ESZ : kOobOff;
      if (p.nup) {   // wave-uniform, rare (exchange-unit outputs only)
        const int crd = ocrd[i];
        const int y = crd & 1023, xo = (crd >> 10) & 1023, n = crd >> 20;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (u < p.nup) {
            const int s = p.up_shift[u];
            const long up_pix = ((long)(n * (p.Hout >> s) + (y >> s)) * (p.Wout >> s) + (xo >> s));
            if (crd >= 0)
              add_vec<T, NB>(v, reinterpret_cast<const unsigned char*>(p.up[u]) + (up_pix * p.Cout * PL + cbase) * ESZ, p.Cout * ESZ);
          }
        }
      }
      if (p.relu) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[nb][q] = v[nb][q] > 0.f ? v[nb][q] : 0.f;
      }
      store_vec_buf<T, NB>(r_out, ooff, out_lo, v);
      if constexpr (!std::is_same<T, H2>::value) {
        if (p.bn_ws && opix[i] >= 0) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double x = (double)(float)(T)v[nb][q];
              bsum[nb][q] += x;
              bsq[nb][q] += x * x;
            }
        }
      }
    }
  }
  if constexpr (!NCHW && !std::is_same<T, H2>::value) {
    if (p.bn_ws) {   // workgroup-uniform
      // lanes li = 0..15 of a kg group hold different pixels of the same 4*NB channels: butterfly over li, then
      // the NW waves through LDS (all stage buffers are free now), then ONE fp64 row per workgroup:
      // bn_ws[tile][0..C) = sum x, [C..2C) = sum x*x -- summed later in a fixed order (deterministic)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            bsum[nb][q] += __shfl_xor(bsum[nb][q], o, 64);
            bsq[nb][q] += __shfl_xor(bsq[nb][q], o, 64);
          }
      __syncthreads();                                   // every wave has left the MFMA loop: smem is free
      double* red = reinterpret_cast<double*>(smem);    // [wave][2][BN]
      if (li == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            red[(wave * 2 + 0) * BN + 4 * NB * kg + 4 * nb + q] = bsum[nb][q];
            red[(wave * 2 + 1) * BN + 4 * NB * kg + 4 * nb + q] = bsq[nb][q];
          }
      }
      __syncthreads();
      if (tid < 2 * BN) {
        const int which = tid / BN, ch = tid - which * BN;
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[(w * 2 + which) * BN + ch];
        if (cb * BN + ch < p.Cout) p.bn_ws[(size_t)tile_id * 2 * p.Cout + which * p.Cout + cb * BN + ch] = a;
      }
    }
  }
  UDP_STAMP_MFMA(6);
#ifdef UDP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UDP_STAMP_MFMA(7);
#endif
}

template <typename T, int KS, int STRIDE, int NB, int MBW, bool NCHW, int NW>
__global__ __launch_bounds__(NW * 64) void conv_mfma_kernel(const ConvParams p) {
  conv_mfma_body<T, KS, STRIDE, NB, MBW, NCHW, NW>(p, blockIdx.x, blockIdx.y);
}

// Horizontal fusion: up to 4 independent convs (the same-depth convs of different HRNet branches, which
// share the instantiation) in ONE launch.  Concurrent streams / graph branches do not overlap such
// launches on this GPU (measured), and each launch costs ~3.7 us of ramp-up and drain at batch 64; one
// grid covering all of them also evens out the per-launch workgroup-count quantisation.
template <typename T, int KS, int STRIDE, int NB, int MBW, int NW>
__global__ __launch_bounds__(NW * 64) void conv_mfma_multi(const ConvMulti m) {
  const unsigned b = blockIdx.x;
  const int j = (b >= m.start[1]) + (b >= m.start[2]) + (b >= m.start[3]);
  const unsigned r = b - m.start[j];
  const unsigned cb = r / m.tiles[j];
  conv_mfma_body<T, KS, STRIDE, NB, MBW, false, NW>(m.p[j], (int)(r - cb * m.tiles[j]), (int)cb);
}

// Persistent form of conv_mfma_kernel for layers whose Cin fits ONE 64-byte chunk (bf16 Cin = 32:
// the high-resolution branch, the most HBM-heavy and most numerous launches).  A workgroup keeps
// the whole weight block in LDS, walks tiles t = blockIdx.x, += gridDim.x and streams tile t+1's halo
// into the second input buffer while tile t feeds the MFMAs and its epilogue runs: the per-tile fixed
// cost (weight DMA, prologue index math, DMA latency) is paid once or hidden.  NHWC output only.
constexpr int MAXGP = 6;   // 16-row staging groups per wave (input tile <= 384 rows)

__device__ __forceinline__ int udiv16(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }   // x < 65536

template <typename T, int KS, int STRIDE, int NB, int MBW>
__global__ __launch_bounds__(256) void conv_mfma_persist1(const ConvParams p) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int BN = NB * 16;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int WGROUPS = TAPS * BN / 16;
  constexpr int WBYTES = TAPS * BN * ROWB;
  constexpr int NSTORE = MBW * (std::is_same<T, float>::value ? NB : NB / 2);   // store instructions per tile and wave
  static_assert(!std::is_same<T, float>::value || true, "");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kg = lane >> 4;
  const int cb = blockIdx.y;

  const int IH = p.IH, IW = p.IW;
  const int npix_in = p.G * IH * IW;
  const int in_groups = (npix_in + 15) >> 4;
  const int in_bytes = in_groups * 16 * ROWB;
  const int RT = p.R * p.TW;
  const int M = p.G * RT;
  const unsigned cinb = (unsigned)p.Cin * ESZ;
  unsigned char* const w_lds = smem;
  unsigned char* const in_lds = smem + WBYTES;

  const unsigned out_pix = (unsigned)p.N * p.Hout * p.Wout;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.in), 0, (unsigned)p.N * p.Hin * p.Win * cinb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.wgt), 0, (unsigned)TAPS * p.CoutPad * cinb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_pix * p.Cout * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.res), 0, p.res ? out_pix * p.Cout * ESZ : 0, 0x00020000);

  // ---- tile-invariant per-lane state
  const int srow = lane >> 2;
  const int spart = lane & 3;
  int s_rel[MAXGP];   // byte offset relative to the tile's input origin (swizzled part included)
  int s_crd[MAXGP];   // iy | ix << 10 | g << 20, -1 = row outside the tile
#pragma unroll
  for (int i = 0; i < MAXGP; ++i) {
    s_rel[i] = 0;
    s_crd[i] = -1;
    if ((wave + 4 * i) * 16 < npix_in) {
      const int row = (wave + 4 * i) * 16 + srow;
      const int tmp = fdiv20(row, p.mIW);
      const int ix = row - (int)__umul24(tmp, IW);
      const int g = fdiv20(tmp, p.mIH);
      const int iy = tmp - (int)__umul24(g, IH);
      s_rel[i] = (int)((__umul24(__umul24(g, p.Hin) + iy, p.Win) + ix) * cinb) + ((spart ^ swz<T>(row)) << 4);
      s_crd[i] = row < npix_in ? (iy | (ix << 10) | (g << 20)) : -1;
    }
  }
  const int cbase = cb * BN + 4 * NB * kg;
  int prow[MBW];    // LDS row of the pixel's (0,0) tap
  int o_rel[MBW];   // byte offset of the lane's channel vector relative to the tile's output origin
  int o_crd[MBW];   // r | x << 10 | g << 20, -1 = no pixel
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    const int m0 = (wave + 4 * i) * 16 + li;
    const int m = m0 < M ? m0 : M - 1;
    const int g = fdiv20(m, p.mRT);
    const int rem = m - (int)__umul24(g, RT);
    const int r = fdiv20(rem, p.mTW);
    const int x = rem - (int)__umul24(r, p.TW);
    prow[i] = (int)__umul24(__umul24(g, IH) + r * STRIDE, IW) + x * STRIDE;
    o_rel[i] = (int)(((__umul24(__umul24(g, p.Hout) + r, p.Wout) + x) * p.Cout + cbase) * ESZ);
    o_crd[i] = (m0 < M && cbase < p.Cout) ? (r | (x << 10) | (g << 20)) : -1;
  }
  const int wswz = swz<T>(li);
  f32x4 bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias[nb] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4 * nb);

  auto tile_origin = [&](int t, int& n0, int& y0, int& x0) {
    const int q1 = udiv16(t, p.mTX);
    const int tx = t - q1 * p.tiles_x;
    const int q2 = udiv16(q1, p.mTY);
    n0 = q2 * p.G;
    y0 = (q1 - q2 * p.tiles_y) * p.R;
    x0 = tx * p.TW;
  };
  auto stage_in = [&](int t, unsigned char* sb) {
    int n0, y0, x0;
    tile_origin(t, n0, y0, x0);
    const int gy0 = y0 * STRIDE - PAD, gx0 = x0 * STRIDE - PAD;
    const int org = ((n0 * p.Hin + gy0) * p.Win + gx0) * (int)cinb;
#pragma unroll
    for (int i = 0; i < MAXGP; ++i) {
      const int gidx = wave + 4 * i;
      if (gidx < in_groups) {
        const int crd = s_crd[i];
        const int gy = gy0 + (crd & 1023), gx = gx0 + ((crd >> 10) & 1023), n = n0 + ((crd >> 20) & 1023);
        const bool ok = crd >= 0 && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win && n < p.N;
        blds16(r_in, ok ? (unsigned)(org + s_rel[i]) : kOobOff, sb + gidx * (16 * ROWB));
      }
    }
  };

  const int ntiles = p.ntiles;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // weights once: LDS row (tap, nb*16 + r) holds cout 4*NB*(r>>2) + 4*nb + (r&3) of this block
  for (int gidx = wave; gidx < WGROUPS; gidx += 4) {
    const int wr = gidx * 16 + srow;
    const int tap = wr / BN;
    const int rho = wr & (BN - 1);
    const int co = (4 * NB) * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3);
    const unsigned e = __umul24(__umul24(tap, p.CoutPad) + cb * BN + co, cinb);
    blds16(r_w, e + ((spart ^ swz<T>(wr)) << 4), w_lds + gidx * (16 * ROWB));
  }
  stage_in(t, in_lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    // every wave has waited for its own DMA of this tile; after the barrier all of it is visible and
    // nobody still reads the other buffer
    __builtin_amdgcn_s_barrier();
    if (t + (int)gridDim.x < ntiles) stage_in(t + gridDim.x, in_lds + ((it + 1) & 1) * in_bytes);
    const unsigned char* sb = in_lds + (it & 1) * in_bytes;

    int n0, y0, x0;
    tile_origin(t, n0, y0, x0);
    const int o_org = ((n0 * p.Hout + y0) * p.Wout + x0) * p.Cout * ESZ;
    // residual of this tile: loads issued here, consumed in the epilogue (they fly under the MFMAs)
    ResRegs<T, NB> rres[MBW];
    if (p.res) {
#pragma unroll
      for (int i = 0; i < MBW; ++i) {
        const int crd = o_crd[i];
        const int y = y0 + (crd & 1023), xo = x0 + ((crd >> 10) & 1023), n = n0 + ((crd >> 20) & 1023);
        const bool ok = crd >= 0 && y < p.Hout && xo < p.Wout && n < p.N;
        load_res_buf<T, NB>(rres[i], r_res, ok ? (unsigned)(o_org + o_rel[i]) : kOobOff);
      }
    }

    f32x4 acc[MBW][NB];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = bias[nb];
    mfma_chunk<T, KS, NB, MBW>(acc, sb, w_lds + li * ROWB, prow, IW, kg, wswz);

    // ---- epilogue of tile t
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
      const int crd = o_crd[i];
      const int y = y0 + (crd & 1023), xo = x0 + ((crd >> 10) & 1023), n = n0 + ((crd >> 20) & 1023);
      const bool ok = crd >= 0 && y < p.Hout && xo < p.Wout && n < p.N;
      const unsigned voff = ok ? (unsigned)(o_org + o_rel[i]) : kOobOff;
      f32x4 v[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) v[nb] = acc[i][nb];
      if (p.res) add_res_regs<T, NB>(v, rres[i]);
      if (p.nup) {   // wave-uniform, rare
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (u < p.nup) {
            const int s = p.up_shift[u];
            const long up_pix = ((long)(n * (p.Hout >> s) + (y >> s)) * (p.Wout >> s) + (xo >> s));
            if (ok) add_vec<T, NB>(v, reinterpret_cast<const unsigned char*>(p.up[u]) + (up_pix * p.Cout + cbase) * ESZ, 0);
          }
        }
      }
      if (p.relu) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[nb][q] = v[nb][q] > 0.f ? v[nb][q] : 0.f;
      }
      store_vec_buf<T, NB>(r_out, voff, 0, v);
    }
    // the next tile's DMA was issued before this tile's NSTORE stores: it has landed once at most
    // NSTORE memory operations are still outstanding (they complete in issue order)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
  }
}

// ---------------------------------------------------------------------------------------------
// Fused BasicBlock for the 32-channel high-resolution branch (bf16):
//     out = relu( conv2( relu(conv1(x) + b1) ) + b2 + x )           pose_hrnet.py:43-59 with BN folded
// The two 3x3 convs of this branch are the HBM-bound launches of the network (three 25 MB tensors
// read + written per conv at batch 128 vs 7.8 GFLOP): fusing them keeps the intermediate map and the
// residual (= the block input, already staged) in LDS -- 2 tensors of HBM traffic per block instead of 5.
// Persistent workgroups (one per CU) walk tiles of R = 8 output rows x the full width: the input halo
// tile (R+4 rows) is LDS-DMA'd one tile ahead; conv1 is evaluated on R+2 rows (25 % halo recompute)
// and written (bias, ReLU, zero outside the image = conv2's zero padding) as bf16 into an LDS image with
// the same swizzled row layout the DMA produces, so conv2 runs the same mfma_chunk on it.
// Both weight blocks (2 x 18 KB) stay resident.
// ---------------------------------------------------------------------------------------------
constexpr int kBlkR = 8;
constexpr int kBlkMaxG = 10;      // 16-row staging groups per wave: input tile <= 640 rows

template <int MBW1, int MBW2, int NW>
__global__ __launch_bounds__(NW * 64) void basic_block_c32_kernel(const ConvParams p) {
  using T = __bf16;
  constexpr int ESZ = 2, NB = 2, BN = 32, TAPS = 9, C = 32;
  constexpr int WBYTES = TAPS * BN * ROWB;               // 18432
  constexpr int WGROUPS = TAPS * BN / 16;
  constexpr int NSTORE = MBW2 * (NB / 2);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kg = lane >> 4;

  const int H = p.Hin, W = p.Win;
  const int IW = p.IW, IH = p.IH;                        // W + 2, R + 4
  const int MH = kBlkR + 2;
  const int npix_in = IH * IW, in_groups = (npix_in + 15) >> 4, in_bytes = in_groups * 16 * ROWB;
  const int npix_mid = MH * IW, mid_bytes = ((npix_mid + 15) >> 4) * 16 * ROWB;
  const int M1 = MH * W, M2 = kBlkR * W;
  const unsigned cinb = C * ESZ;
  unsigned char* const w1_lds = smem;
  unsigned char* const w2_lds = smem + WBYTES;
  unsigned char* const mid_lds = smem + 2 * WBYTES;
  unsigned char* const in_lds = mid_lds + mid_bytes;     // two buffers

  const unsigned npix = (unsigned)p.N * H * W;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, npix * cinb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, TAPS * BN * cinb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt2), 0, TAPS * BN * cinb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, npix * cinb, 0x00020000);

  // ---- tile-invariant per-lane state: staging
  const int srow = lane >> 2, spart = lane & 3;
  constexpr int SG = (kBlkMaxG * 4 + NW - 1) / NW;   // staging groups per wave
  int s_rel[SG], s_crd[SG];                  // byte offset rel. to the tile origin; iy | ix << 10
#pragma unroll
  for (int i = 0; i < SG; ++i) {
    s_rel[i] = 0;
    s_crd[i] = -1;
    if ((wave + NW * i) * 16 < npix_in) {
      const int row = (wave + NW * i) * 16 + srow;
      const int iy = fdiv20(row, p.mIW);
      const int ix = row - (int)__umul24(iy, IW);
      s_rel[i] = (int)((__umul24(iy, W) + ix) * cinb) + ((spart ^ swz<T>(row)) << 4);
      s_crd[i] = row < npix_in ? (iy | (ix << 10)) : -1;
    }
  }
  // conv1: the lane's MBW1 mid pixels; conv2: its MBW2 output pixels
  int prow1[MBW1], mrow1[MBW1];      // sIn row of the (0,0) tap; sMid row to write | mid row index << 16, -1 = none
#pragma unroll
  for (int i = 0; i < MBW1; ++i) {
    const int m0 = (wave + NW * i) * 16 + li;
    const int m = m0 < M1 ? m0 : M1 - 1;
    const int mr = fdiv20(m, p.mTW);
    const int mx = m - (int)__umul24(mr, W);
    prow1[i] = (int)__umul24(mr, IW) + mx;
    mrow1[i] = m0 < M1 ? (((int)__umul24(mr, IW) + mx + 1) | (mr << 16)) : -1;
  }
  int prow2[MBW2], o_rel[MBW2], o_r[MBW2];  // sMid row of the (0,0) tap; output byte offset rel. to the tile; row r or -1
#pragma unroll
  for (int i = 0; i < MBW2; ++i) {
    const int m0 = (wave + NW * i) * 16 + li;
    const int m = m0 < M2 ? m0 : M2 - 1;
    const int r = fdiv20(m, p.mTW);
    const int x = m - (int)__umul24(r, W);
    prow2[i] = (int)__umul24(r, IW) + x;
    o_rel[i] = (int)((__umul24(r, W) + x) * cinb) + 16 * kg;
    o_r[i] = m0 < M2 ? r : -1;
  }
  const int wswz = swz<T>(li);
  const int cbase = 4 * NB * kg;     // the lane's 8 consecutive output channels
  f32x4 bias1[NB], bias2[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    bias1[nb] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4 * nb);
    bias2[nb] = *reinterpret_cast<const f32x4*>(p.bias2 + cbase + 4 * nb);
  }

  auto stage_in = [&](int t, unsigned char* sb) {
    const int n = udiv16(t, p.mTY);
    const int y0 = (t - n * p.tiles_y) * kBlkR;
    const int gy0 = y0 - 2;
    const int org = ((n * H + gy0) * W - 1) * (int)cinb;
#pragma unroll
    for (int i = 0; i < SG; ++i) {
      const int gidx = wave + NW * i;
      if (gidx < in_groups) {
        const int crd = s_crd[i];
        const int gy = gy0 + (crd & 1023), gx = ((crd >> 10) & 1023) - 1;
        const bool ok = crd >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        blds16(r_in, ok ? (unsigned)(org + s_rel[i]) : kOobOff, sb + gidx * (16 * ROWB));
      }
    }
  };

  const int ntiles = p.ntiles;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // zero the intermediate image once: its halo columns are never written again (conv2's zero padding)
  for (int i = tid * 16; i < mid_bytes; i += NW * 64 * 16) *reinterpret_cast<u32x4*>(mid_lds + i) = u32x4{0u, 0u, 0u, 0u};
  // both weight blocks once: LDS row (tap, nb*16 + r) holds cout 4*NB*(r>>2) + 4*nb + (r&3)
  for (int gidx = wave; gidx < WGROUPS; gidx += NW) {
    const int wr = gidx * 16 + srow;
    const int tap = wr / BN;
    const int rho = wr & (BN - 1);
    const int co = (4 * NB) * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3);
    const unsigned e = (unsigned)(tap * BN + co) * cinb + ((spart ^ swz<T>(wr)) << 4);
    blds16(r_w1, e, w1_lds + gidx * (16 * ROWB));
    blds16(r_w2, e, w2_lds + gidx * (16 * ROWB));
  }
  stage_in(t, in_lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    // this tile's input has landed in every wave's view after the barrier; nobody reads the other input
    // buffer or the intermediate image of the previous tile any more
    __builtin_amdgcn_s_barrier();
    if (t + (int)gridDim.x < ntiles) stage_in(t + gridDim.x, in_lds + ((it + 1) & 1) * in_bytes);
    const unsigned char* sb = in_lds + (it & 1) * in_bytes;
    const int n = udiv16(t, p.mTY);
    const int y0 = (t - n * p.tiles_y) * kBlkR;

    {   // ---- conv1 on the R+2 intermediate rows -> LDS (bias, ReLU, zero outside the image)
      f32x4 acc[MBW1][NB];
#pragma unroll
      for (int i = 0; i < MBW1; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = bias1[nb];
      mfma_chunk<T, 3, NB, MBW1>(acc, sb, w1_lds + li * ROWB, prow1, IW, kg, wswz);
#pragma unroll
      for (int i = 0; i < MBW1; ++i) {
        const int mr = mrow1[i] >> 16, row = mrow1[i] & 0xFFFF;
        if (mrow1[i] >= 0) {
          const bool inside = (unsigned)(y0 - 1 + mr) < (unsigned)H;
          bf16x8 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a = acc[i][0][q], b = acc[i][1][q];
            o[q] = (__bf16)(inside && a > 0.f ? a : 0.f);
            o[4 + q] = (__bf16)(inside && b > 0.f ? b : 0.f);
          }
          *reinterpret_cast<bf16x8*>(mid_lds + row * ROWB + ((kg ^ swz<T>(row)) << 4)) = o;
        }
      }
    }
    __syncthreads();
    {   // ---- conv2 from the LDS image + residual from the staged input -> global
      f32x4 acc[MBW2][NB];
#pragma unroll
      for (int i = 0; i < MBW2; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = bias2[nb];
      mfma_chunk<T, 3, NB, MBW2>(acc, mid_lds, w2_lds + li * ROWB, prow2, IW, kg, wswz);
      const int o_org = ((n * H + y0) * W) * (int)cinb;
#pragma unroll
      for (int i = 0; i < MBW2; ++i) {
        const int r = o_r[i];
        const bool ok = r >= 0 && y0 + r < H;
        const int rrow = prow2[i] + 2 * IW + 1;                 // the block input at the output pixel
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(sb + rrow * ROWB + ((kg ^ swz<T>(rrow)) << 4));
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = acc[i][0][q] + (float)x[q], b = acc[i][1][q] + (float)x[4 + q];
          o[q] = (__bf16)(a > 0.f ? a : 0.f);
          o[4 + q] = (__bf16)(b > 0.f ? b : 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_out, ok ? (unsigned)(o_org + o_rel[i]) : kOobOff, 0, 0);
      }
    }
    // the next tile's DMA was issued before this tile's NSTORE stores
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
  }
}

// Stem conv1: 3x3 stride-2 conv on the NCHW fp32 network input (Cin = 3), direct
// VALU form (27 taps), + folded BN + ReLU, NHWC output.  pose_hrnet.py:290-292,
// :437-439.  Images n >= flip_from read image n - flip_from mirrored along W
// (flip-test second pass, function.py:154-156).
template <typename T>
__global__ __launch_bounds__(256, 4) void stem_conv_kernel(const ConvParams p) {
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
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {   // rolled on purpose: a full unroll needs > 256 VGPRs
    const int ky = tap / 3, kx = tap - ky * 3;
    const int gy = yo * 2 - 1 + ky;
    const int gx = xo * 2 - 1 + kx;
    const bool ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
    const int sx = mirror ? p.Win - 1 - gx : gx;
    const float* wr = &w_s[tap * 3 * 64 + cg * 16];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float v = ok ? in[((size_t)ci * p.Hin + gy) * p.Win + sx] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; q += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + ci * 64 + q);
        acc[q + 0] = fmaf(v, w4[0], acc[q + 0]);
        acc[q + 1] = fmaf(v, w4[1], acc[q + 1]);
        acc[q + 2] = fmaf(v, w4[2], acc[q + 2]);
        acc[q + 3] = fmaf(v, w4[3], acc[q + 3]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 16; q += 4) {
    f32x4 v = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
    if (p.relu) {
#pragma unroll
      for (int z = 0; z < 4; ++z) v[z] = v[z] > 0.f ? v[z] : 0.f;
    }
    st4<T>(p.out, (size_t)pix, 64, cg * 16 + q, v);
  }
}

// bf16 stem on the matrix pipe: conv1 as a GEMM with K = 27 taps*channels (padded to 32).  No LDS: a lane
// gathers the 8 K-values of its pixel straight from the NCHW fp32 input (neighbouring pixels share cache
// lines), splits each into bf16 hi + lo (the input keeps ~16 bits through two MFMAs), the 64 x 32 weight
// block lives in registers.  HBM-bound (75 MB in, 201 MB out at batch 128) instead of VALU-bound.
__global__ __launch_bounds__(256) void stem_mfma_kernel(const ConvParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const float* wg = reinterpret_cast<const float*>(p.wgt);
  bf16x8 wf[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int cout = 16 * (li >> 2) + 4 * nb + (li & 3);       // row li of tile nb (the lane-owns-16-couts permutation)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * kg + j;
      wf[nb][j] = (__bf16)(k < 27 ? wg[k * 64 + cout] : 0.f);
    }
  }
  const int cbase = 16 * kg;
  f32x4 bias[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bias[nb] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4 * nb);
  int koff[8];                                                // dy+1 | (dx+1) << 2 | ch << 4, -1 = padding k
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * kg + j, tap = k / 3;
    koff[j] = k < 27 ? ((tap / 3) | ((tap % 3) << 2) | ((k % 3) << 4)) : -1;
  }
  const long total = (long)p.N * p.Hout * p.Wout;
  const long ntile = (total + 15) / 16;
  const size_t plane = (size_t)p.Hin * p.Win;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntile; tile += (long)gridDim.x * 4) {
    const long pix = tile * 16 + li;
    const bool okp = pix < total;
    const long pp = okp ? pix : total - 1;
    const int xo = pp % p.Wout;
    const long t2 = pp / p.Wout;
    const int yo = t2 % p.Hout;
    const int n = t2 / p.Hout;
    const bool mirror = n >= p.flip_from;
    const float* in = reinterpret_cast<const float*>(p.in) + (size_t)(mirror ? n - p.flip_from : n) * 3 * plane;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ko = koff[j];
      const int gy = yo * 2 - 1 + (ko & 3), gx = xo * 2 - 1 + ((ko >> 2) & 3);
      const bool ok = ko >= 0 && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
      const int sx = mirror ? p.Win - 1 - gx : gx;
      const float v = ok ? in[(size_t)(ko >> 4) * plane + (size_t)gy * p.Win + sx] : 0.f;
      hi[j] = (__bf16)v;
      lo[j] = (__bf16)(v - (float)hi[j]);
    }
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], hi, bias[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], lo, acc[nb], 0, 0, 0);
    }
    if (okp) {
      __bf16* o = reinterpret_cast<__bf16*>(p.out) + (size_t)pix * 64 + cbase;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 ov;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = acc[2 * h][q], b = acc[2 * h + 1][q];
          if (p.relu) {
            a = a > 0.f ? a : 0.f;
            b = b > 0.f ? b : 0.f;
          }
          ov[q] = (__bf16)a;
          ov[4 + q] = (__bf16)b;
        }
        *reinterpret_cast<bf16x8*>(o + 8 * h) = ov;
      }
    }
  }
}

// Stem convs on the matrix pipe, general form: KS x KS stride-2 conv on the NCHW fp32 network input as a GEMM with
// K = KS*KS*3 (27 -> one 32-deep k-step; 147 -> five: the RSN 7x7 stem, network.py:125-137), for bf16 and for
// split-fp16 output.  The weight fragments (fragment-major: [k-step][cout block][plane][lane][8]) are built once
// per workgroup in LDS from the fp32 [ky][kx][ci][64] weights; a lane gathers the 8 K values of its pixel straight
// from the input (neighbouring pixels share cache lines) and splits each into hi + lo (bf16: the input keeps ~16
// bits through two MFMAs on bf16 weights; H2: three fp16 MFMAs on hi/lo weights, as everywhere in that mode).
template <typename T, int KS>
__global__ __launch_bounds__(256) void stem_mfma_k(const ConvParams p) {
  constexpr bool SPLIT = std::is_same<T, H2>::value;
  constexpr int K = KS * KS * 3, NS = (K + 31) / 32, PAD = KS / 2, WPL = SPLIT ? 2 : 1;
  using E = typename std::conditional<SPLIT, _Float16, __bf16>::type;
  using Frag = typename std::conditional<SPLIT, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* wl = reinterpret_cast<E*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const float* wg = reinterpret_cast<const float*>(p.wgt);
  for (int e = tid; e < NS * 4 * WPL * 512; e += 256) {
    const int j = e & 7, ln = (e >> 3) & 63, pl = (e >> 9) % WPL, nb = ((e >> 9) / WPL) & 3, st = (e >> 9) / (WPL * 4);
    const int k = 32 * st + 8 * (ln >> 4) + j;
    const int cout = 16 * ((ln & 15) >> 2) + 4 * nb + (ln & 3);       // row ln&15 of tile nb (lane-owns-16-couts permutation)
    const float w = k < K ? wg[k * 64 + cout] : 0.f;
    if constexpr (SPLIT) {
      const _Float16 hi = (_Float16)w;
      wl[e] = pl == 0 ? hi : (_Float16)((w - (float)hi) * kLoScale);
    } else {
      wl[e] = (__bf16)w;
    }
  }
  __syncthreads();
  const int cbase = 16 * kg;
  f32x4 bias[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bias[nb] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4 * nb);
  unsigned koff[NS][4];            // per k-step the lane's 8 K values as 16-bit (ky | kx << 4 | ch << 8), 0xFFFF = padding
#pragma unroll
  for (int st = 0; st < NS; ++st)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      unsigned v = 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = 32 * st + 8 * kg + 2 * h + q, tap = k / 3;
        const unsigned c = k < K ? (unsigned)((tap / KS) | ((tap % KS) << 4) | ((k % 3) << 8)) : 0xFFFFu;
        v |= c << (16 * q);
      }
      koff[st][h] = v;
    }
  const long total = (long)p.N * p.Hout * p.Wout;
  const long ntile = (total + 15) / 16;
  const size_t plane = (size_t)p.Hin * p.Win;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntile; tile += (long)gridDim.x * 4) {
    const long pix = tile * 16 + li;
    const bool okp = pix < total;
    const long pp = okp ? pix : total - 1;
    const int xo = pp % p.Wout;
    const long t2 = pp / p.Wout;
    const int yo = t2 % p.Hout;
    const int n = t2 / p.Hout;
    const bool mirror = n >= p.flip_from;
    const float* in = reinterpret_cast<const float*>(p.in) + (size_t)(mirror ? n - p.flip_from : n) * 3 * plane;
    f32x4 acc[4], accx[SPLIT ? 4 : 1];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = bias[nb];
#pragma unroll
    for (int nb = 0; nb < (SPLIT ? 4 : 1); ++nb) accx[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      Frag hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned ko = (koff[st][j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        const int gy = yo * 2 - PAD + (int)(ko & 15u), gx = xo * 2 - PAD + (int)((ko >> 4) & 15u);
        const bool ok = ko != 0xFFFFu && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
        const int sx = mirror ? p.Win - 1 - gx : gx;
        const float v = ok ? in[(size_t)(ko >> 8) * plane + (size_t)gy * p.Win + sx] : 0.f;
        hi[j] = (E)v;
        lo[j] = (E)((v - (float)hi[j]) * (SPLIT ? kLoScale : 1.f));
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const Frag wh = *reinterpret_cast<const Frag*>(wl + ((st * 4 + nb) * WPL) * 512 + lane * 8);
        if constexpr (SPLIT) {
          const Frag wlo = *reinterpret_cast<const Frag*>(wl + ((st * 4 + nb) * WPL + 1) * 512 + lane * 8);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hi, acc[nb], 0, 0, 0);
          accx[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, lo, accx[nb], 0, 0, 0);
          accx[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, hi, accx[nb], 0, 0, 0);
        } else {
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hi, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, lo, acc[nb], 0, 0, 0);
        }
      }
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] += accx[nb] * kLoInv;
    }
    if (p.relu) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[nb][q] = acc[nb][q] > 0.f ? acc[nb][q] : 0.f;
    }
    if (okp) {
      if constexpr (SPLIT) {
        unsigned char* o = reinterpret_cast<unsigned char*>(p.out) + (size_t)pix * 256 + cbase * 2;   // 64 hi + 64 lo fp16 per pixel
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f16x8 oh, ol;
          h2_split8(acc[2 * h], acc[2 * h + 1], oh, ol);
          *reinterpret_cast<f16x8*>(o + 16 * h) = oh;
          *reinterpret_cast<f16x8*>(o + 128 + 16 * h) = ol;
        }
      } else {
        __bf16* o = reinterpret_cast<__bf16*>(p.out) + (size_t)pix * 64 + cbase;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 ov;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ov[q] = (__bf16)acc[2 * h][q];
            ov[4 + q] = (__bf16)acc[2 * h + 1][q];
          }
          *reinterpret_cast<bf16x8*>(o + 8 * h) = ov;
        }
      }
    }
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
    f32x4 v = ld4<T>(p.in, (size_t)pix, p.in_pitch, p.in_coff + c);
    if (p.res) v += ld4<T>(p.res, (size_t)pix, p.res_pitch, p.res_coff + c);
    for (int u = 0; u < p.nup; ++u) {
      const int s = p.up_shift[u];
      const size_t up_pix = ((size_t)(n * (p.Hout >> s) + (y >> s)) * (p.Wout >> s) + (xo >> s));
      v += ld4<T>(p.up[u], up_pix, p.Cout, c);
    }
    if (p.relu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
    }
    st4<T>(p.out, (size_t)pix, p.out_pitch, p.out_coff + c, v);
  }
}

// RSN stem (RSN/exps/RSN18.coco/network.py:125-137): 7x7 stride-2 pad-3 conv on the NCHW fp32 input
// (Cin = 3, 147 taps, direct VALU form) + folded BN + ReLU -> NHWC.  Same thread layout as
// stem_conv_kernel: 64 pixels x 4 groups of 16 output channels per workgroup; weights [ky][kx][ci][64].
template <typename T>
__global__ __launch_bounds__(256, 4) void stem7_conv_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float w7_s[];   // 147*64 weights + 64 bias
  const int tid = threadIdx.x;
  const float* wg = reinterpret_cast<const float*>(p.wgt);
  for (int i = tid; i < 147 * 64; i += 256) w7_s[i] = wg[i];
  if (tid < 64) w7_s[147 * 64 + tid] = p.bias[tid];
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
  for (int q = 0; q < 16; ++q) acc[q] = w7_s[147 * 64 + cg * 16 + q];
#pragma unroll 1
  for (int tap = 0; tap < 49; ++tap) {
    const int ky = tap / 7, kx = tap - ky * 7;
    const int gy = yo * 2 - 3 + ky;
    const int gx = xo * 2 - 3 + kx;
    const bool ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
    const int sx = mirror ? p.Win - 1 - gx : gx;
    const float* wr = &w7_s[tap * 3 * 64 + cg * 16];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float v = ok ? in[((size_t)ci * p.Hin + gy) * p.Win + sx] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; q += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + ci * 64 + q);
        acc[q + 0] = fmaf(v, w4[0], acc[q + 0]);
        acc[q + 1] = fmaf(v, w4[1], acc[q + 1]);
        acc[q + 2] = fmaf(v, w4[2], acc[q + 2]);
        acc[q + 3] = fmaf(v, w4[3], acc[q + 3]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 16; q += 4) {
    f32x4 v = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
    if (p.relu) {
#pragma unroll
      for (int z = 0; z < 4; ++z) v[z] = v[z] > 0.f ? v[z] : 0.f;
    }
    st4<T>(p.out, (size_t)pix, 64, cg * 16 + q, v);
  }
}

// MaxPool2d(3, stride 2, pad 1) on NHWC (network.py:131,136); padding never wins (-inf).
template <typename T>
__global__ __launch_bounds__(256) void maxpool3_kernel(const ConvParams p) {
  const int C4 = p.Cout >> 2;
  const long total = (long)p.N * p.Hout * p.Wout * C4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (idx % C4) * 4;
    const long pix = idx / C4;
    const int xo = pix % p.Wout;
    const long t2 = pix / p.Wout;
    const int y = t2 % p.Hout;
    const int n = t2 / p.Hout;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < 3; ++ky) {
      const int gy = y * 2 - 1 + ky;
      if (gy < 0 || gy >= p.Hin) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int gx = xo * 2 - 1 + kx;
        if (gx < 0 || gx >= p.Win) continue;
        const f32x4 v = ld4<T>(p.in, (size_t)(n * p.Hin + gy) * p.Win + gx, p.in_pitch, p.in_coff + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = v[q] > m[q] ? v[q] : m[q];
      }
    }
    st4<T>(p.out, (size_t)pix, p.out_pitch, p.out_coff + c, m);
  }
}

// F.interpolate(mode='bilinear', align_corners=True) on NHWC (network.py:246-255): src = dst*(in-1)/(out-1),
// fp32 weights like ATen's upsample_bilinear2d.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_ac_kernel(const ConvParams p) {
  const int C4 = p.Cout >> 2;
  const long total = (long)p.N * p.Hout * p.Wout * C4;
  const float sy = p.Hout > 1 ? (float)(p.Hin - 1) / (float)(p.Hout - 1) : 0.f;
  const float sx = p.Wout > 1 ? (float)(p.Win - 1) / (float)(p.Wout - 1) : 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (idx % C4) * 4;
    const long pix = idx / C4;
    const int xo = pix % p.Wout;
    const long t2 = pix / p.Wout;
    const int y = t2 % p.Hout;
    const int n = t2 / p.Hout;
    const float fy = sy * (float)y, fx = sx * (float)xo;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.Hin - 1 ? 1 : 0), x1 = x0 + (x0 < p.Win - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const size_t ib = (size_t)n * p.Hin * p.Win;
    const f32x4 v00 = ld4<T>(p.in, ib + (size_t)y0 * p.Win + x0, p.in_pitch, p.in_coff + c);
    const f32x4 v01 = ld4<T>(p.in, ib + (size_t)y0 * p.Win + x1, p.in_pitch, p.in_coff + c);
    const f32x4 v10 = ld4<T>(p.in, ib + (size_t)y1 * p.Win + x0, p.in_pitch, p.in_coff + c);
    const f32x4 v11 = ld4<T>(p.in, ib + (size_t)y1 * p.Win + x1, p.in_pitch, p.in_coff + c);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = hy * (hx * v00[q] + lx * v01[q]) + ly * (hx * v10[q] + lx * v11[q]);
    st4<T>(p.out, (size_t)pix, p.out_pitch, p.out_coff + c, v);
  }
}

// ---------------------------------------------------------------------------
// Host side: tile selection + dispatch
// ---------------------------------------------------------------------------
// Picks the (G, R, TW) tile and NB for one conv; returns the dynamic LDS bytes.
size_t conv_choose_tile(ConvParams& p, int ks, int stride, int dtype, int* nb_out, bool grouped = false) {
  // tuning knobs (environment overrides are a measurement aid for tools/profile_layers.py)
  auto knob = [](const char* name, long dflt) {
    const char* v = getenv(name);
    return v ? atol(v) : dflt;
  };
  const int kMaxM = (int)knob("UDP_POSE_MAXM", 256);
  const int ck = dtype == UDP_F32 ? 16 : 32;
  const int planes = dtype == UDP_F16X2 ? 2 : 1;
  // split-fp16 tiles are twice the bytes: K-deep convs keep ONE stage buffer so that two workgroups fit a CU
  // (UDP_POSE_H2_DBUF=1: double-buffered, one workgroup of up to 156 KB per CU)
  static const bool h2_dbuf = getenv("UDP_POSE_H2_DBUF") != nullptr;
  p.sbuf = planes == 2 && !h2_dbuf;
  const int nstage = ceil_div(p.Cin, ck) > 1 && !p.sbuf ? 2 : 1;
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
  int NB = (p.CoutPad % 64 == 0 && !grouped) ? 4 : 2;
  auto npix = [&](int g, int r) { return g * ((r - 1) * stride + ks) * ((TW - 1) * stride + ks); };
  auto lds = [&](int g, int r, int nb) {
    return (size_t)(((npix(g, r) + 15) / 16) * 16 + ks * ks * nb * 16) * ROWB * nstage * planes;
  };
  auto wgs = [&](int g, int r, int nb) {
    return (long)ceil_div(p.N, g) * ceil_div(p.Hout, r) * ceil_div(p.Wout, TW) * (p.CoutPad / (nb * 16));
  };
  // small problems: trade tile size for workgroups (256 CUs, aim for >= 2 per CU)
  const long kMinWgs = grouped ? 0 : knob("UDP_POSE_MINWGS", 512);   // a grouped launch fills the chip with its siblings
  if (wgs(G, R, NB) < kMinWgs && NB == 4) NB = 2;
  while (wgs(G, R, NB) < kMinWgs && G > 1) G = (G + 1) / 2;
  // 76 KB: two workgroups per CU.  Split-fp16 tiles are twice the bytes; the K-deep ones take the whole CU
  // (one workgroup of 156 KB) rather than shrink below the 9*BN weight rows they are staged with
  const size_t kLimit = (size_t)knob("UDP_POSE_LDS_KB", planes == 2 && nstage == 2 ? 156 : 76) * 1024;
  while ((lds(G, R, NB) > kLimit || npix(G, R) > MAXG * 64) && G > 1) --G;
  // stride 2: the halo tile is 4x the output pixels -- rather keep 4 cout blocks per staged tile and
  // shrink the tile (+0.4 % images/s; UDP_POSE_NO_S2_NB4 restores the old order)
  static const bool s2nb4 = getenv("UDP_POSE_NO_S2_NB4") == nullptr;
  if (s2nb4 && stride == 2 && NB == 4)
    while ((lds(G, R, NB) > kLimit || npix(G, R) > MAXG * 64) && R > 2) R = (R + 1) / 2;
  while ((lds(G, R, NB) > kLimit || npix(G, R) > MAXG * 64) && NB == 4) NB = 2;
  while ((lds(G, R, NB) > kLimit || npix(G, R) > MAXG * 64) && R > 1) R = (R + 1) / 2;
  p.G = G;
  p.R = R;
  p.TW = TW;
  p.IH = (R - 1) * stride + ks;
  p.IW = (TW - 1) * stride + ks;
  p.tiles_x = ceil_div(p.Wout, TW);
  p.tiles_y = ceil_div(p.Hout, R);
  auto magic = [](int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); };   // fdiv20
  p.mIW = magic(p.IW);
  p.mIH = magic(p.IH);
  p.mRT = magic(R * TW);
  p.mTW = magic(TW);
  p.ntiles = ceil_div(p.N, G) * p.tiles_y * p.tiles_x;
  auto magic32 = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); };
  p.mTX = magic32(p.tiles_x);
  p.mTY = magic32(p.tiles_y);
  *nb_out = NB;
  return lds(G, R, NB);
}

template <typename T, int KS, int STRIDE, int NB, int MBW, bool NCHW, int NW>
static int describe_one(const ConvParams& p, size_t lds, Launch* out) {
  static bool attr_set = false;
  const void* kern = reinterpret_cast<const void*>(&conv_mfma_kernel<T, KS, STRIDE, NB, MBW, NCHW, NW>);
  if (!attr_set) {
    UDP_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  out->fn = kern;
  out->grid = dim3(ceil_div(p.N, p.G) * p.tiles_y * p.tiles_x, p.CoutPad / (NB * 16));
  out->block = dim3(NW * 64);
  out->lds = (unsigned)lds;
  out->p = p;
  return UDP_OK;
}

template <typename T, int KS, int STRIDE, int NB, bool NCHW>
static int describe_mbw(const ConvParams& p, int mbw, size_t lds, Launch* out) {
  // mbw = 16-pixel blocks per wave with 4 waves.  Tiles of more than 128 pixels run with 8 waves (two per
  // SIMD, half the blocks each): the same LDS tile feeds twice the waves, which hides LDS / DMA latency
  static const bool w8 = getenv("UDP_POSE_CONV_8W") != nullptr;
  if (w8 && std::is_same<T, __bf16>::value && !NCHW && mbw >= 3)
    return mbw == 3 ? describe_one<T, KS, STRIDE, NB, 2, NCHW, 8>(p, lds, out) : describe_one<T, KS, STRIDE, NB, 2, NCHW, 8>(p, lds, out);
  switch (mbw) {
    case 1: return describe_one<T, KS, STRIDE, NB, 1, NCHW, 4>(p, lds, out);
    case 2: return describe_one<T, KS, STRIDE, NB, 2, NCHW, 4>(p, lds, out);
    case 3: return describe_one<T, KS, STRIDE, NB, 3, NCHW, 4>(p, lds, out);
    case 4: return describe_one<T, KS, STRIDE, NB, 4, NCHW, 4>(p, lds, out);
  }
  return fail(UDP_ERR_UNSUPPORTED, "conv tile of %d pixel blocks per wave has no kernel", mbw);
}

template <typename T>
static int describe_conv_t(const ConvParams& p, int ks, int stride, int nb, int mbw, size_t lds, Launch* out) {
#define UDP_CASE(K, S, B)                                                              \
  if (ks == K && stride == S && nb == B && !p.out_nchw_f32) return describe_mbw<T, K, S, B, false>(p, mbw, lds, out);
#define UDP_CASE_OUT(K, S, B)                                                          \
  if (ks == K && stride == S && nb == B && p.out_nchw_f32) return describe_mbw<T, K, S, B, true>(p, mbw, lds, out);
  UDP_CASE(3, 1, 2) UDP_CASE(3, 1, 4) UDP_CASE(3, 2, 2) UDP_CASE(3, 2, 4)
  UDP_CASE(1, 1, 2) UDP_CASE(1, 1, 4) UDP_CASE(1, 2, 2) UDP_CASE(1, 2, 4)
  UDP_CASE_OUT(1, 1, 2) UDP_CASE_OUT(1, 1, 4) UDP_CASE_OUT(3, 1, 2) UDP_CASE_OUT(3, 1, 4)
#undef UDP_CASE
#undef UDP_CASE_OUT
  return fail(UDP_ERR_UNSUPPORTED, "conv ks=%d stride=%d nb=%d nchw_out=%d has no kernel", ks, stride, nb,
              p.out_nchw_f32);
}

template <int KS, int STRIDE, int NB, int MBW>
static int describe_persist_one(const ConvParams& p, size_t lds, int grid_x, Launch* out) {
  static bool attr_set = false;
  const void* kern = reinterpret_cast<const void*>(&conv_mfma_persist1<__bf16, KS, STRIDE, NB, MBW>);
  if (!attr_set) {
    UDP_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  out->fn = kern;
  out->grid = dim3(grid_x, p.CoutPad / (NB * 16));
  out->block = dim3(256);
  out->lds = (unsigned)lds;
  out->p = p;
  return UDP_OK;
}

template <int KS, int STRIDE, int NB>
static int describe_persist_mbw(const ConvParams& p, int mbw, size_t lds, int grid_x, Launch* out) {
  switch (mbw) {
    case 1: return describe_persist_one<KS, STRIDE, NB, 1>(p, lds, grid_x, out);
    case 2: return describe_persist_one<KS, STRIDE, NB, 2>(p, lds, grid_x, out);
    case 3: return describe_persist_one<KS, STRIDE, NB, 3>(p, lds, grid_x, out);
    case 4: return describe_persist_one<KS, STRIDE, NB, 4>(p, lds, grid_x, out);
  }
  return 1;
}

// Persistent single-chunk form (bf16, Cin = 32): returns 1 when it does not apply.
static int describe_persist(const ConvParams& p, int ks, int stride, int nb, int mbw, Launch* out) {
  const int npix = p.G * p.IH * p.IW;
  if (p.out_nchw_f32 || p.Cin != 32 || npix > MAXGP * 64 || p.ntiles >= 65536 || p.bn_ws || p.in_stuff2) return 1;
  if (p.in_pitch != p.Cin || p.in_coff || p.out_pitch != p.Cout || p.out_coff || (p.res && (p.res_pitch != p.Cout || p.res_coff)))
    return 1;                                         // channel-slice views: generic kernel
  const size_t lds = (size_t)(ks * ks * nb * 16 + 2 * ((npix + 15) / 16) * 16) * ROWB;
  int per_cu = (int)((150 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) return 1;
  const int cblocks = p.CoutPad / (nb * 16);
  const int budget = ceil_div(256 * per_cu, cblocks);
  if (p.ntiles < 2 * budget) return 1;               // one tile per workgroup: nothing to overlap
  const int rounds = ceil_div(p.ntiles, budget);
  const int grid_x = ceil_div(p.ntiles, rounds);
#define UDP_CASE(K, S, B) \
  if (ks == K && stride == S && nb == B) return describe_persist_mbw<K, S, B>(p, mbw, lds, grid_x, out);
  UDP_CASE(3, 1, 2) UDP_CASE(3, 1, 4) UDP_CASE(3, 2, 2) UDP_CASE(3, 2, 4) UDP_CASE(1, 1, 2) UDP_CASE(1, 1, 4)
#undef UDP_CASE
  return 1;
}

// Fused BasicBlock (bf16, 32 channels): returns 1 when the shape does not qualify.
template <int MBW1, int MBW2, int NW>
static int describe_block_one(const ConvParams& p, size_t lds, int grid_x, Launch* out) {
  static bool attr_set = false;
  const void* kern = reinterpret_cast<const void*>(&basic_block_c32_kernel<MBW1, MBW2, NW>);
  if (!attr_set) {
    UDP_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  out->fn = kern;
  out->grid = dim3(grid_x);
  out->block = dim3(NW * 64);
  out->lds = (unsigned)lds;
  out->p = p;
  return UDP_OK;
}

int describe_block(ConvParams p, int dtype, Launch* out) {
  if (dtype != UDP_BF16 || p.Cin != 32 || p.Cout != 32) return fail(UDP_ERR_UNSUPPORTED, "fused BasicBlock: bf16 with 32 channels only");
  if (p.Hin % kBlkR || p.Hin != p.Hout || p.Win != p.Wout) return fail(UDP_ERR_UNSUPPORTED, "fused BasicBlock: height must be a multiple of %d", kBlkR);
  p.R = kBlkR;
  p.IH = kBlkR + 4;
  p.IW = p.Win + 2;
  p.TW = p.Win;
  p.G = 1;
  const int rows_in = p.IH * p.IW, rows_mid = (kBlkR + 2) * p.IW;
  if (rows_in > kBlkMaxG * 64) return fail(UDP_ERR_UNSUPPORTED, "fused BasicBlock: map too wide (%d)", p.Win);
  const int m1 = (kBlkR + 2) * p.Win, m2 = kBlkR * p.Win;
  // 8 waves (two per SIMD) for the wide maps: one workgroup per CU has to hide its own LDS / MFMA latencies
  int nw = m2 > 128 && getenv("UDP_POSE_BLOCK_4W") == nullptr ? 8 : 4;
  if (m2 > 256 && getenv("UDP_POSE_BLOCK_16W") != nullptr) nw = 16;
  const int mbw1 = ceil_div(m1, nw * 16), mbw2 = ceil_div(m2, nw * 16);
  p.tiles_x = 1;
  p.tiles_y = p.Hin / kBlkR;
  p.ntiles = p.N * p.tiles_y;
  if (p.ntiles >= 65536) return fail(UDP_ERR_UNSUPPORTED, "fused BasicBlock: too many tiles");
  auto magic20 = [](int d) { return (unsigned)(((1u << 20) + d - 1) / d); };
  auto magic32 = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); };
  p.mIW = magic20(p.IW);
  p.mTW = magic20(p.Win);
  p.mTY = magic32(p.tiles_y);
  const size_t lds = 2 * 9 * 32 * ROWB + (size_t)((rows_mid + 15) / 16) * 16 * ROWB + 2 * (size_t)((rows_in + 15) / 16) * 16 * ROWB;
  if (lds > 160 * 1024) return fail(UDP_ERR_UNSUPPORTED, "fused BasicBlock: %zu bytes of LDS", lds);
  const int grid_x = p.ntiles < 256 ? p.ntiles : ceil_div(p.ntiles, ceil_div(p.ntiles, 256));
  if (nw == 16 && mbw1 == 2 && mbw2 == 2) return describe_block_one<2, 2, 16>(p, lds, grid_x, out);
  if (nw == 8 && mbw1 == 4 && mbw2 == 3) return describe_block_one<4, 3, 8>(p, lds, grid_x, out);   // 48 columns (256x192 input)
  if (nw == 8 && mbw1 == 2 && mbw2 == 2) return describe_block_one<2, 2, 8>(p, lds, grid_x, out);   // 24 columns
  if (nw == 4 && mbw1 == 8 && mbw2 == 6) return describe_block_one<8, 6, 4>(p, lds, grid_x, out);
  if (nw == 4 && mbw1 == 4 && mbw2 == 3) return describe_block_one<4, 3, 4>(p, lds, grid_x, out);
  if (nw == 4 && mbw1 == 3 && mbw2 == 2) return describe_block_one<3, 2, 4>(p, lds, grid_x, out);   // 16 columns (tests)
  if (nw == 4 && mbw1 == 2 && mbw2 == 1) return describe_block_one<2, 1, 4>(p, lds, grid_x, out);
  return fail(UDP_ERR_UNSUPPORTED, "fused BasicBlock: no kernel for %d columns", p.Win);
}

// Fills `out` with the kernel, grid and arguments of one fused conv (tile choice included).
int describe_conv(ConvParams p, int dtype, int ks, int stride, Launch* out) {
  if (p.Cin % 16 != 0) return fail(UDP_ERR_UNSUPPORTED, "conv Cin=%d is not a multiple of 16", p.Cin);
  if (!p.out_nchw_f32 && p.Cout % 16 != 0)
    return fail(UDP_ERR_UNSUPPORTED, "NHWC conv Cout=%d is not a multiple of 16", p.Cout);
  const size_t esz = dtype == UDP_F32 ? 4 : 2;                          // bytes per element of one plane
  const size_t pix_esz = dtype == UDP_BF16 ? 2 : 4;                     // bytes per element of a pixel (H2: two planes)
  if ((size_t)p.N * p.Hin * p.Win * p.in_pitch * pix_esz >= 0x7FFF0000u || (size_t)p.N * p.Hout * p.Wout * p.out_pitch * 4 >= 0x7FFF0000u ||
      (p.in_coff * esz) % 16 || (p.in_pitch * esz) % 16 ||
      (!p.out_nchw_f32 && ((p.out_coff * esz) % 16 || (p.out_pitch * esz) % 16)) ||
      (p.res && ((p.res_coff * esz) % 16 || (p.res_pitch * esz) % 16)))
    return fail(UDP_ERR_UNSUPPORTED, "conv channel views must be 16-byte aligned and tensors < 2 GiB");
  if (false ||
      p.N >= 2048)
    return fail(UDP_ERR_UNSUPPORTED, "conv tensor exceeds the 2 GiB the 32-bit buffer offsets cover; split the batch");
  if (p.in_stuff2 && (dtype == UDP_F16X2 || stride != 1 || (p.Hin & 1) || (p.Win & 1)))
    return fail(UDP_ERR_UNSUPPORTED, "in_stuff2: a stride-1 fp32 / bf16 conv over an even-sized stuffed image");
  if (p.nout2 && p.wfmt != 1) return fail(UDP_ERR_UNSUPPORTED, "second outputs need the weight-stationary split-fp16 conv (wfmt 1)");
  if (p.wfmt == 1) {
    if (dtype != UDP_F16X2) return fail(UDP_ERR_ARG, "wfmt 1 (fragment-major weights) needs UDP_F16X2");
    return describe_conv_ws(p, ks, stride, out);
  }
  if (p.wfmt != 0) return fail(UDP_ERR_ARG, "conv wfmt=%d", p.wfmt);
  int nb = 2;
  const size_t lds = conv_choose_tile(p, ks, stride, dtype, &nb);
  if (p.CoutPad % (nb * 16) != 0)
    return fail(UDP_ERR_UNSUPPORTED, "conv CoutPad=%d is not a multiple of %d", p.CoutPad, nb * 16);
  const int mbw = ceil_div(ceil_div(p.G * p.R * p.TW, 16), 4);
  if (getenv("UDP_POSE_DEBUG_TILES"))
    fprintf(stderr, "conv k%d s%d %dx%d C%d->%d: G=%d R=%d TW=%d NB=%d mbw=%d lds=%zu wgs=%d\n", ks, stride, p.Hout, p.Wout, p.Cin,
            p.Cout, p.G, p.R, p.TW, nb, mbw, lds, p.ntiles * (p.CoutPad / (nb * 16)));
  if (dtype == UDP_BF16 && getenv("UDP_POSE_NO_PERSIST") == nullptr) {
    const int rc = describe_persist(p, ks, stride, nb, mbw, out);
    if (rc <= 0) return rc;
  }
  if (dtype == UDP_F32) return describe_conv_t<float>(p, ks, stride, nb, mbw, lds, out);
  if (dtype == UDP_F16X2) return describe_conv_t<H2>(p, ks, stride, nb, mbw, lds, out);
  return describe_conv_t<__bf16>(p, ks, stride, nb, mbw, lds, out);
}

// A conv that will share one launch with the same-depth convs of other branches (hrnet.hip merges them
// into a conv_mfma_multi node): every member uses the instantiation <bf16, 3, 1, NB=2, MBW=4, 4 waves>.
// Returns 1 when the conv does not qualify (the caller falls back to describe_conv).
int describe_conv_grouped(ConvParams p, int dtype, int ks, int stride, Launch* out) {
  if (p.wfmt == 1 && dtype == UDP_F16X2 && stride == 1 && (ks == 1 || ks == 3) && !p.out_nchw_f32 && !p.nup) {
    // member of a conv_ws_multi launch when its tile comes out with 6 pixel blocks per wave (groupable != 0),
    // a launch of its own otherwise
    return describe_conv_ws(p, ks, stride, out, true);
  }
  if (p.wfmt != 0 || (dtype != UDP_BF16 && dtype != UDP_F16X2) || (ks != 3 && ks != 1) || stride != 1 || p.out_nchw_f32 || p.nup || p.Cin % 16 || p.Cout % 16 || p.CoutPad % 32 ||
      (p.in_stuff2 && dtype != UDP_BF16))
    return 1;
  const size_t pix_esz = dtype == UDP_BF16 ? 2 : 4;
  if ((size_t)p.N * p.Hin * p.Win * p.in_pitch * pix_esz >= 0x7FFF0000u || (size_t)p.N * p.Hout * p.Wout * p.out_pitch * pix_esz >= 0x7FFF0000u ||
      (p.in_coff * 2) % 16 || (p.in_pitch * 2) % 16 ||
      (p.out_coff * 2) % 16 || (p.out_pitch * 2) % 16 || (p.res && ((p.res_coff * 2) % 16 || (p.res_pitch * 2) % 16)) || p.N >= 2048)
    return 1;
  int nb = 2;
  const size_t lds = conv_choose_tile(p, ks, stride, dtype, &nb, true);
  if (nb != 2 || p.G * p.R * p.TW > 256) return 1;
  if (getenv("UDP_POSE_DEBUG_TILES"))
    fprintf(stderr, "grouped conv %dx%d C%d->%d: G=%d R=%d TW=%d NB=2 lds=%zu wgs=%d\n", p.Hout, p.Wout, p.Cin, p.Cout, p.G, p.R,
            p.TW, lds, p.ntiles * (p.CoutPad / 32));
  const int mbw = ceil_div(ceil_div(p.G * p.R * p.TW, 16), 4);
  const bool m3 = mbw <= 3 && getenv("UDP_POSE_MULTI_MBW4") == nullptr;
  int rc;
  if (dtype == UDP_F16X2) {
    if (ks == 3)
      rc = m3 ? describe_one<H2, 3, 1, 2, 3, false, 4>(p, lds, out) : describe_one<H2, 3, 1, 2, 4, false, 4>(p, lds, out);
    else
      rc = m3 ? describe_one<H2, 1, 1, 2, 3, false, 4>(p, lds, out) : describe_one<H2, 1, 1, 2, 4, false, 4>(p, lds, out);
  } else if (ks == 3) {
    rc = m3 ? describe_one<__bf16, 3, 1, 2, 3, false, 4>(p, lds, out) : describe_one<__bf16, 3, 1, 2, 4, false, 4>(p, lds, out);
  } else {
    rc = m3 ? describe_one<__bf16, 1, 1, 2, 3, false, 4>(p, lds, out) : describe_one<__bf16, 1, 1, 2, 4, false, 4>(p, lds, out);
  }
  if (rc) return rc;
  // storage type (0 bf16, 1 split fp16), kernel size, pixel blocks per wave of the instantiation that can run it
  out->groupable = (dtype == UDP_F16X2 ? 100 : 0) + ks * 10 + (m3 ? 3 : 4);
  return UDP_OK;
}

// Kernel + attribute for a merged launch of `n` groupable convs; fills the kernel argument.
int describe_multi(const Launch* members, int n, ConvMulti* m, Launch* out) {
  static bool attr_set = false;
  const void* kerns[8] = {reinterpret_cast<const void*>(&conv_mfma_multi<__bf16, 3, 1, 2, 3, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<__bf16, 3, 1, 2, 4, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<__bf16, 1, 1, 2, 3, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<__bf16, 1, 1, 2, 4, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<H2, 3, 1, 2, 3, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<H2, 3, 1, 2, 4, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<H2, 1, 1, 2, 3, 4>),
                          reinterpret_cast<const void*>(&conv_mfma_multi<H2, 1, 1, 2, 4, 4>)};
  if (!attr_set) {
    for (const void* k : kerns) UDP_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (n < 2 || n > 4) return fail(UDP_ERR_ARG, "describe_multi: %d members", n);
  if (members[0].groupable / 100 == 3) return describe_ws_multi(members, n, m, out);       // weight-stationary split-fp16 members
  const int h2 = members[0].groupable / 100;
  const int ks = members[0].groupable / 10 % 10;
  int mbw = 3;
  for (int j = 0; j < n; ++j) {
    if (members[j].groupable / 10 != members[0].groupable / 10) return fail(UDP_ERR_ARG, "describe_multi: members of different kernel sizes / storage types");
    if (members[j].groupable % 10 > mbw) mbw = members[j].groupable % 10;
  }
  const void* kern = kerns[4 * h2 + (ks == 3 ? 0 : 2) + (mbw == 3 ? 0 : 1)];
  memset(m, 0, sizeof(*m));
  unsigned total = 0, lds = 0;
  for (int j = 0; j < n; ++j) {
    m->p[j] = members[j].p;
    m->start[j] = total;
    m->tiles[j] = members[j].grid.x;
    total += members[j].grid.x * members[j].grid.y;
    if (members[j].lds > lds) lds = members[j].lds;
  }
  for (int j = n; j < 5; ++j) m->start[j] = j < 4 ? 0xFFFFFFFFu : total;
  for (int j = n; j < 4; ++j) m->tiles[j] = 1;
  out->fn = kern;
  out->grid = dim3(total);
  out->block = dim3(256);
  out->lds = lds;
  out->groupable = 0;
  return UDP_OK;
}

int describe_stem(const ConvParams& p, int dtype, Launch* out) {
  if (p.Cout != 64) return fail(UDP_ERR_UNSUPPORTED, "stem conv expects 64 output channels, got %d", p.Cout);
  const long total = (long)p.N * p.Hout * p.Wout;
  if (dtype == UDP_BF16 && getenv("UDP_POSE_STEM_VALU") == nullptr) {
    const long ntile = (total + 15) / 16;
    out->fn = reinterpret_cast<const void*>(&stem_mfma_kernel);
    out->grid = dim3((unsigned)((ntile + 3) / 4 < 4096 ? (ntile + 3) / 4 : 4096));
    out->block = dim3(256);
    out->lds = 0;
    out->p = p;
    return UDP_OK;
  }
  if (dtype == UDP_F16X2 && getenv("UDP_POSE_STEM_VALU") == nullptr) {
    const long ntile = (total + 15) / 16;
    out->fn = reinterpret_cast<const void*>(&stem_mfma_k<H2, 3>);
    out->grid = dim3((unsigned)((ntile + 3) / 4 < 2048 ? (ntile + 3) / 4 : 2048));
    out->block = dim3(256);
    out->lds = 1 * 4 * 2 * 1024;
    out->p = p;
    return UDP_OK;
  }
  out->fn = dtype == UDP_F32     ? reinterpret_cast<const void*>(&stem_conv_kernel<float>)
            : dtype == UDP_F16X2 ? reinterpret_cast<const void*>(&stem_conv_kernel<H2>)
                                 : reinterpret_cast<const void*>(&stem_conv_kernel<__bf16>);
  out->grid = dim3((unsigned)((total + 63) / 64));
  out->block = dim3(256);
  out->lds = 0;
  out->p = p;
  return UDP_OK;
}

int describe_fuse(const ConvParams& p, int dtype, Launch* out) {
  if (p.Cout % 4 != 0) return fail(UDP_ERR_UNSUPPORTED, "fuse Cout=%d is not a multiple of 4", p.Cout);
  const long total = (long)p.N * p.Hout * p.Wout * (p.Cout / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  out->fn = dtype == UDP_F32     ? reinterpret_cast<const void*>(&fuse_sum_kernel<float>)
            : dtype == UDP_F16X2 ? reinterpret_cast<const void*>(&fuse_sum_kernel<H2>)
                                 : reinterpret_cast<const void*>(&fuse_sum_kernel<__bf16>);
  out->grid = dim3((unsigned)blocks);
  out->block = dim3(256);
  out->lds = 0;
  out->p = p;
  return UDP_OK;
}

#ifdef UDP_STAMPS
extern "C" int udp_debug_set_stamps(unsigned long long* dev_buf) {
  UDP_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dev_buf, sizeof(dev_buf)));
  return ws_set_stamps(dev_buf);          // every translation unit has its own copy of the pointer
}
#endif

template <typename KF, typename KB, typename KH>
static int describe_elementwise(const ConvParams& p, int dtype, KF kf, KB kb, KH kh, Launch* out) {
  if (p.Cout % 4 != 0) return fail(UDP_ERR_UNSUPPORTED, "element-wise op: C=%d is not a multiple of 4", p.Cout);
  const long total = (long)p.N * p.Hout * p.Wout * (p.Cout / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  out->fn = dtype == UDP_F32 ? reinterpret_cast<const void*>(kf)
                             : dtype == UDP_F16X2 ? reinterpret_cast<const void*>(kh) : reinterpret_cast<const void*>(kb);
  out->grid = dim3((unsigned)blocks);
  out->block = dim3(256);
  out->lds = 0;
  out->p = p;
  return UDP_OK;
}

int describe_maxpool(const ConvParams& p, int dtype, Launch* out) {
  return describe_elementwise(p, dtype, &maxpool3_kernel<float>, &maxpool3_kernel<__bf16>, &maxpool3_kernel<H2>, out);
}

int describe_bilinear(const ConvParams& p, int dtype, Launch* out) {
  return describe_elementwise(p, dtype, &bilinear_ac_kernel<float>, &bilinear_ac_kernel<__bf16>, &bilinear_ac_kernel<H2>, out);
}

// 7x7 stride-2 stem (RSN, network.py:400-412) with the input window staged through LDS.  stem_mfma_k<T, 7> gathers a
// lane's 40 K values per 16-pixel tile straight from global memory (scattered 4-byte loads: 670 us per 128 images, 12 %
// of the RSN-18 step, against 472 MB of input + output = ~120 us of HBM time).  Here a workgroup owns a tile of 2 output
// rows x 32 output columns: its 9 x 69 x 3 fp32 input window is loaded coalesced (register-staged one tile ahead,
// mirrored columns for the flip-test half, zeros outside the image), and the lanes gather from LDS.  Same GEMM
// (K = 147 -> 160, five k-steps, weights fragment-major in LDS), same summation order, same results.
template <typename T>
__global__ __launch_bounds__(256) void stem7_lds_kernel(const ConvParams p) {
  constexpr bool SPLIT = std::is_same<T, H2>::value;
  constexpr int KS = 7, K = KS * KS * 3, NS = (K + 31) / 32, PAD = 3, WPL = SPLIT ? 2 : 1;
  constexpr int TR = 2, TC = 32, WR = 2 * TR + 5, WC = 2 * TC + 5, WP = 72, WIN = 3 * WR * WC, PER = (WIN + 255) / 256;
  using E = typename std::conditional<SPLIT, _Float16, __bf16>::type;
  using Frag = typename std::conditional<SPLIT, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* wl = reinterpret_cast<E*>(smem);
  float* win = reinterpret_cast<float*>(smem + NS * 4 * WPL * 1024);       // [ch][WR][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const float* wg = reinterpret_cast<const float*>(p.wgt);
  for (int e = tid; e < NS * 4 * WPL * 512; e += 256) {
    const int j = e & 7, ln = (e >> 3) & 63, pl = (e >> 9) % WPL, nb = ((e >> 9) / WPL) & 3, st = (e >> 9) / (WPL * 4);
    const int k = 32 * st + 8 * (ln >> 4) + j;
    const int cout = 16 * ((ln & 15) >> 2) + 4 * nb + (ln & 3);
    const float w = k < K ? wg[k * 64 + cout] : 0.f;
    if constexpr (SPLIT) {
      const _Float16 hi = (_Float16)w;
      wl[e] = pl == 0 ? hi : (_Float16)((w - (float)hi) * kLoScale);
    } else {
      wl[e] = (__bf16)w;
    }
  }
  const int cbase = 16 * kg;
  f32x4 bias[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bias[nb] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4 * nb);
  // per k-step the lane's 8 K values as 16-bit window offsets (ch * WR + ky) * WP + kx, 0xFFFF = K padding
  unsigned koff[NS][4];
#pragma unroll
  for (int st = 0; st < NS; ++st)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      unsigned v = 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = 32 * st + 8 * kg + 2 * h + q, tap = k / 3;
        const unsigned c = k < K ? (unsigned)(((k % 3) * WR + tap / KS) * WP + tap % KS) : 0xFFFFu;
        v |= c << (16 * q);
      }
      koff[st][h] = v;
    }
  const int tiles_x = (p.Wout + TC - 1) / TC, tiles_y = (p.Hout + TR - 1) / TR;
  const long ntile = (long)p.N * tiles_y * tiles_x;
  const size_t plane = (size_t)p.Hin * p.Win;
  // the window of tile t, this thread's PER elements (coalesced along x), into registers
  auto fetch = [&](long t, float (&v)[PER]) __attribute__((always_inline)) {
    const int tx = (int)(t % tiles_x);
    const long t2 = t / tiles_x;
    const int ty = (int)(t2 % tiles_y), n = (int)(t2 / tiles_y);
    const bool mirror = n >= p.flip_from;
    const float* in = reinterpret_cast<const float*>(p.in) + (size_t)(mirror ? n - p.flip_from : n) * 3 * plane;
    const int gy0 = ty * TR * 2 - PAD, gx0 = tx * TC * 2 - PAD;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      const int ch = e / (WR * WC), rem = e - ch * (WR * WC);
      const int iy = rem / WC, ix = rem - iy * WC;
      const int gy = gy0 + iy, gx = gx0 + ix;
      const bool ok = e < WIN && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
      v[i] = ok ? in[(size_t)ch * plane + (size_t)gy * p.Win + (mirror ? p.Win - 1 - gx : gx)] : 0.f;
    }
  };
  auto put = [&](const float (&v)[PER]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      if (e < WIN) {
        const int ch = e / (WR * WC), rem = e - ch * (WR * WC);
        const int iy = rem / WC, ix = rem - iy * WC;
        win[(ch * WR + iy) * WP + ix] = v[i];
      }
    }
  };
  const int r = wave >> 1, xl = (wave & 1) * 16 + li;          // the lane's output pixel inside the tile
  const int wbase = (2 * r) * WP + 2 * xl;
  float pre[PER];
  long tile = blockIdx.x;
  if (tile < ntile) fetch(tile, pre);
  for (; tile < ntile; tile += gridDim.x) {
    __syncthreads();                       // the previous tile's gathers are done (and, first time, the weights are in)
    put(pre);
    __syncthreads();
    if (tile + gridDim.x < ntile) fetch(tile + gridDim.x, pre);          // next window flies under this tile's MFMAs
    const int tx = (int)(tile % tiles_x);
    const long t2 = tile / tiles_x;
    const int ty = (int)(t2 % tiles_y), n = (int)(t2 / tiles_y);
    const int yo = ty * TR + r, xo = tx * TC + xl;
    const bool okp = yo < p.Hout && xo < p.Wout;
    f32x4 acc[4], accx[SPLIT ? 4 : 1];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = bias[nb];
#pragma unroll
    for (int nb = 0; nb < (SPLIT ? 4 : 1); ++nb) accx[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      Frag hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned ko = (koff[st][j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        const float v = ko != 0xFFFFu ? win[wbase + (int)ko] : 0.f;
        hi[j] = (E)v;
        lo[j] = (E)((v - (float)hi[j]) * (SPLIT ? kLoScale : 1.f));
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const Frag wh = *reinterpret_cast<const Frag*>(wl + ((st * 4 + nb) * WPL) * 512 + lane * 8);
        if constexpr (SPLIT) {
          const Frag wlo = *reinterpret_cast<const Frag*>(wl + ((st * 4 + nb) * WPL + 1) * 512 + lane * 8);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hi, acc[nb], 0, 0, 0);
          accx[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, lo, accx[nb], 0, 0, 0);
          accx[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, hi, accx[nb], 0, 0, 0);
        } else {
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hi, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, lo, acc[nb], 0, 0, 0);
        }
      }
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] += accx[nb] * kLoInv;
    }
    if (p.relu) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[nb][q] = acc[nb][q] > 0.f ? acc[nb][q] : 0.f;
    }
    if (okp) {
      const size_t pix = ((size_t)n * p.Hout + yo) * p.Wout + xo;
      if constexpr (SPLIT) {
        unsigned char* o = reinterpret_cast<unsigned char*>(p.out) + pix * 256 + cbase * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f16x8 oh, ol;
          h2_split8(acc[2 * h], acc[2 * h + 1], oh, ol);
          *reinterpret_cast<f16x8*>(o + 16 * h) = oh;
          *reinterpret_cast<f16x8*>(o + 128 + 16 * h) = ol;
        }
      } else {
        __bf16* o = reinterpret_cast<__bf16*>(p.out) + pix * 64 + cbase;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 ov;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ov[q] = (__bf16)acc[2 * h][q];
            ov[4 + q] = (__bf16)acc[2 * h + 1][q];
          }
          *reinterpret_cast<bf16x8*>(o + 8 * h) = ov;
        }
      }
    }
  }
}

int describe_stem7(const ConvParams& p, int dtype, Launch* out) {
  if (p.Cout != 64) return fail(UDP_ERR_UNSUPPORTED, "7x7 stem expects 64 output channels, got %d", p.Cout);
  const long total = (long)p.N * p.Hout * p.Wout;
  if (dtype != UDP_F32 && getenv("UDP_POSE_STEM_VALU") == nullptr) {
    // bf16 / split fp16: the 7x7 stem as a K = 147 -> 160 GEMM on the matrix pipe
    static bool attr_set = false;
    const bool gather = getenv("UDP_POSE_STEM7_GATHER") != nullptr;        // A/B: the global-gather form
    const void* kb = gather ? reinterpret_cast<const void*>(&stem_mfma_k<__bf16, 7>) : reinterpret_cast<const void*>(&stem7_lds_kernel<__bf16>);
    const void* kh = gather ? reinterpret_cast<const void*>(&stem_mfma_k<H2, 7>) : reinterpret_cast<const void*>(&stem7_lds_kernel<H2>);
    if (!attr_set) {
      for (const void* k : {reinterpret_cast<const void*>(&stem_mfma_k<__bf16, 7>), reinterpret_cast<const void*>(&stem_mfma_k<H2, 7>),
                            reinterpret_cast<const void*>(&stem7_lds_kernel<__bf16>), reinterpret_cast<const void*>(&stem7_lds_kernel<H2>)})
        UDP_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_set = true;
    }
    out->fn = dtype == UDP_F16X2 ? kh : kb;
    out->block = dim3(256);
    out->lds = 5 * 4 * (dtype == UDP_F16X2 ? 2 : 1) * 1024;
    out->p = p;
    if (gather) {
      const long ntile = (total + 15) / 16;
      out->grid = dim3((unsigned)((ntile + 3) / 4 < 1024 ? (ntile + 3) / 4 : 1024));
    } else {
      const long ntile = (long)p.N * ((p.Hout + 1) / 2) * ((p.Wout + 31) / 32);
      out->grid = dim3((unsigned)(ntile < 1024 ? ntile : 1024));
      out->lds += 3 * 9 * 72 * sizeof(float);
    }
    return UDP_OK;
  }
  out->fn = dtype == UDP_F32     ? reinterpret_cast<const void*>(&stem7_conv_kernel<float>)
            : dtype == UDP_F16X2 ? reinterpret_cast<const void*>(&stem7_conv_kernel<H2>)
                                 : reinterpret_cast<const void*>(&stem7_conv_kernel<__bf16>);
  out->grid = dim3((unsigned)((total + 63) / 64));
  out->block = dim3(256);
  out->lds = (147 * 64 + 64) * sizeof(float);
  out->p = p;
  return UDP_OK;
}

int run_launch(const Launch& l, hipStream_t s) {
  void* args[] = {const_cast<ConvParams*>(&l.p)};
  UDP_HIP_CHECK(hipLaunchKernel(l.fn, l.grid, l.block, args, l.lds, s));
  return UDP_OK;
}

int conv_h2_overflow(hipStream_t s, int reset, int* flag) { return h2_overflow_fetch(s, reset, flag); }

}  // namespace udp
