// Weight-stationary split-fp16 convolution kernels of the HRNet / RSN forward for gfx950 (MI355X, CDNA4) and their
// host-side tile choice, dispatch and merged-launch ordering.  Split from conv.hip (same op:
//     out = act( conv(in) + bias [+ res] [+ sum_k nearest_up(up_k)] ),
// deep_hrnet/lib/models/pose_hrnet.py:43-59, :80-100, :189-273, :344-383).
#include "conv_dev.h"

namespace udp {

// ---------------------------------------------------------------------------------------------
// Weight-stationary form for split-fp16 convs ("ws"): the MFMA A operand (weights) never passes through
// LDS.  A wave owns one pair of 16-cout blocks and keeps that pair's fragments of the current tap row
// (KS taps x 2 blocks x hi/lo) in registers -- loaded straight from global memory (L2-resident), one tap
// row ahead -- and sweeps PB pixel blocks under them; LDS holds only the input halo tile (hi + lo image,
// double-buffered over the K chunks).  Against conv_mfma_kernel<H2> (which stages 9*BN weight rows per
// chunk next to the tile) this halves the LDS-DMA instructions, leaves LDS room for two workgroups per CU
// at any K depth, drops the LDS reads per MFMA from 10/18 to 2/6, and a staged input tile serves 4 cout
// pairs (128 channels) instead of one.
//   workgroup = 4 waves = CP cout pairs x PG = 4/CP pixel groups; wave (cp, pg) computes pixel blocks
//   pg*PB .. pg*PB+PB-1 of the tile (M <= 16*PB*PG pixels) x couts (blockIdx.y*CP + cp)*32 .. +31.
// ---------------------------------------------------------------------------------------------
// residual of one pixel block (8 consecutive split-fp16 channels per lane) held raw from the prologue to the epilogue
template <int NB>
struct ResH2 {
  u32x4 hi[NB / 2], lo[NB / 2];
};
template <int NB>
__device__ __forceinline__ void load_res_h2(ResH2<NB>& o, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned lo_off) {
#pragma unroll
  for (int h = 0; h < NB / 2; ++h) {
    o.hi[h] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * h, 0, UDP_H2_RES_AUX);
    o.lo[h] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + lo_off + 16 * h, 0, UDP_H2_RES_AUX);
  }
}
template <int NB>
__device__ __forceinline__ void add_res_h2(f32x4 (&v)[NB], const ResH2<NB>& o) {
#pragma unroll
  for (int h = 0; h < NB / 2; ++h) h2_add8(v[2 * h], v[2 * h + 1], __builtin_bit_cast(f16x8, o.hi[h]), __builtin_bit_cast(f16x8, o.lo[h]));
}

template <int KS, int STRIDE, int PB, int CP, bool NCHW, bool OUT2 = false>
__device__ __forceinline__ void conv_ws_body(const ConvParams& p, const int tile_id, const int cby) {
  using T = H2;
  constexpr int CK = 32, ESZ = 2, NB = 2, NW = 4;
  constexpr int PAD = KS / 2, TAPS = KS * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  UDP_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kg = lane >> 4;
  const int cp = wave % CP, pg = wave / CP;          // 4 waves = CP cout pairs x 4/CP pixel groups

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
  const int stage_bytes = 2 * in_bytes;             // [hi image][lo image]
  const int nchunks = (p.Cin + CK - 1) / CK;
  const bool ragged = (p.Cin % CK) != 0;
  const int RT = p.R * p.TW;
  const int M = p.G * RT;
  const unsigned inpb = (unsigned)p.in_pitch * ESZ * 2;
  const unsigned outpb = (unsigned)p.out_pitch * ESZ * 2, respb = (unsigned)p.res_pitch * ESZ * 2;
  const unsigned in_lo = (unsigned)p.in_pitch * ESZ;
  const unsigned out_lo = (unsigned)p.out_pitch * ESZ, res_lo = (unsigned)p.res_pitch * ESZ;

  const unsigned out_pix = (unsigned)p.N * p.Hout * p.Wout;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.in), 0, (unsigned)p.N * p.Hin * p.Win * inpb, 0x00020000);
  const unsigned npairs = (unsigned)p.CoutPad >> 5;
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.wgt), 0, (unsigned)TAPS * nchunks * npairs * 4096u, 0x00020000);
  // (no `out`: second outputs only, udp_conv_op.n_out2 -- a zero-length descriptor drops the stores)
#if UDP_WS_DBG & 32      // (timing-only build: the one-chunk convs store nothing)
  const bool dbg_nost = nchunks == 1;
#else
  constexpr bool dbg_nost = false;
#endif
#if UDP_WS_DBG & 64      // (timing-only build: the one-chunk convs read no residual)
  const bool dbg_nores = nchunks == 1;
#else
  constexpr bool dbg_nores = false;
#endif
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (!p.out || dbg_nost) ? 0 : NCHW ? out_pix * p.Cout * 4 : out_pix * outpb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bias), 0, (unsigned)p.CoutPad * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.res), 0, (p.res && !dbg_nores) ? out_pix * respb : 0, 0x00020000);

  const int cwave = (cby * CP + cp) * 32;   // first cout of the wave's pair of blocks
  const int cbase = cwave + 8 * kg;                // the lane's 8 consecutive output channels
  // ---- weights (fragment-major, udp_conv_op.wfmt == 1): the 1 KiB block (tap, chunk, cout pair, nb, plane) holds
  // lane l's 8 K values at 16*l -- A-fragment row li of block nb is cout cwave + 8*(li>>2) + 4*nb + (li&3), so
  // that a lane ends up with 8 consecutive couts.  One contiguous load per fragment; cin is zero-padded.
  const unsigned wvoff = (unsigned)lane * 16u;
  const unsigned wpair = (unsigned)(cby * CP + cp) * 4096u;
  // A fragments of one tap (2 blocks x hi/lo = 16 registers) in a ring of three: the fragments of step s + 2
  // stream in from L2 while step s feeds the MFMAs (step = one tap of one K chunk)
  constexpr int AD = UDP_WS_AD;          // depth of the A ring (fragments of step s + AD - 1 are in flight)
  f16x8 ah[AD][NB], al[AD][NB];
  const int nsteps = nchunks * TAPS;
  auto load_a = [&](int s, f16x8 (&h)[NB], f16x8 (&l)[NB]) __attribute__((always_inline)) {
    const int c = s / TAPS, tap = s - c * TAPS;
    const unsigned soff = (unsigned)(tap * nchunks + c) * (npairs * 4096u) + wpair;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      h[nb] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, wvoff + 2048u * nb, soff, 0));
      l[nb] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, wvoff + 2048u * nb + 1024u, soff, 0));
    }
  };

  // The prologue issues every load it can before it waits for any: A fragments of steps 0 and 1 and the bias first
  // (L2), then the chunk-0 tile DMA, then the residual (consumed in the epilogue) -- ONE memory round trip before
  // the first MFMA.  (The in-kernel timeline had three in a row: the bias, used to initialise the accumulators, made
  // hipcc wait for the DMA issued before it; only then were the A and residual loads issued, and the residual was
  // added -- waited for -- before the loop: 4.2 us of every 15-68 us workgroup lifetime, tools/stamp_multi.py.)
#pragma unroll
  for (int k = 0; k < AD - 1; ++k) load_a(k < nsteps ? k : nsteps - 1, ah[k], al[k]);
  f32x4 bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    bias[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bias, (unsigned)(cbase + 4 * nb) * 4u, 0, 0));

  // ---- per-lane DMA offsets of the input halo tile (chunk 0); masked rows -> kOobOff -> zeros
  const int srow = lane >> 2, spart = lane & 3;
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
      const bool ok = row < npix_in && n < p.N && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
      const unsigned pix = __umul24(__umul24(n, p.Hin) + gy, p.Win) + gx;
      off = ok ? pix * inpb + p.in_coff * ESZ + ((spart ^ swz<T>(row)) << 4) : kOobOff;
    }
    src_off[i] = off;
  }
  auto stage = [&](int c, unsigned char* sb) __attribute__((always_inline)) {
    const unsigned coff = (unsigned)c * (CK * ESZ);
    const bool cut = ragged && c == nchunks - 1;
    const int parts_left = (p.Cin - c * CK) / 8;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int gidx = wave + NW * i;
      if (gidx < in_groups) {
        unsigned off = src_off[i] + coff;
        if (cut && (spart ^ swz<T>(gidx * 16 + srow)) >= parts_left) off = kOobOff;
        blds16(r_in, off, sb + gidx * (16 * ROWB));
        blds16(r_in, off + in_lo, sb + in_bytes + gidx * (16 * ROWB));
      }
    }
  };

  UDP_STAMP(1);
  stage(0, smem);      // chunk 0 is on its way while the rest of the per-lane state is set up

  // ---- the lane's PB output pixels
  int prow[PB], opix[PB], ocrd[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int m0 = (pg * PB + i) * 16 + li;
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

  // ONE accumulator set: the weights are stored scaled by 2^wexp (power of two, exact) so that their largest
  // magnitude sits in [2^13, 2^14): the lo plane then holds the plain fp16 residual w*2^wexp - hi (no second scale
  // needed to keep it normal), and the three MFMAs of a product -- hi*Xhi, lo*Xhi, (hi*2^-11)*Xlo' -- all add into
  // the same fp32 accumulator, which carries conv*2^wexp; the epilogue computes acc*2^-wexp + bias (+ residual).
  // Against the earlier pair of accumulators (cross terms apart, folded in at the end) this frees 8*PB registers,
  // which now hold the residual from the prologue to the epilogue.
  const float winv = __builtin_ldexpf(1.f, -p.wexp);
  f32x4 acc[PB][NB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // residual: issued right behind the chunk-0 DMA (no res: a zero-length descriptor, the loads return zeros -- no
  // branch around loads, see the A ring), added in the epilogue.  The barriers keep hipcc from moving these loads in
  // front of the DMA: the first wait below counts them as YOUNGER than it.
  constexpr int NR = NCHW ? 0 : PB * NB;          // buffer loads per lane
  ResH2<NB> rres[NCHW ? 1 : PB];
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  if constexpr (!NCHW) {
#pragma unroll
    for (int i = 0; i < PB; ++i)
      load_res_h2(rres[i], r_res, opix[i] >= 0 ? (unsigned)opix[i] * respb + (p.res_coff + cbase) * ESZ : kOobOff, res_lo);
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  UDP_STAMP(2);
  int c = 0, tap = 0;                     // chunk / tap of the current step
  const unsigned char* sb = smem;
  auto step = [&](auto BUFC, int s) __attribute__((always_inline)) {
    constexpr int BUF = decltype(BUFC)::value;
    // the A loads go first: vector-memory operations complete in issue order, fragments queued behind the
    // next chunk's DMA would wait for it
    // (issued unconditionally -- past the end the last step's fragments are fetched again: a prefetch under a
    // branch makes hipcc assume nothing newer is in flight and wait with vmcnt(0) at every use)
#if !(UDP_WS_DBG & 1)
    load_a(s + AD - 1 < nsteps ? s + AD - 1 : nsteps - 1, ah[(BUF + AD - 1) % AD], al[(BUF + AD - 1) % AD]);
#endif
    if (tap == 0) {
      // chunk c's DMA has landed (vector-memory operations complete in issue order: all but the 2*NB A loads
      // just issued, which may stay in flight; letting the previous step's stay in flight too changes nothing)
      if (c == 1) UDP_STAMP(9);                           // (diagnostic builds only: chunk-boundary stamps of chunks 1, 2, 4)
      if (c == 2) UDP_STAMP(11);
      if (c == 4) UDP_STAMP(13);
      // (at s == 0 the residual loads, issued behind the chunk-0 DMA, may stay in flight as well)
      bool refill = false;
      if constexpr (STRIDE == 2 || OUT2) refill = p.sbuf && c > 0;   // (not in the merged kernel: its members keep two stages)
      if (refill) {
        // ONE stage buffer (convs launched on their own whose two stage buffers exceed half the LDS -- stride-2 convs with
        // several K chunks: 4x input tile, 110 KB; 3x3 convs on 48-pixel-wide rows of 64+ channels: 128 KB -- i.e. one
        // workgroup, four waves, per CU; with one buffer two workgroups fit and hide each other's DMA):
        // refill it once every wave has left chunk c - 1, wait for it (the A loads just issued are older), go on
        __syncthreads();
        stage(c, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sb = smem;
      } else {
        if (s == 0)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB + NR) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");
        if (s == 0) UDP_STAMP(3);
        __syncthreads();                                    // ... for every wave; nobody reads the other stage any more
        if (s == 0) UDP_STAMP(4);
        if (c == 1) UDP_STAMP(10);
        if (c == 2) UDP_STAMP(12);
        if (c == 4) UDP_STAMP(14);
#if !(UDP_WS_DBG & 4)
        if (c + 1 < nchunks && !((STRIDE == 2 || OUT2) && p.sbuf)) stage(c + 1, smem + ((c + 1) & 1) * stage_bytes);
#endif
        sb = smem + (c & 1) * stage_bytes;
      }
    }
    // the cross term hi * Xlo: activations keep their lo plane scaled by 2^11 (storage format), so the weight side
    // carries 2^-11 -- four packed fp16 multiplies per fragment and step (exact unless the product goes subnormal,
    // i.e. for weights below 2^-17 of the layer's largest: an absolute error of 2^-25 there)
    f16x8 a2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) a2[nb] = ah[BUF][nb] * (_Float16)0x1p-11f;
    const int tap_rows = KS == 1 ? 0 : (tap / KS) * IW + tap % KS;
    // B fragments (hi, lo) in a ring of three: the LDS reads of block i + 2 are issued before the MFMAs of
    // block i (one wave per SIMD has nothing else to cover the ~200-cycle LDS latency with)
    f16x8 xh[3], xl[3];
    auto load_b = [&](int i, f16x8& h, f16x8& l) __attribute__((always_inline)) {
      const int row = prow[i] + tap_rows;
      const unsigned char* q = sb + row * ROWB + ((kg ^ swz<T>(row)) << 4);
      h = *reinterpret_cast<const f16x8*>(q);
      l = *reinterpret_cast<const f16x8*>(q + in_bytes);
    };
    load_b(0, xh[0], xl[0]);
    if (PB > 1) load_b(1, xh[1], xl[1]);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
#if !(UDP_WS_DBG & 2)
      if (i + 2 < PB) load_b(i + 2, xh[(i + 2) % 3], xl[(i + 2) % 3]);
#endif
      __builtin_amdgcn_sched_barrier(0);   // keep those LDS reads ahead of this block's MFMAs (hipcc sinks them otherwise)
#if UDP_WS_DBG & 16      // (timing-only build: one MFMA instead of six per block, everything else in place)
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[BUF][0] + al[BUF][1] + a2[0] + a2[1], xh[i % 3] + xl[i % 3], acc[i][0], 0, 0, 0);
#else
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[BUF][nb], xh[i % 3], acc[i][nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[BUF][nb], xh[i % 3], acc[i][nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[nb], xl[i % 3], acc[i][nb], 0, 0, 0);
#endif
    }
    if (++tap == TAPS) {
      tap = 0;
      ++c;
    }
  };
  for (int s = 0; s < nsteps; s += AD) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
    if (s + 2 < nsteps) step(std::integral_constant<int, 2>{}, s + 2);
    if constexpr (AD > 3) if (s + 3 < nsteps) step(std::integral_constant<int, 3>{}, s + 3);
    if constexpr (AD > 4) if (s + 4 < nsteps) step(std::integral_constant<int, 4>{}, s + 4);
    if constexpr (AD > 5) if (s + 5 < nsteps) step(std::integral_constant<int, 5>{}, s + 5);
  }
  UDP_STAMP(5);
  // conv = acc * 2^-wexp; + bias; + residual (hi + lo * 2^-11)
#pragma unroll
  for (int i = 0; i < PB; ++i) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][nb][q] = __builtin_fmaf(acc[i][nb][q], winv, bias[nb][q]);
    if constexpr (!NCHW) add_res_h2(acc[i], rres[i]);
  }

  // ---- epilogue: lane holds couts cbase .. cbase + 7 of pixel i
#pragma unroll
  for (int i = 0; i < PB; ++i) {
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
            const int sh = p.up_shift[u];
            const long up_pix = ((long)(n * (p.Hout >> sh) + (y >> sh)) * (p.Wout >> sh) + (xo >> sh));
            if (crd >= 0)
              add_vec<T, NB>(v, reinterpret_cast<const unsigned char*>(p.up[u]) + (up_pix * p.Cout * 2 + cbase) * ESZ, p.Cout * ESZ);
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
      // (only in the kernels of convs launched on their own: with this code the merged kernel spills 4 KB per lane)
      if (OUT2 && p.nout2) {   // wave-uniform, rare (RSN bottleneck): out2_k = out AS STORED + add2_k (udp_conv_op.n_out2)
        f16x8 sh, sl;
        h2_split8(v[0], v[1], sh, sl);
        f32x4 vr[NB];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vr[0][q] = (float)sh[q] + (float)sl[q] * kLoInv;
          vr[1][q] = (float)sh[4 + q] + (float)sl[4 + q] * kLoInv;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u < p.nout2 && opix[i] >= 0) {
            const size_t px = (size_t)opix[i];
            const _Float16* a = reinterpret_cast<const _Float16*>(p.add2[u]) + px * (2 * (size_t)p.add2_pitch[u]) + p.add2_coff[u] + cbase;
            const f16x8 ah8 = *reinterpret_cast<const f16x8*>(a), al8 = *reinterpret_cast<const f16x8*>(a + p.add2_pitch[u]);
            f32x4 w[NB];
#pragma unroll
            for (int q = 0; q < 4; ++q) {      // fuse_sum_kernel's order: addend first, then the conv output
              w[0][q] = ((float)ah8[q] + (float)al8[q] * kLoInv) + vr[0][q];
              w[1][q] = ((float)ah8[4 + q] + (float)al8[4 + q] * kLoInv) + vr[1][q];
            }
            f16x8 oh, ol;
            h2_split8(w[0], w[1], oh, ol);
            _Float16* o2 = reinterpret_cast<_Float16*>(p.out2[u]) + px * (2 * (size_t)p.out2_pitch[u]) + p.out2_coff[u] + cbase;
            *reinterpret_cast<f16x8*>(o2) = oh;
            *reinterpret_cast<f16x8*>(o2 + p.out2_pitch[u]) = ol;
          }
        }
      }
    }
  }
  UDP_STAMP(6);
#ifdef UDP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UDP_STAMP(7);
#endif
}

// Flat workgroup index of a conv -> (tile, cout block).  The cout blocks of one pixel tile stage the same input
// tile: they are placed 8 workgroups apart -- same XCD (workgroup i runs on XCD i % 8), started together -- so the
// second read of the tile hits that XCD's L2 instead of HBM: groups of 8 tiles x all cout blocks, cout-block-major
// inside the group (the last ntiles % 8 tiles form a smaller group).  Wave-uniform.
__host__ __device__ __forceinline__ void ws_decode(unsigned b, unsigned ntiles, unsigned nc, int& tile, int& cby) {
  if (nc == 1) {
    tile = (int)b;
    cby = 0;
    return;
  }
  const unsigned full = (ntiles >> 3) * 8u * nc;
  unsigned base = 0, width = 8, r = b;
  if (b < full) {
    const unsigned g = b / (8u * nc);
    r = b - g * 8u * nc;
    base = g * 8u;
  } else {
    r = b - full;
    base = (ntiles >> 3) * 8u;
    width = ntiles & 7u;
  }
  const unsigned c = r / width;
  cby = (int)c;
  tile = (int)(base + r - c * width);
}

template <int KS, int STRIDE, int PB, int CP, bool NCHW>
__global__ __launch_bounds__(256, 2) void conv_ws_h2_kernel(const ConvParams p) {
  int tile, cby;
  ws_decode(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, tile, cby);   // dispatch order = flat index
  conv_ws_body<KS, STRIDE, PB, CP, NCHW, KS == 3 && STRIDE == 1 && !NCHW>(p, tile, cby);
}

// Merged launch of up to 4 independent weight-stationary convs (same-depth convs of different HRNet branches):
// every member runs the 6-pixel-blocks-per-wave body with its own cout-pair split (ConvMulti::code = CP).
#ifndef UDP_WS_MPB
#define UDP_WS_MPB 6      // pixel blocks per wave of the merged kernel's members
#define UDP_WS_MOCC 2     // workgroups per CU the merged kernel is compiled for (waves per SIMD)
#endif
template <int KS>
__global__ __launch_bounds__(256, UDP_WS_MOCC) void conv_ws_multi(const ConvMulti m) {
  const unsigned b = blockIdx.x;
  int j, tile, cby;
  // (readfirstlane: the dynamically indexed kernel-argument reads are uniform, the compiler does not see it and
  // would wrap every weight load of the body in a waterfall loop)
  int code;
  // the table entry is fetched together with tab_n (its address only needs blockIdx; entry 0 when out of range)
  const unsigned e = __builtin_amdgcn_readfirstlane(m.tab[(b >> 3) < (unsigned)kMultiTab ? (b >> 3) : 0u]);
  if (m.tab_n) {
    j = (int)(e & 3u);
    code = 1 << ((e >> 2) & 3u);
    cby = (int)((e >> 4) & 63u);
    tile = (int)((e >> 10) + (b & 7u));
  } else {
    int sg = 0;
#pragma unroll
    for (int k = 1; k < kMultiSegs; ++k) sg += b >= m.seg_start[k];
    j = __builtin_amdgcn_readfirstlane(m.seg_mem[sg]);
    const unsigned r = __builtin_amdgcn_readfirstlane(b - m.seg_start[sg] + m.seg_first[sg]);
    ws_decode(r, m.tiles[j], m.ncby[j], tile, cby);
    tile = __builtin_amdgcn_readfirstlane(tile);
    cby = __builtin_amdgcn_readfirstlane(cby);
    code = m.code[j];
  }
  switch (code) {
    case 1: conv_ws_body<KS, 1, UDP_WS_MPB, 1, false>(m.p[j], tile, (int)cby); break;
    case 2: conv_ws_body<KS, 1, UDP_WS_MPB, 2, false>(m.p[j], tile, (int)cby); break;
    default: conv_ws_body<KS, 1, UDP_WS_MPB, 4, false>(m.p[j], tile, (int)cby); break;
  }
}

// Weight-stationary split-fp16 conv (conv_ws_h2_kernel): tile choice + dispatch.  Returns 1 when the
// shape does not qualify (the caller falls back to conv_mfma_kernel<H2>).
template <int KS, int STRIDE, int PB, int CP, bool NCHW>
static int describe_ws_one(const ConvParams& p, size_t lds, Launch* out) {
  static bool attr_set = false;
  const void* kern = reinterpret_cast<const void*>(&conv_ws_h2_kernel<KS, STRIDE, PB, CP, NCHW>);
  if (!attr_set) {
    UDP_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  out->fn = kern;
  out->grid = dim3(p.ntiles, p.CoutPad / (CP * 32));
  out->block = dim3(256);
  out->lds = (unsigned)lds;
  out->p = p;
  return UDP_OK;
}
template <int KS, int STRIDE, bool NCHW>
static int describe_ws_pb(const ConvParams& p, int pb, int cp, size_t lds, Launch* out) {
#define UDP_WS(B, C) \
  if (pb == B && cp == C) return describe_ws_one<KS, STRIDE, B, C, NCHW>(p, lds, out);
  UDP_WS(2, 1) UDP_WS(3, 1) UDP_WS(4, 1) UDP_WS(6, 1) UDP_WS(2, 2) UDP_WS(3, 2) UDP_WS(4, 2) UDP_WS(6, 2)
  UDP_WS(2, 4) UDP_WS(3, 4) UDP_WS(4, 4) UDP_WS(6, 4)
#undef UDP_WS
  return 1;
}
struct WsTile {
  int cp, pb, G, R, TW, wgs, sbuf;
  size_t lds;
};
// Tile of a (cout pairs per workgroup, pixel blocks per wave) candidate; false if it cannot be built.
static bool ws_tile(const ConvParams& p, int ks, int stride, int cp, int pb, WsTile* t, bool own_launch) {
  const int pg = 4 / cp;
  const int maxM = 16 * pb * pg;
  // column split: the full width, or 2..4 equal column tiles -- whichever fills the wave's pixel blocks best
  // (a 72-column map tiles exactly as 3 x 24 columns x 4 rows = 6 blocks, but only as 9 of 12 block slots at 36)
  bool have = false;
  double best_util = 0.0;
  for (int split = 1; split <= 4 || !have; ++split) {
    if (split > 8) break;
    const int TW = ceil_div(p.Wout, split);
    if (TW > 64 || TW > maxM || (have && TW < 8)) continue;
    int maxR = maxM / TW;
    if (maxR > p.Hout) maxR = p.Hout;
    int R = largest_divisor_leq(p.Hout, maxR);
    if (R * 2 <= maxR) R = maxR;
    int G = 1;
    if (R == p.Hout && TW == p.Wout) {
      G = maxM / (R * TW);
      if (G > p.N) G = p.N;
      if (G < 1) G = 1;
    }
    auto npix = [&](int g, int r) { return g * ((r - 1) * stride + ks) * ((TW - 1) * stride + ks); };
    while (npix(G, R) > MAXG * 64 && G > 1) --G;
    while (npix(G, R) > MAXG * 64 && R > 1) R = (R + 1) / 2;
    if (npix(G, R) > MAXG * 64) continue;
    int nstage = ceil_div(p.Cin, 32) > 1 ? 2 : 1;                               // stage buffers x (hi, lo) images
    size_t lds = (size_t)((npix(G, R) + 15) / 16) * 16 * ROWB * 2 * nstage;
    // a second workgroup per CU is worth more than the second stage buffer (conv_ws_body, `refill`; the kernels of convs
    // launched on their own have it: stride 2, and 3x3 stride 1 outside a launch group).  UDP_POSE_WS_SBUF: 0 off, 2 stride 2 only
    static const int sbuf_mode = getenv("UDP_POSE_WS_SBUF") == nullptr ? 1 : atoi(getenv("UDP_POSE_WS_SBUF"));
    const bool sbuf_kernel = stride == 2 || (ks == 3 && own_launch && sbuf_mode != 2);
    const bool sbuf = sbuf_kernel && nstage == 2 && lds > 80 * 1024 && lds / 2 <= 80 * 1024 && sbuf_mode != 0;
    if (sbuf) {
      nstage = 1;
      lds /= 2;
    }
    if (lds > 160 * 1024) continue;
    // pixel blocks per wave the tile really needs (the halo limit may have shrunk it): the smallest instantiated
    // count that covers them, so no wave idles under masked blocks
    const int need = ceil_div(ceil_div(G * R * TW, 16), pg);
    if (need > pb) continue;
    const int pbe = need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : 6;
    const int tiles = ceil_div(p.N, G) * ceil_div(p.Hout, R) * ceil_div(p.Wout, TW);
    // score = useful pixels / pixel slots over the whole layer (ragged last tiles and masked blocks are waste)
    //         x the share of the staged halo tile that is not halo (narrow tiles stage more of it)
    const double util = (double)p.N * p.Hout * p.Wout / ((double)tiles * 16 * pbe * pg) *
                        ((double)G * R * TW * stride * stride / (double)npix(G, R));
    if (!have || util > best_util + 0.05) {
      have = true;
      best_util = util;
      t->cp = cp;
      t->pb = pbe;
      t->G = G;
      t->R = R;
      t->TW = TW;
      t->lds = lds;
      t->sbuf = sbuf ? 1 : 0;
      t->wgs = tiles * (p.CoutPad / (cp * 32));
    }
  }
  return have;
}

static long g_ws_fill_wgs = 512;
void ws_set_fill_wgs(long wgs) { g_ws_fill_wgs = wgs; }

int describe_conv_ws(ConvParams p, int ks, int stride, Launch* out, bool grouped) {
  if ((stride != 1 && stride != 2) || (ks != 3 && ks != 1) || (p.out_nchw_f32 && stride != 1))
    return fail(UDP_ERR_UNSUPPORTED, "fragment-major weights (wfmt 1): conv ks=%d stride=%d nchw_out=%d has no weight-stationary kernel",
                ks, stride, p.out_nchw_f32);
  auto knob = [](const char* name, long dflt) {
    const char* v = getenv(name);
    return v ? atol(v) : dflt;
  };
  const int pairs = p.CoutPad / 32;
  const int force_cp = (int)knob("UDP_POSE_WS_CP", 0), force_pb = (int)knob("UDP_POSE_WS_PB", 0);
  // two workgroups on each of the 256 CUs; a member of a merged launch fills the chip with its siblings, and a conv of
  // a sub-batch lane (hrnet.hip: the other lane's kernels run beside it) is better off with the fatter tile from a
  // quarter of that on: W32 7.81 -> 7.98 k images/s at 128 (64: 7.88, 192: 7.83, same box)
  const long min_wgs = grouped ? 0 : knob("UDP_POSE_WS_MINWGS", g_ws_fill_wgs);
  // candidates from the fattest wave tile down: the first one that fills the chip wins, else the one with
  // the most workgroups
  WsTile best{}, t{};
  bool have = false;
  for (int pb : {6, 4, 3, 2}) {
    for (int cp : {4, 2, 1}) {
      if (pairs % cp || (force_cp && cp != force_cp) || (force_pb && pb != force_pb)) continue;
      if (grouped && stride == 1 && UDP_WS_MPB != 6 && pb != UDP_WS_MPB) continue;   // (diagnostic builds of the merged kernel)
      if (!ws_tile(p, ks, stride, cp, pb, &t, !grouped && !p.out_nchw_f32)) continue;   // (the NCHW-output kernels keep two stage buffers)
      const bool fills = t.wgs >= min_wgs, best_fills = have && best.wgs >= min_wgs;
      if (!have || (!best_fills && (fills || t.wgs > best.wgs))) {
        best = t;
        have = true;
      }
    }
  }
  if (!have) return fail(UDP_ERR_UNSUPPORTED, "weight-stationary conv: no tile for %dx%d C%d->%d", p.Hout, p.Wout, p.Cin, p.Cout);
  p.G = best.G;
  p.R = best.R;
  p.TW = best.TW;
  p.sbuf = best.sbuf;
  p.IH = (p.R - 1) * stride + ks;
  p.IW = (p.TW - 1) * stride + ks;
  p.tiles_x = ceil_div(p.Wout, p.TW);
  p.tiles_y = ceil_div(p.Hout, p.R);
  auto magic = [](int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); };
  p.mIW = magic(p.IW);
  p.mIH = magic(p.IH);
  p.mRT = magic(p.R * p.TW);
  p.mTW = magic(p.TW);
  p.ntiles = ceil_div(p.N, p.G) * p.tiles_y * p.tiles_x;
  if (getenv("UDP_POSE_DEBUG_TILES"))
    fprintf(stderr, "ws conv k%d s%d %dx%d C%d->%d: G=%d R=%d TW=%d CP=%d PB=%d lds=%zu%s wgs=%d\n", ks, stride, p.Hout, p.Wout, p.Cin,
            p.Cout, p.G, p.R, p.TW, best.cp, best.pb, best.lds, best.sbuf ? " (one stage buffer)" : "", best.wgs);
  int rc = 1;
  if (ks == 3 && stride == 1 && !p.out_nchw_f32) rc = describe_ws_pb<3, 1, false>(p, best.pb, best.cp, best.lds, out);
  if (ks == 3 && stride == 1 && p.out_nchw_f32) rc = describe_ws_pb<3, 1, true>(p, best.pb, best.cp, best.lds, out);   // the net's NCHW fp32 output (RSN head)
  if (ks == 3 && stride == 2) rc = describe_ws_pb<3, 2, false>(p, best.pb, best.cp, best.lds, out);
  if (ks == 1 && stride == 1 && !p.out_nchw_f32) rc = describe_ws_pb<1, 1, false>(p, best.pb, best.cp, best.lds, out);
  if (ks == 1 && stride == 1 && p.out_nchw_f32) rc = describe_ws_pb<1, 1, true>(p, best.pb, best.cp, best.lds, out);   // (HRNet final_layer)
  if (ks == 1 && stride == 2) rc = describe_ws_pb<1, 2, false>(p, best.pb, best.cp, best.lds, out);
  if (rc == 1) return fail(UDP_ERR_UNSUPPORTED, "weight-stationary conv: no kernel for PB=%d CP=%d", best.pb, best.cp);
  if (rc == UDP_OK && p.nout2 && !(ks == 3 && stride == 1))
    return fail(UDP_ERR_UNSUPPORTED, "second outputs: 3x3 stride-1 convs only");
  if (rc == UDP_OK && grouped && best.pb == UDP_WS_MPB && stride == 1 && !p.nout2) {
    out->groupable = 300 + ks * 10 + 6;        // storage/kernel family 3 = split fp16 weight-stationary
    out->ws_cp = best.cp;
  }
  return rc;
}

// Dispatch order of a merged weight-stationary launch.  Members arrive deepest-K first: long, matrix-bound
// workgroups (a 256-channel workgroup lives ~65 us, 72 K steps) down to the short HBM-bound 32-channel ones
// (~15 us).  Workgroups are dispatched in flat-index order, breadth first over the CUs.  Member after member
// (longest first) leaves the second half of the launch to the shallow member alone -- HBM-bound at ~4.9 TB/s while
// the matrix pipe idles, after a first half in which HBM idles (in-kernel timeline, tools/stamp_multi.py).  So the
// grid alternates chunks of the deep members (list A, longest first) with chunks of the shallowest member (list
// B), in about the proportion that exhausts both lists together: every CU holds both kinds for the whole launch, the
// deepest workgroups still all start in the first round, and the last workgroups to start are short ones.
// Chunks are multiples of 8 workgroups (one per XCD -- workgroup i runs on XCD i % 8, an uneven pattern would
// leave some XCDs with all the long workgroups).  UDP_POSE_WS_ORDER=lpt keeps the plain member-after-member order.
static void ws_order(ConvMulti* m, int n) {
  struct Seg {
    unsigned start, first, cnt;
    int mem;
  };
  unsigned cnt_of[4];
  unsigned total = 0;
  for (int j = 0; j < n; ++j) {
    cnt_of[j] = (j + 1 < n ? m->start[j + 1] : m->start[4]) - m->start[j];
    total += cnt_of[j];
  }
  const char* mode = getenv("UDP_POSE_WS_ORDER");
  const bool lpt = mode && strcmp(mode, "lpt") == 0;
  // list A runs 25 % ahead of its share, so the deep members are exhausted first and the launch ends on the short
  // workgroups of B alone (tail = one 15 us workgroup instead of a 20-25 us one): +2 % (UDP_POSE_WS_BIAS, percent)
  const unsigned bias = getenv("UDP_POSE_WS_BIAS") ? (unsigned)atoi(getenv("UDP_POSE_WS_BIAS")) : 125u;
  auto build = [&](unsigned g) {
    std::vector<Seg> segs;
    unsigned lo[4] = {0, 0, 0, 0};
    unsigned at = 0;
    auto take = [&](int j, unsigned cnt) {
      if (cnt > cnt_of[j] - lo[j]) cnt = cnt_of[j] - lo[j];
      if (!cnt) return;
      if (!segs.empty() && segs.back().mem == j && segs.back().first + segs.back().cnt == lo[j])
        segs.back().cnt += cnt;
      else
        segs.push_back({at, lo[j], cnt, j});
      lo[j] += cnt;
      at += cnt;
    };
    if (!lpt) {
      const unsigned nb = cnt_of[n - 1];
      const unsigned na = total - nb;
      unsigned ta = 0, tb = 0;                 // taken from A / B so far
      while (ta < na && tb < nb) {
        // the list that is behind its share goes next (ties: A, so that the deepest workgroups lead the grid)
        if ((unsigned long long)ta * nb * 100ull <= (unsigned long long)tb * na * bias) {
          int j = 0;
          while (lo[j] == cnt_of[j]) ++j;
          const unsigned c = cnt_of[j] - lo[j] < g ? cnt_of[j] - lo[j] : g;
          take(j, c);
          ta += c;
        } else {
          const unsigned c = nb - tb < g ? nb - tb : g;
          take(n - 1, c);
          tb += c;
        }
      }
    }
    for (int j = 0; j < n; ++j) take(j, cnt_of[j] - lo[j]);
    return segs;
  };
  // chunk size: 32 workgroups (UDP_POSE_WS_G); the per-8 lookup table (ConvMulti::tab) takes any number of chunks,
  // the segment table searched without it holds kMultiSegs
  unsigned g = getenv("UDP_POSE_WS_G") ? (unsigned)atoi(getenv("UDP_POSE_WS_G")) : 32u;
  if (g < 8 || g % 8) g = 32;
  std::vector<Seg> segs = build(g);
  m->tab_n = 0;
  bool ok = total % 8 == 0 && total / 8 <= (unsigned)kMultiTab && getenv("UDP_POSE_WS_NOTAB") == nullptr;
  for (int j = 0; j < n && ok; ++j)
    ok = m->tiles[j] % 8 == 0 && m->ncby[j] <= 64 && m->tiles[j] < (1u << 22) && (m->code[j] == 1 || m->code[j] == 2 || m->code[j] == 4);
  for (size_t k = 0; k < segs.size() && ok; ++k) ok = segs[k].start % 8 == 0 && segs[k].first % 8 == 0;
  if (ok) {
    for (const Seg& sg : segs) {
      const int j = sg.mem;
      for (unsigned b = sg.start; b < sg.start + sg.cnt; b += 8) {
        int tile, cby;
        ws_decode(b - sg.start + sg.first, m->tiles[j], m->ncby[j], tile, cby);
        const unsigned cl = m->code[j] == 1 ? 0u : m->code[j] == 2 ? 1u : 2u;      // log2 of the member's CP
        m->tab[b >> 3] = (unsigned)j | (cl << 2) | ((unsigned)cby << 4) | ((unsigned)tile << 10);
      }
    }
    m->tab_n = total / 8;
  }
  // the searched table (read by the kernel only when there is no lookup table): coarser chunks until it fits
  while (segs.size() > (size_t)kMultiSegs) {
    g += 8;
    segs = build(g);
  }
  for (int k = 0; k < kMultiSegs; ++k) {
    const bool have = (size_t)k < segs.size();
    m->seg_start[k] = have ? segs[k].start : 0xFFFFFFFFu;
    m->seg_first[k] = have ? segs[k].first : 0;
    m->seg_mem[k] = have ? segs[k].mem : 0;
  }
}

// Diagnostic (tests/test_host_cpu.py, no GPU needed): the dispatch tables ws_order builds for members of
// tiles[j] x ncby[j] workgroups, resolved for every flat block exactly as conv_ws_multi resolves them (table entry
// when there is one, segment search otherwise).  out_*[b] for b < total; returns total, or -1.
extern "C" int udp_debug_multi_order(const unsigned* tiles, const unsigned* ncby, const int* code, int n, unsigned cap,
                                     unsigned* out_member, unsigned* out_tile, unsigned* out_cby, int* used_table) {
  if (!tiles || !ncby || !code || n < 1 || n > 4 || !out_member || !out_tile || !out_cby) return -1;
  static ConvMulti m;
  memset(&m, 0, sizeof(m));
  unsigned total = 0;
  for (int j = 0; j < n; ++j) {
    m.start[j] = total;
    m.tiles[j] = tiles[j];
    m.ncby[j] = ncby[j];
    m.code[j] = code[j];
    total += tiles[j] * ncby[j];
  }
  for (int j = n; j < 5; ++j) m.start[j] = j < 4 ? 0xFFFFFFFFu : total;
  m.start[4] = total;
  for (int j = n; j < 4; ++j) m.tiles[j] = m.ncby[j] = 1;
  if (total > cap) return -1;
  ws_order(&m, n);
  if (used_table) *used_table = m.tab_n != 0;
  for (unsigned b = 0; b < total; ++b) {
    int j, tile, cby;
    if (m.tab_n) {
      const unsigned e = m.tab[b >> 3];
      j = (int)(e & 3u);
      if ((1 << ((e >> 2) & 3u)) != m.code[j]) return -1;
      cby = (int)((e >> 4) & 63u);
      tile = (int)((e >> 10) + (b & 7u));
    } else {
      int sg = 0;
      for (int k = 1; k < kMultiSegs; ++k) sg += b >= m.seg_start[k];
      j = m.seg_mem[sg];
      ws_decode(b - m.seg_start[sg] + m.seg_first[sg], m.tiles[j], m.ncby[j], tile, cby);
    }
    out_member[b] = (unsigned)j;
    out_tile[b] = (unsigned)tile;
    out_cby[b] = (unsigned)cby;
  }
  return (int)total;
}

// Kernel + attribute for a merged launch of `n` weight-stationary members; fills the kernel argument.
int describe_ws_multi(const Launch* members, int n, ConvMulti* m, Launch* out) {
  static bool ws_attr_set = false;
  const void* wk[2] = {reinterpret_cast<const void*>(&conv_ws_multi<3>), reinterpret_cast<const void*>(&conv_ws_multi<1>)};
  if (!ws_attr_set) {
    for (const void* k : wk) UDP_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ws_attr_set = true;
  }
  memset(m, 0, sizeof(*m));
  unsigned total = 0, lds = 0;
  for (int j = 0; j < n; ++j) {
    if (members[j].groupable != members[0].groupable) return fail(UDP_ERR_ARG, "describe_multi: mixed weight-stationary members");
    m->p[j] = members[j].p;
    m->start[j] = total;
    m->tiles[j] = members[j].grid.x;
    m->ncby[j] = members[j].grid.y;
    m->code[j] = members[j].ws_cp;
    total += members[j].grid.x * members[j].grid.y;
    if (members[j].lds > lds) lds = members[j].lds;
  }
  for (int j = n; j < 5; ++j) m->start[j] = j < 4 ? 0xFFFFFFFFu : total;
  for (int j = n; j < 4; ++j) m->tiles[j] = m->ncby[j] = 1;
  ws_order(m, n);
  out->fn = wk[members[0].groupable / 10 % 10 == 3 ? 0 : 1];
  out->grid = dim3(total);
  out->block = dim3(256);
  out->lds = lds;
  out->groupable = 0;
  return UDP_OK;
}

#ifdef UDP_STAMPS
int ws_set_stamps(unsigned long long* dev_buf) {
  UDP_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dev_buf, sizeof(dev_buf)));
  return UDP_OK;
}
// the weight-stationary kernels' stamps alone (libudp_pose_hip_stamps_ws.so links the product conv.o)
extern "C" int udp_debug_set_stamps_ws(unsigned long long* dev_buf) { return ws_set_stamps(dev_buf); }
#endif

// ---------------------------------------------------------------------------------------------------------------
// Two 1x1 convs chained through registers (udp_conv_op.chain_cout; the layer1 Bottlenecks, pose_hrnet.py:80-100: conv3
// + bn3 + shortcut + ReLU of one block, conv1 + bn1 + ReLU of the next):
//     Y = act(W . X + b [+ R])            Cin = 32 * NCH  ->  Cout = 32 * NP      (written to `out`)
//     Z = act2(W2 . Y_as_stored + b2)     Cout            ->  Cout2 = 64          (written to `chain`)
// Both run at the memory roofline on their own; chained, the 4 * planes-channel map Y is written once and not read back
// (1.41 -> 1.01 GB per pair at 128 images).  PIXEL-stationary: a wave owns 16 * PB pixels, keeps their X fragments in
// registers and walks the cout pairs of the first conv; the accumulators of pair p -- lane (kg, li) holds couts
// 32p + 8kg .. + 7 of pixel li, the fragment-major row order (udp_conv_op.wfmt) -- are, after the epilogue and the
// split into (hi, lo), exactly the B fragment of K chunk p of the second conv: no LDS, no barrier, the waves of a
// workgroup only share the weight fragments in L1 / L2.  Every accumulator sees the MFMAs of the separate convs in
// the same order (chunk by chunk: hi * Xhi, lo * Xhi, hi * 2^-11 * Xlo), so Y and Z are theirs bit for bit.
// Chain fields ride in ConvParams members the 1x1 conv does not use (no up-sampled addends here): wgt2 / bias2 = W2 /
// b2, up[0] = Z, up_shift[0..2] = Cout2, wexp2, act2.
// LDSW: the weights of both convs (and the biases) live in LDS -- 128 KiB + 1.25 KiB for Cin = 64, staged once by a
// persistent 8-wave workgroup per CU whose waves then walk the pixel groups: the vector-memory queue carries only what
// comes from / goes to HBM (with the fragments streamed from L2 they are 128 of the 224 vector-memory instructions of a
// wave, each 16 cycles of the CU's address unit, and every one of them queues behind the residual loads).
template <int NCH, int PB, int RD, bool LDSW>
__global__ __launch_bounds__(LDSW ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2))) void conv_chain_kernel(const ConvParams p) {
  using T = H2;
  constexpr int ESZ = 2, NB = 2, Q = 2;          // Q: cout pairs of the second conv (Cout2 = 64)
  constexpr int NWAVES = LDSW ? 8 : 4;
  constexpr int AD = LDSW ? 2 : 3;               // fragment ring: from L2 the step after next is in flight, from LDS the next one
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const unsigned npix = (unsigned)p.N * p.Hout * p.Wout;
  constexpr int NP = 8;                          // cout pairs of the first conv (Cout = 256) = K chunks of the second.  A constant:
  // the pairs are straight-line code -- a pair under `if (k < NP)` makes hipcc drain every load at the join (vmcnt(0)), and
  // with it every prefetch across pairs
  const int cout2 = p.up_shift[0];
  const unsigned inpb = (unsigned)p.in_pitch * ESZ * 2, outpb = (unsigned)p.out_pitch * ESZ * 2, respb = (unsigned)p.res_pitch * ESZ * 2;
  const unsigned zpb = (unsigned)cout2 * ESZ * 2;
  const unsigned in_lo = (unsigned)p.in_pitch * ESZ, out_lo = (unsigned)p.out_pitch * ESZ, res_lo = (unsigned)p.res_pitch * ESZ;
  const unsigned z_lo = (unsigned)cout2 * ESZ;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (p.sbuf & 8) ? 0 : npix * inpb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (p.sbuf & 1) ? 0 : (unsigned)NCH * NP * 4096u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt2), 0, (p.sbuf & 1) ? 0 : (unsigned)NP * Q * 4096u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, (unsigned)p.CoutPad * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias2), 0, (unsigned)cout2 * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, (p.res && !(p.sbuf & 4)) ? npix * respb : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.sbuf & 2) ? 0 : npix * outpb, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.up[0]), 0, (p.sbuf & 2) ? 0 : npix * zpb, 0x00020000);
  const float winv = __builtin_ldexpf(1.f, -p.wexp), winv2 = __builtin_ldexpf(1.f, -p.up_shift[1]);
  const unsigned wvoff = (unsigned)lane * 16u;

  constexpr unsigned kW1 = (unsigned)NCH * 8 * 4096u, kBias = kW1 + 8u * Q * 4096u;     // LDS offsets: second conv's weights, biases
  if constexpr (LDSW) {
    for (unsigned blk = wave; blk < kBias / 1024u; blk += NWAVES) {
      const bool first = blk * 1024u < kW1;
      blds16(first ? r_w : r_w2, (first ? blk * 1024u : blk * 1024u - kW1) + wvoff, smem + blk * 1024u);
    }
    float* sbias = reinterpret_cast<float*>(smem + kBias);
    if (threadIdx.x < 256 + 64) sbias[threadIdx.x] = threadIdx.x < 256 ? p.bias[threadIdx.x] : p.bias2[threadIdx.x - 256];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const unsigned ngroups = (npix + PB * 16 - 1) / (PB * 16);
  for (unsigned grp = (unsigned)blockIdx.x * NWAVES + wave; grp < ngroups; grp += LDSW ? gridDim.x * NWAVES : ngroups) {
  // (per-lane constants re-derived inside the loop: hoisted out, the offsets of all eight pairs live across it -- spills)
  unsigned wv = wvoff;
  int kgl = kg;
  if constexpr (LDSW) {
    asm volatile("" : "+v"(wv));
    asm volatile("" : "+v"(kgl));
  }
  unsigned pix[PB];
  bool ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    pix[i] = grp * (PB * 16) + i * 16 + li;
    ok[i] = pix[i] < npix;
  }
  // X fragments of the wave's pixels: lane (kg, li) holds channels 32c + 8kg .. + 7 of pixel li (hi and lo planes)
  f16x8 xh[PB][NCH], xl[PB][NCH];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const unsigned off = ok[i] ? pix[i] * inpb + (unsigned)(p.in_coff + 32 * c + 8 * kgl) * ESZ : kOobOff;
      xh[i][c] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_in, off, 0, 0));
      xl[i][c] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_in, off + in_lo, 0, 0));
    }
  // weight fragments: steps of one cout pair pr = NCH chunks of the first conv, then Q pairs of the second; the
  // fragments of the step after next stream in from L2 (ring of three, as in conv_ws_body)
  constexpr int S = NCH + Q;
  f16x8 ah[AD][NB], al[AD][NB];
  auto load_a = [&](int pr, int st, f16x8 (&h)[NB], f16x8 (&l)[NB]) __attribute__((always_inline)) {
    const bool first = st < NCH;
    const unsigned soff = first ? (unsigned)(st * NP + pr) * 4096u : (unsigned)(pr * Q + (st - NCH)) * 4096u;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if constexpr (LDSW) {
        const unsigned char* q = smem + (first ? 0u : kW1) + soff + wv + 2048u * nb;
        h[nb] = *reinterpret_cast<const f16x8*>(q);
        l[nb] = *reinterpret_cast<const f16x8*>(q + 1024);
      } else {
        h[nb] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(first ? r_w : r_w2, wv + 2048u * nb, soff, 0));
        l[nb] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(first ? r_w : r_w2, wv + 2048u * nb + 1024u, soff, 0));
      }
    }
  };
  // residual of the pair RD - 1 ahead in flight (vector-memory operations complete in issue order: the weight fragments
  // requested behind a residual load arrive with it, so the distance also bounds how long a fragment can be held up)
  ResH2<NB> rres[RD][PB];
  auto load_res = [&](int pr, ResH2<NB> (&r)[PB]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PB; ++i)
      load_res_h2(r[i], r_res, ok[i] ? pix[i] * respb + (unsigned)(p.res_coff + 32 * pr + 8 * kgl) * ESZ : kOobOff, res_lo);
  };
#pragma unroll
  for (int k = 0; k < RD - 1; ++k) load_res(k, rres[k]);
  load_a(0, 0, ah[0], al[0]);
  if constexpr (AD > 2) load_a(0, 1, ah[1], al[1]);

  f32x4 acc2[PB][Q][NB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc2[i][q][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mfma3 = [&](f32x4 (&acc)[NB], const f16x8 (&h)[NB], const f16x8 (&l)[NB], const f16x8 (&a2)[NB], const f16x8 bh, const f16x8 bl)
      __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[nb], bh, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l[nb], bh, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[nb], bl, acc[nb], 0, 0, 0);
      };

  // one cout pair of the first conv = one K chunk of the second; BUF0 = ring slot of its first step (S steps per pair)
  auto pair_body = [&](auto KC) __attribute__((always_inline)) {
    constexpr int pr = decltype(KC)::value, BUF0 = (pr * S) % AD, SIDE = pr % RD;
    if constexpr (pr + RD - 1 < NP) load_res(pr + RD - 1, rres[(SIDE + RD - 1) % RD]);
    f32x4 bias[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if constexpr (LDSW)
        bias[nb] = *reinterpret_cast<const f32x4*>(smem + kBias + (unsigned)(32 * pr + 8 * kgl + 4 * nb) * 4u);
      else
        bias[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bias, (unsigned)(32 * pr + 8 * kgl + 4 * nb) * 4u, 0, 0));
    }
    f32x4 acc[PB][NB];
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 yh[PB], yl[PB];
#pragma unroll
    for (int st = 0; st < S; ++st) {
      constexpr int dummy = 0;
      (void)dummy;
      const int buf = (BUF0 + st) % AD;
      {   // fragments of the step AD - 1 ahead (unconditional: past the end the last step's again)
        int npr = pr, nst = st + AD - 1;
        if (nst >= S) {
          nst -= S;
          npr = pr + 1 < NP ? pr + 1 : pr;
          if (pr + 1 >= NP) nst = S - 1;
        }
        load_a(npr, nst, ah[(buf + AD - 1) % AD], al[(buf + AD - 1) % AD]);
      }
      f16x8 a2[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) a2[nb] = ah[buf][nb] * (_Float16)0x1p-11f;
      if (st < NCH) {
#pragma unroll
        for (int i = 0; i < PB; ++i) mfma3(acc[i], ah[buf], al[buf], a2, xh[i][st], xl[i][st]);
        if (st == NCH - 1) {
          // epilogue of the first conv for this pair: conv_ws_body's, value by value
          const int cbase = 32 * pr + 8 * kgl;
#pragma unroll
          for (int i = 0; i < PB; ++i) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[i][nb][q] = __builtin_fmaf(acc[i][nb][q], winv, bias[nb][q]);
            add_res_h2(acc[i], rres[SIDE][i]);
            if (p.relu) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][nb][q] = acc[i][nb][q] > 0.f ? acc[i][nb][q] : 0.f;
            }
            h2_split8(acc[i][0], acc[i][1], yh[i], yl[i]);
            const unsigned ooff = ok[i] ? pix[i] * outpb + (unsigned)(p.out_coff + cbase) * ESZ : kOobOff;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yh[i]), r_out, ooff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yl[i]), r_out, ooff + out_lo, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < PB; ++i) mfma3(acc2[i][st - NCH], ah[buf], al[buf], a2, yh[i], yl[i]);
      }
    }
  };
  // S steps per pair over a ring of three: the ring slot of a pair's first step cycles with period 3 / gcd(S, 3)
  pair_body(std::integral_constant<int, 0>{});
  pair_body(std::integral_constant<int, 1>{});
  pair_body(std::integral_constant<int, 2>{});
  pair_body(std::integral_constant<int, 3>{});
  pair_body(std::integral_constant<int, 4>{});
  pair_body(std::integral_constant<int, 5>{});
  pair_body(std::integral_constant<int, 6>{});
  pair_body(std::integral_constant<int, 7>{});
  // epilogue of the second conv
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int cbase = 32 * q + 8 * kgl;
    f32x4 b2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if constexpr (LDSW)
        b2[nb] = *reinterpret_cast<const f32x4*>(smem + kBias + 1024u + (unsigned)(cbase + 4 * nb) * 4u);
      else
        b2[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bias2, (unsigned)(cbase + 4 * nb) * 4u, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      f32x4 v[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[nb][k] = __builtin_fmaf(acc2[i][q][nb][k], winv2, b2[nb][k]);
          if (p.up_shift[2]) v[nb][k] = v[nb][k] > 0.f ? v[nb][k] : 0.f;
        }
      store_vec_buf<T, NB>(r_z, ok[i] ? pix[i] * zpb + (unsigned)cbase * ESZ : kOobOff, z_lo, v);
    }
  }
  }   // pixel groups
}

int describe_conv_chain(ConvParams p, Launch* out) {
  const int cout2 = p.up_shift[0];
  if (p.Cin % 32 || (p.Cin != 64 && p.Cin != 128) || p.CoutPad != p.Cout || p.Cout != 256 || cout2 != 64 || p.nup ||
      p.nout2 || p.out_nchw_f32 || !p.out || !p.up[0] || !p.wgt2 || !p.bias2 || p.Hin != p.Hout || p.Win != p.Wout)
    return fail(UDP_ERR_UNSUPPORTED, "chained 1x1 convs: C%d -> %d -> %d (64 | 128 -> 256 -> 64, dense NHWC outputs)", p.Cin, p.Cout, cout2);
  // A/B (UDP_POSE_CHAIN_VAR): 0 = weights in LDS where they fit (Cin = 64), 1 = streamed from L2 everywhere
  static const int var = getenv("UDP_POSE_CHAIN_VAR") ? atoi(getenv("UDP_POSE_CHAIN_VAR")) : 0;
  constexpr int PB = 2;
  const long npix = (long)p.N * p.Hout * p.Wout;
  if (npix * (long)std::max(std::max(p.in_pitch, p.out_pitch), p.res_pitch) * 4 >= (1L << 31))
    return fail(UDP_ERR_UNSUPPORTED, "chained 1x1 convs: tensor beyond the 2 GiB buffer-descriptor range");
  const long groups = (npix + PB * 16 - 1) / (PB * 16);
  out->lds = 0;
  if (p.Cin == 64 && var == 0) {
    out->fn = reinterpret_cast<const void*>(&conv_chain_kernel<2, PB, 3, true>);
    out->lds = (2 * 8 + 8 * 2) * 4096 + 1280;
    out->grid = dim3((unsigned)std::min<long>(256, (groups + 7) / 8));
    out->block = dim3(512);
    UDP_HIP_CHECK(hipFuncSetAttribute(out->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)out->lds));
  } else {
    out->fn = p.Cin == 64 ? reinterpret_cast<const void*>(&conv_chain_kernel<2, PB, 3, false>) : reinterpret_cast<const void*>(&conv_chain_kernel<4, PB, 2, false>);
    out->grid = dim3((unsigned)((groups + 3) / 4));
    out->block = dim3(256);
  }
  out->groupable = 0;
  out->p = p;
  out->p.sbuf = getenv("UDP_POSE_CHAIN_DBG") ? atoi(getenv("UDP_POSE_CHAIN_DBG")) : 0;   // timing-only ablations (results WRONG): 1 no weights, 2 no stores, 4 no residual, 8 no X
  return UDP_OK;
}

int conv_ws_h2_overflow(hipStream_t s, int reset, int* flag) { return h2_overflow_fetch(s, reset, flag); }

}  // namespace udp
