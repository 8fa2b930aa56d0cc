// Shared host-side helpers of libudp_pose_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/udp_pose_hip.h"

namespace udp {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define UDP_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return udp::fail(UDP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                       __FILE__, __LINE__);                                                  \
  } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Split-fp16 ("f16x2") range guard.  fp16 tops out at 65504: a value of magnitude >= 65520 splits into hi = inf, and
// the NaN that grows out of it downstream does not survive a ReLU (max(NaN, 0) = 0), so a finiteness test of the
// final heat-maps can miss it.  Every kernel that WRITES split-fp16 values therefore raises this flag at the source
// (one per translation unit; udp_f16x2_overflow() ORs them).  m = the largest magnitude about to be split (NaN
// compares false, so `!(m < limit)` catches it too).
static __device__ int g_h2_overflow;
__device__ __forceinline__ void h2_range_check(float m) {
  if (!(m < 65520.f)) g_h2_overflow = 1;
}
// flag of the calling translation unit after `s` has drained; cleared when `reset`
static inline int h2_overflow_fetch(hipStream_t s, int reset, int* flag) {
  int v = 0;
  UDP_HIP_CHECK(hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_h2_overflow), sizeof(int), 0, hipMemcpyDeviceToHost, s));
  UDP_HIP_CHECK(hipStreamSynchronize(s));
  if (v && reset) {
    const int z = 0;
    UDP_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_h2_overflow), &z, sizeof(int), 0, hipMemcpyHostToDevice, s));
    UDP_HIP_CHECK(hipStreamSynchronize(s));
  }
  *flag |= v != 0;
  return UDP_OK;
}

// Parameters of one fused conv launch (device-visible, passed by value).
struct ConvParams {
  const void* in;
  void* out;
  const void* wgt;
  const float* bias;
  const void* wgt2;      // fused BasicBlock: second conv's weights / bias
  const float* bias2;
  const void* res;
  const void* up[3];
  int up_shift[3];
  int nup;
  int N, Hin, Win, Cin, Hout, Wout, Cout, CoutPad;
  int in_pitch, in_coff, out_pitch, out_coff, res_pitch, res_coff;   // channel-slice views (elements); pitch = channels per pixel of the stored tensor
  int G, R, TW;          // tile = G images x R rows x TW cols of output
  int IH, IW;            // input halo tile per image
  int tiles_x, tiles_y;  // tiles per image group
  unsigned mTX, mTY;     // ceil(2^32/d), 0 for d == 1 (tile index decode, t < 65536)
  int ntiles;            // all tiles of the launch (persistent workgroups stride over them)
  unsigned mIW, mIH, mRT, mTW;  // ceil(2^20/d): x/d == (x*m) >> 20 for x*d < 2^20 (24-bit multiply)
  int relu;
  int out_nchw_f32;      // epilogue writes NCHW fp32 (network output) instead of NHWC T
  int flip_from;         // stem only: images >= flip_from read image (n - flip_from) mirrored in x
  int sbuf;              // conv_mfma_kernel: one stage buffer instead of two (set by conv_choose_tile)
  int wfmt;              // udp_conv_op.wfmt: 1 = fragment-major split-fp16 weights (conv_ws_h2_kernel)
  int wexp;              // udp_conv_op.wexp: those weights are stored scaled by 2^wexp
  int in_stuff2;         // udp_conv_op.in_stuff2: conv_mfma_kernel reads the [Hin/2][Win/2] input as its zero-stuffed image
  double* bn_ws;         // training: per-workgroup BatchNorm partial sums of the output, [tile][2*Cout] (or null)
  // udp_conv_op.n_out2: second outputs out2[k] = out (as stored) + add2[k] of the weight-stationary split-fp16 convs
  void* out2[2];
  const void* add2[2];
  int nout2, out2_coff[2], out2_pitch[2], add2_coff[2], add2_pitch[2];
};

// Kernel argument of conv_mfma_multi: up to 4 independent convs in one launch (flat block index ->
// sub-problem j, tile, cout block).
constexpr int kMultiSegs = 64;
constexpr int kMultiTab = 480;
struct ConvMulti {
  ConvParams p[4];
  unsigned start[5];      // first flat block of sub-problem j; unused entries 0xFFFFFFFF, start[4] = total
  unsigned tiles[4];      // tiles (grid.x) of sub-problem j
  unsigned ncby[4];       // conv_ws_multi: cout blocks (grid.y) of sub-problem j
  int code[4];            // conv_ws_multi: cout pairs per workgroup (CP) of sub-problem j
  // conv_ws_multi dispatch order: the grid is a sequence of segments, segment s = workgroups seg_first[s] ..
  // of member seg_mem[s], placed at flat blocks seg_start[s] .. seg_start[s + 1] - 1 (unused: 0xFFFFFFFF)
  unsigned seg_start[kMultiSegs];
  unsigned seg_first[kMultiSegs];
  int seg_mem[kMultiSegs];
  // the same order resolved per run of 8 workgroups, when every boundary is a multiple of 8: tab[b >> 3] =
  // member | log2(CP) << 2 | cout block << 4 | first tile << 10, workgroup b takes tile (tab >> 10) + (b & 7).  ONE scalar load whose
  // address only needs blockIdx, instead of the table search + three dependent loads (1.6 us of every workgroup's
  // 15-60 us, tools/stamp_multi.py).  tab_n == 0: search the segment table.
  unsigned tab_n;
  unsigned tab[kMultiTab];
};

}  // namespace udp

namespace udp {
// One kernel launch, described once and then either enqueued on a stream or added to a hipGraph.
struct Launch {
  const void* fn = nullptr;
  dim3 grid, block;
  unsigned lds = 0;
  int groupable = 0;   // conv described by describe_conv_grouped: may be merged with its group siblings
  int ws_cp = 0;       // weight-stationary member: its cout pairs per workgroup
  ConvParams p;
};
// conv_ws.hip: the weight-stationary split-fp16 convs (tile choice + kernel of one conv; merged launch of 2..4)
int describe_conv_ws(ConvParams p, int ks, int stride, Launch* out, bool grouped = false);
int describe_ws_multi(const Launch* members, int n, ConvMulti* m, Launch* out);
int describe_conv_chain(ConvParams p, Launch* out);   // two 1x1 convs chained through registers (udp_conv_op.chain_cout)
void ws_set_fill_wgs(long wgs);                       // workgroups a conv launched on its own should reach (tile choice)
int ws_set_stamps(unsigned long long* dev_buf);      // diagnostic builds (-DUDP_STAMPS) only
int conv_h2_overflow(hipStream_t s, int reset, int* flag);       // conv.hip / conv_ws.hip / psa.hip: their g_h2_overflow
int conv_ws_h2_overflow(hipStream_t s, int reset, int* flag);
int psa_h2_overflow(hipStream_t s, int reset, int* flag);
}  // namespace udp
