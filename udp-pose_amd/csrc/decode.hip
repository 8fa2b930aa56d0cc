// UDP heat-map decode + flip-test fuse on gfx950.  HBM-bound: every map is read
// once (coalesced), everything else lives in LDS / registers.
//
// Replaces deep_hrnet/lib/core/inference.py: get_max_preds :30-58, post (DARK /
// Taylor) :60-145, the offset branch of get_final_preds :156-174, transform_preds
// :20-27, preds_in_input_space :177-179; and deep_hrnet/lib/utils/transforms.py
// flip_back :15-29 / flip_back_offset :31-47 with the average of
// deep_hrnet/lib/core/function.py:171.
//
// This file is compiled with -ffp-contract=off: the Gaussian blur follows the
// oracle's operation order (k[r]*x[c] + sum_d k[r+d]*(x[c-d]+x[c+d]), one fp32
// rounding per operation, rows then columns) so that the arg-max taken on a
// blurred map is bit-identical to the CPU oracle's.
#include "common.h"

namespace udp {

constexpr int kMaxTaps = 15;

struct Taps {
  float k[kMaxTaps];
  int ksize;
};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i;
}

// dst[y][x] = symmetric 1-D filter of src along x (axis=1) or y (axis=0).
__device__ void blur_pass(const float* __restrict__ src, float* __restrict__ dst, int h, int w,
                          const Taps& t, int axis) {
  const int r = (t.ksize - 1) / 2;
  for (int i = threadIdx.x; i < h * w; i += blockDim.x) {
    const int y = i / w, x = i - y * w;
    float acc = t.k[r] * src[i];
    for (int d = 1; d <= r; ++d) {
      float a, b;
      if (axis == 1) {
        a = src[y * w + reflect101(x - d, w)];
        b = src[y * w + reflect101(x + d, w)];
      } else {
        a = src[reflect101(y - d, h) * w + x];
        b = src[reflect101(y + d, h) * w + x];
      }
      const float pair = a + b;
      acc = acc + t.k[r + d] * pair;
    }
    dst[i] = acc;
  }
}

// Block-wide arg-max with first-index tie-break (np.argmax), plus min.
// Results are broadcast through red_* (shared).  NaNs never win a comparison.
struct MaxMin {
  float mx;
  int idx;
  float mn;
};

__device__ MaxMin block_argmax_min(const float* __restrict__ a, int count, float* red_v, int* red_i,
                                   float* red_m) {
  float bv = -INFINITY, bm = INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const float v = a[i];
    if (v > bv) {
      bv = v;
      bi = i;
    }
    bm = v < bm ? v : bm;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_down(bv, off);
    const int oi = __shfl_down(bi, off);
    const float om = __shfl_down(bm, off);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
    bm = om < bm ? om : bm;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
    red_v[wave] = bv;
    red_i[wave] = bi;
    red_m[wave] = bm;
  }
  __syncthreads();
  MaxMin r;
  r.mx = red_v[0];
  r.idx = red_i[0];
  r.mn = red_m[0];
  const int nw = blockDim.x >> 6;
  for (int k = 1; k < nw; ++k) {
    if (red_v[k] > r.mx || (red_v[k] == r.mx && red_i[k] < r.idx)) {
      r.mx = red_v[k];
      r.idx = red_i[k];
    }
    r.mn = red_m[k] < r.mn ? red_m[k] : r.mn;
  }
  if (r.idx == 0x7fffffff) r.idx = 0;  // all-NaN map: np.argmax would return the first NaN
  return r;
}

// transform_preds (inference.py:20-27) + preds_in_input_space (:177-179).
// f32_coords: the reference's coords array is float32 here (no DARK) -> its
// products stay fp32; otherwise fp64 coords times fp32-rounded scale factors.
__device__ void write_preds(double cx, double cy, bool f32_coords, const double* center,
                            const double* scale, int cs_is_f32, int h, int w, double* preds,
                            double* preds_in) {
  const double wm1 = (double)w - 1.0, hm1 = (double)h - 1.0;
  if (f32_coords) {
    const float fx = (float)cx, fy = (float)cy;
    preds_in[0] = (double)(fx / (float)wm1 * (float)(4.0 * w - 1.0));
    preds_in[1] = (double)(fy / (float)hm1 * (float)(4.0 * h - 1.0));
    if (cs_is_f32) {
      const float s0 = (float)scale[0] * 200.0f, s1 = (float)scale[1] * 200.0f;
      const float sx = s0 / (float)wm1, sy = s1 / (float)hm1;
      const float px = fx * sx + (float)center[0] - s0 * 0.5f;
      const float py = fy * sy + (float)center[1] - s1 * 0.5f;
      preds[0] = (double)px;
      preds[1] = (double)py;
    } else {
      const double s0 = scale[0] * 200.0, s1 = scale[1] * 200.0;
      const double sx = s0 / wm1, sy = s1 / hm1;
      // float32 coords * float64 scalar -> float64 in NumPy >= 2
      preds[0] = (double)(float)((double)fx * sx + center[0] - s0 * 0.5);
      preds[1] = (double)(float)((double)fy * sy + center[1] - s1 * 0.5);
    }
  } else {
    preds_in[0] = cx / wm1 * (4.0 * w - 1.0);
    preds_in[1] = cy / hm1 * (4.0 * h - 1.0);
    if (cs_is_f32) {
      const float s0 = (float)scale[0] * 200.0f, s1 = (float)scale[1] * 200.0f;
      const float sx = s0 / (float)wm1, sy = s1 / (float)hm1;
      const float h0 = s0 * 0.5f, h1 = s1 * 0.5f;
      preds[0] = cx * (double)sx + (double)(float)center[0] - (double)h0;
      preds[1] = cy * (double)sy + (double)(float)center[1] - (double)h1;
    } else {
      const double s0 = scale[0] * 200.0, s1 = scale[1] * 200.0;
      preds[0] = cx * (s0 / wm1) + center[0] - s0 * 0.5;
      preds[1] = cy * (s1 / hm1) + center[1] - s1 * 0.5;
    }
  }
}

// One workgroup per (image, joint) map.  LDS: raw map, row-pass, blurred map.
__global__ __launch_bounds__(256) void decode_gaussian_kernel(
    const float* __restrict__ hm, int j, int h, int w, const double* __restrict__ center,
    const double* __restrict__ scale, int cs_is_f32, int post, Taps t7, double* __restrict__ preds,
    float* __restrict__ maxvals, double* __restrict__ preds_in, int* __restrict__ argidx) {
  extern __shared__ float sm[];
  __shared__ float red_v[4], red_m[4];
  __shared__ int red_i[4];
  const int hw = h * w;
  float* raw = sm;
  float* tmp = sm + hw;
  float* blr = sm + 2 * hw;
  const int map = blockIdx.x;
  const int n = map / j;
  const float* src = hm + (size_t)map * hw;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) raw[i] = src[i];
  __syncthreads();
  const MaxMin mm = block_argmax_min(raw, hw, red_v, red_i, red_m);
  const float maxori = mm.mx;
  const bool keep = maxori > 0.0f;
  const int px = keep ? mm.idx % w : 0;
  const int py = keep ? mm.idx / w : 0;
  if (threadIdx.x == 0) {
    maxvals[map] = maxori;
    if (argidx) argidx[map] = mm.idx;
  }
  if (!post) {
    if (threadIdx.x == 0)
      write_preds((double)px, (double)py, true, center + 2 * n, scale + 2 * n, cs_is_f32, h, w,
                  preds + 2 * map, preds_in + 2 * map);
    return;
  }
  blur_pass(raw, tmp, h, w, t7, 1);
  __syncthreads();
  blur_pass(tmp, blr, h, w, t7, 0);
  __syncthreads();
  const MaxMin bm = block_argmax_min(blr, hw, red_v, red_i, red_m);
  if (threadIdx.x == 0) {
    const float mx = bm.mx, mn = bm.mn;
    const float range = mx - mn;
    // log(clip((b - mn) / (mx - mn) * maxori, 1e-3, 50)) at the replicate-padded 3x3 around the peak
    double s[3][3];
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        int yy = py + dy, xx = px + dx;
        yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
        const float rs = (blr[yy * w + xx] - mn) / range * maxori;
        // np.clip / np.log propagate NaN (flat map: 0/0 in the rescale); fminf/fmaxf would not.
        // fp32 log taken as the rounded fp64 log (correctly rounded, like NumPy's fp32 loop nearly always is)
        s[dy + 1][dx + 1] = (rs == rs) ? (double)(float)log((double)fminf(fmaxf(rs, 0.001f), 50.0f)) : (double)NAN;
      }
    const double I = s[1][1], Ix1 = s[1][2], Ix1_ = s[1][0], Iy1 = s[2][1], Iy1_ = s[0][1];
    const double Ix1y1 = s[2][2], Ix1_y1_ = s[0][0];
    const double dx = 0.5 * (Ix1 - Ix1_), dy = 0.5 * (Iy1 - Iy1_);
    const double dxx = Ix1 - 2 * I + Ix1_, dyy = Iy1 - 2 * I + Iy1_;
    const double dxy = 0.5 * (Ix1y1 - Ix1 - Iy1 + I + I - Ix1_ - Iy1_ + Ix1_y1_);
    const double det = dxx * dyy - dxy * dxy;
    double sx = 0.0, sy = 0.0;
    if (det != 0.0) {
      const double inv = 1.0 / det;
      sx = inv * (dyy * dx - dxy * dy);
      sy = inv * (dxx * dy - dxy * dx);
    }
    const double cx = (double)px - sx, cy = (double)py - sy;
    write_preds(cx, cy, false, center + 2 * n, scale + 2 * n, cs_is_f32, h, w, preds + 2 * map,
                preds_in + 2 * map);
  }
}

// Offset head: channels (3j, 3j+1, 3j+2) = (heat-map, x-offset, y-offset).
__global__ __launch_bounds__(256) void decode_offset_kernel(
    const float* __restrict__ out, int j, int h, int w, const double* __restrict__ center,
    const double* __restrict__ scale, int cs_is_f32, float kpd, Taps t15, Taps t7,
    double* __restrict__ preds, float* __restrict__ maxvals, double* __restrict__ preds_in,
    int* __restrict__ argidx) {
  extern __shared__ float sm[];
  __shared__ float red_v[4], red_m[4];
  __shared__ int red_i[4];
  __shared__ float off_s[2];
  const int hw = h * w;
  float* raw = sm;
  float* tmp = sm + hw;
  float* blr = sm + 2 * hw;
  const int map = blockIdx.x;
  const int n = map / j;
  const float* src = out + (size_t)map * 3 * hw;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) raw[i] = src[i];
  __syncthreads();
  blur_pass(raw, tmp, h, w, t15, 1);
  __syncthreads();
  blur_pass(tmp, blr, h, w, t15, 0);
  __syncthreads();
  const MaxMin mm = block_argmax_min(blr, hw, red_v, red_i, red_m);
  const bool keep = mm.mx > 0.0f;
  const int px = keep ? mm.idx % w : 0;
  const int py = keep ? mm.idx / w : 0;
  // 7x7 blur of kpd*offset at (py,px) only: 7 row-pass values, then the column pass.
  if (threadIdx.x < 2) {
    const float* o = src + (size_t)(1 + threadIdx.x) * hw;
    float rowv[7];
    for (int dy = -3; dy <= 3; ++dy) {
      const int yy = reflect101(py + dy, h);
      float acc = t7.k[3] * (o[yy * w + px] * kpd);
      for (int d = 1; d <= 3; ++d) {
        const float a = o[yy * w + reflect101(px - d, w)] * kpd;
        const float b = o[yy * w + reflect101(px + d, w)] * kpd;
        const float pair = a + b;
        acc = acc + t7.k[3 + d] * pair;
      }
      rowv[dy + 3] = acc;
    }
    float acc = t7.k[3] * rowv[3];
    for (int d = 1; d <= 3; ++d) {
      const float pair = rowv[3 - d] + rowv[3 + d];
      acc = acc + t7.k[3 + d] * pair;
    }
    off_s[threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    maxvals[map] = mm.mx;
    if (argidx) argidx[map] = mm.idx;
    const float cx = (float)px + off_s[0];
    const float cy = (float)py + off_s[1];
    write_preds((double)cx, (double)cy, true, center + 2 * n, scale + 2 * n, cs_is_f32, h, w,
                preds + 2 * map, preds_in + 2 * map);
  }
}

__global__ __launch_bounds__(256) void flip_fuse_kernel(const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        const int* __restrict__ src_ch,
                                                        const float* __restrict__ sign, int c, int h,
                                                        int w, long total, float divisor,
                                                        float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = i % w;
    long t = i / w;
    const int y = t % h;
    t /= h;
    const int ch = t % c;
    const long n = t / c;
    const float fb = b[((n * c + src_ch[ch]) * h + y) * w + (w - 1 - x)] * sign[ch];
    out[i] = (a[i] + fb) * 0.5f / divisor;   // divisor 1 (exact) or RSN's 255 (test.py:185)
  }
}

}  // namespace udp

using namespace udp;

extern "C" int udp_gaussian_taps_host(int ksize, float* taps) {
  if (!taps || ksize < 1 || ksize > kMaxTaps || (ksize & 1) == 0)
    return fail(UDP_ERR_ARG, "udp_gaussian_taps_host: ksize=%d (odd, 1..%d)", ksize, kMaxTaps);
  static const double t1[] = {1.0}, t3[] = {0.25, 0.5, 0.25}, t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                      t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
  const double* fixed = ksize == 1 ? t1 : ksize == 3 ? t3 : ksize == 5 ? t5 : ksize == 7 ? t7 : nullptr;
  const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1.0) + 0.8;
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0.0;
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    const double t = fixed ? fixed[i] : exp(scale2x * x * x);
    taps[i] = (float)t;
    sum += (double)taps[i];
  }
  const double inv = 1.0 / sum;
  for (int i = 0; i < ksize; ++i) taps[i] = (float)((double)taps[i] * inv);
  return UDP_OK;
}

static int make_taps(int ksize, Taps* t) {
  memset(t, 0, sizeof(*t));
  t->ksize = ksize;
  return udp_gaussian_taps_host(ksize, t->k);
}

static int check_decode_args(const char* who, const void* hm, int n, int j, int h, int w, const void* c,
                             const void* s, const void* preds, const void* maxvals, const void* pin) {
  if (!hm || !c || !s || !preds || !maxvals || !pin) return fail(UDP_ERR_ARG, "%s: null pointer", who);
  if (n < 0 || j <= 0 || h <= 0 || w <= 0) return fail(UDP_ERR_ARG, "%s: bad shape n=%d j=%d h=%d w=%d", who, n, j, h, w);
  if ((size_t)h * w * 3 * sizeof(float) > 150 * 1024)
    return fail(UDP_ERR_UNSUPPORTED, "%s: map %dx%d does not fit the LDS working set", who, h, w);
  return UDP_OK;
}

extern "C" int udp_decode_gaussian(const float* heatmaps, int n, int j, int h, int w, const double* center,
                                   const double* scale, int cs_is_f32, int post_process, double* preds,
                                   float* maxvals, double* preds_in, int32_t* argidx, void* stream) {
  int rc = check_decode_args("udp_decode_gaussian", heatmaps, n, j, h, w, center, scale, preds, maxvals, preds_in);
  if (rc) return rc;
  if (n == 0) return UDP_OK;
  Taps t7;
  make_taps(7, &t7);
  const size_t lds = (size_t)h * w * 3 * sizeof(float);
  static bool attr = false;
  if (!attr) {
    UDP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_gaussian_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(decode_gaussian_kernel, dim3(n * j), dim3(256), lds, (hipStream_t)stream, heatmaps, j, h,
                     w, center, scale, cs_is_f32, post_process, t7, preds, maxvals, preds_in, argidx);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

extern "C" int udp_decode_offset(const float* heatmaps, int n, int j, int h, int w, const double* center,
                                 const double* scale, int cs_is_f32, float kpd, double* preds, float* maxvals,
                                 double* preds_in, int32_t* argidx, void* stream) {
  int rc = check_decode_args("udp_decode_offset", heatmaps, n, j, h, w, center, scale, preds, maxvals, preds_in);
  if (rc) return rc;
  if (n == 0) return UDP_OK;
  Taps t15, t7;
  make_taps(15, &t15);
  make_taps(7, &t7);
  const size_t lds = (size_t)h * w * 3 * sizeof(float);
  static bool attr = false;
  if (!attr) {
    UDP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_offset_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(decode_offset_kernel, dim3(n * j), dim3(256), lds, (hipStream_t)stream, heatmaps, j, h, w,
                     center, scale, cs_is_f32, kpd, t15, t7, preds, maxvals, preds_in, argidx);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

extern "C" int udp_flip_fuse_scaled(const float* a, const float* b, const int32_t* src_ch, const float* sign, int n,
                                    int c, int h, int w, float divisor, float* out, void* stream) {
  if (!a || !b || !src_ch || !sign || !out) return fail(UDP_ERR_ARG, "udp_flip_fuse: null pointer");
  if (n < 0 || c <= 0 || h <= 0 || w <= 0 || !(divisor > 0.f)) return fail(UDP_ERR_ARG, "udp_flip_fuse: bad shape / divisor");
  const long total = (long)n * c * h * w;
  if (total == 0) return UDP_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(flip_fuse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, src_ch,
                     sign, c, h, w, total, divisor, out);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

extern "C" int udp_flip_fuse(const float* a, const float* b, const int32_t* src_ch, const float* sign, int n,
                             int c, int h, int w, float* out, void* stream) {
  return udp_flip_fuse_scaled(a, b, src_ch, sign, n, c, h, w, 1.0f, out, stream);
}
