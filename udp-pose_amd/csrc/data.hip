// UDP data path + loss kernels for gfx950 (all HBM-bound, one pass over the data).
//
//   udp_warp_affine     cv2.warpAffine(INTER_LINEAR, border 0) + ToTensor + Normalize
//                       (deep_hrnet/pose_engine.py:76-81,40-43;
//                        deep_hrnet/lib/dataset/JointsDataset.py:226-227)
//   udp_target_gaussian JointsDataset.generate_target, gaussian  (:301-348)
//   udp_target_offset   JointsDataset.generate_target, offset    (:349-381)
//   udp_mse_loss        JointsMSELoss / JointsMSELoss_offset fwd+grad
//                       (deep_hrnet/lib/core/loss.py:15-39 / :41-76)
// Compiled with -ffp-contract=off (same reason as decode.hip).
#include "common.h"

namespace udp {

// OpenCV's 8-bit bilinear warp: coordinates in 1/32 pixel via 10-bit fixed point,
// 15-bit tap weights (32-fx)*(32-fy)*32, value = (sum + 2^14) >> 15.
__global__ __launch_bounds__(256) void warp_affine_kernel(const uint8_t* __restrict__ frame, int fh, int fw,
                                                          int row_stride, const double* __restrict__ mats,
                                                          int n, int oh, int ow, float m0, float m1, float m2,
                                                          float s0, float s1, float s2, int flip_lr, int swap_rb,
                                                          float* __restrict__ out) {
  const long total = (long)n * oh * ow;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = i % ow;
    const long t = i / ow;
    const int y = t % oh;
    const int b = t / oh;
    const double* m = mats + 6 * b;
    const long adelta = (long)rint(m[0] * (double)x * 1024.0);
    const long bdelta = (long)rint(m[3] * (double)x * 1024.0);
    const long X0 = (long)rint((m[1] * (double)y + m[2]) * 1024.0) + 16;
    const long Y0 = (long)rint((m[4] * (double)y + m[5]) * 1024.0) + 16;
    const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const long sx = X >> 5, sy = Y >> 5;
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
    const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0ok = sx >= 0 && sx < fw, x1ok = sx + 1 >= 0 && sx + 1 < fw;
    const bool y0ok = sy >= 0 && sy < fh, y1ok = sy + 1 >= 0 && sy + 1 < fh;
    const uint8_t* r0 = frame + (y0ok ? sy : 0) * (long)row_stride;
    const uint8_t* r1 = frame + (y1ok ? sy + 1 : 0) * (long)row_stride;
    // flip_lr: the source is the frame mirrored left-right (data_numpy[:, ::-1, :], JointsDataset.py:219):
    // same fixed-point taps, columns looked up mirrored
    const long c0 = (x0ok ? (flip_lr ? fw - 1 - sx : sx) : 0) * 3, c1 = (x1ok ? (flip_lr ? fw - 2 - sx : sx + 1) : 0) * 3;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = swap_rb ? 2 - c : c;      // cv2.COLOR_BGR2RGB (:195-196)
      const int p00 = (y0ok && x0ok) ? r0[c0 + sc] : 0, p01 = (y0ok && x1ok) ? r0[c1 + sc] : 0;
      const int p10 = (y1ok && x0ok) ? r1[c0 + sc] : 0, p11 = (y1ok && x1ok) ? r1[c1 + sc] : 0;
      int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + 16384) >> 15;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      const float f = (float)v / 255.0f;
      out[(((long)b * 3 + c) * oh + y) * ow + x] = (f - mean[c]) / stdv[c];
    }
  }
}

// AID information dropping on the normalized crop (lib/utils/transforms.py:144-224): a dropped pixel is
// uint8 0 before ToTensor/Normalize, i.e. (0 - mean)/std here.
//   Cutout: ((cx - x)/rx)^2 + ((cy - y)/ry)^2 <= 1 in fp64, `P` ellipses per image (rx <= 0: unused slot)
//   HideAndSeek: grid g; the reference indexes img[x:x_end, y:y_end] with x stepping over the WIDTH and y
//   over the HEIGHT (:172-177) -- rows are cut by the x cells, columns by the y cells; kept as is.
__global__ __launch_bounds__(256) void aid_apply_kernel(float* __restrict__ img, int n, int h, int w,
                                                        const double* __restrict__ cutout, int P,
                                                        const int32_t* __restrict__ hs_grid,
                                                        const uint8_t* __restrict__ hs_mask, int mx, int my,
                                                        float d0, float d1, float d2) {
  const long total = (long)n * h * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = i % w;
    const long t = i / w;
    const int y = t % h;
    const int b = t / h;
    bool drop = false;
    for (int k = 0; k < P && !drop; ++k) {
      const double* e = cutout + ((long)b * P + k) * 4;
      if (e[2] > 0.0) {
        const double xo = (e[0] - (double)x) / e[2], yo = (e[1] - (double)y) / e[3];
        drop = xo * xo + yo * yo <= 1.0;
      }
    }
    const int g = hs_grid ? hs_grid[b] : 0;
    if (!drop && g > 0 && y < w && x < h) drop = hs_mask[((long)b * mx + y / g) * my + x / g] != 0;
    if (drop) {
      float* o = img + (long)b * 3 * h * w + (long)y * w + x;
      o[0] = d0;
      o[(long)h * w] = d1;
      o[2L * h * w] = d2;
    }
  }
}

__global__ __launch_bounds__(256) void target_gaussian_kernel(const float* __restrict__ joints,
                                                              const float* __restrict__ vis, int n, int j,
                                                              int img_w, int img_h, int W, int H, float sigma,
                                                              float* __restrict__ target,
                                                              float* __restrict__ weight) {
  const long total = (long)n * j * H * W;
  const double stx = ((double)img_w - 1.0) / ((double)W - 1.0);
  const double sty = ((double)img_h - 1.0) / ((double)H - 1.0);
  const int tmp = (int)(sigma * 3.0f);
  const int size = 2 * tmp + 1;
  const double two_s2 = 2.0 * (double)sigma * (double)sigma;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = i % W;
    long t = i / W;
    const int y = t % H;
    t /= H;  // joint map index n*j + jj
    const double jx = (double)joints[2 * t], jy = (double)joints[2 * t + 1];
    const double ax = jx / stx, ay = jy / sty;
    const int mu_x = (int)(ax + 0.5), mu_y = (int)(ay + 0.5);
    const int ulx = mu_x - tmp, uly = mu_y - tmp;
    const int brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
    const bool outside = ulx >= W || uly >= H || brx < 0 || bry < 0;
    const float v = vis[t];
    const float tw = outside ? 0.0f : v;
    if (x == 0 && y == 0) weight[t] = tw;
    float val = 0.0f;
    const int gx = x - ulx, gy = y - uly;
    if (!outside && tw > 0.5f && gx >= 0 && gx < size && gy >= 0 && gy < size) {
      const double x0 = (double)(size / 2) + (ax - (double)mu_x);
      const double y0 = (double)(size / 2) + (ay - (double)mu_y);
      const double dx = (double)gx - x0, dy = (double)gy - y0;
      val = (float)exp(-(dx * dx + dy * dy) / two_s2);
    }
    target[i] = val;
  }
}

__global__ __launch_bounds__(256) void target_offset_kernel(const float* __restrict__ joints,
                                                            const float* __restrict__ vis, int n, int j,
                                                            int img_w, int img_h, int W, int H, float kpd,
                                                            float* __restrict__ target,
                                                            float* __restrict__ weight) {
  const long total = (long)n * j * H * W;
  const double stx = ((double)img_w - 1.0) / ((double)W - 1.0);
  const double sty = ((double)img_h - 1.0) / ((double)H - 1.0);
  const long hw = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = i % W;
    long t = i / W;
    const int y = t % H;
    t /= H;
    const double xo = ((double)joints[2 * t] / stx - (double)x) / (double)kpd;
    const double yo = ((double)joints[2 * t + 1] / sty - (double)y) / (double)kpd;
    const double dis = xo * xo + yo * yo;
    const float v = vis[t];
    if (x == 0 && y == 0) weight[t] = v;
    const bool keep = dis <= 1.0 && dis >= 0.0 && v > 0.5f;
    float* o = target + t * 3 * hw + (long)y * W + x;
    o[0] = keep ? 1.0f : 0.0f;
    o[hw] = keep ? (float)xo : 0.0f;
    o[2 * hw] = keep ? (float)yo : 0.0f;
  }
}

// loss[0] += 0.5/norm * sum (w (p-g))^2 on heat-map channels, loss[1] likewise on the
// offset channels weighted by the ground-truth disk mask; grad = d loss / d pred.
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ pred,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ weight, int b, int j, int hw,
                                                       int is_offset, double* __restrict__ loss,
                                                       float* __restrict__ grad) {
  __shared__ double red[2][4];
  const int k = is_offset ? 3 : 1;
  const long total = (long)b * j * k * hw;
  const double norm = (double)j * (double)b * (double)hw;
  double acc_h = 0.0, acc_o = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int pidx = i % hw;
    const long ch = i / hw;          // (b*j + jj)*k + kk
    const int kk = ch % k;
    const long bj = ch / k;
    const double p = (double)pred[i], g = (double)target[i];
    double d, wgt;
    if (kk == 0) {
      wgt = (double)weight[bj];
      d = (p - g) * wgt;
      acc_h += d * d;
    } else {
      wgt = (double)target[(bj * k) * hw + pidx];
      d = (p - g) * wgt;
      acc_o += d * d;
    }
    if (grad) grad[i] = (float)(d * wgt / norm);
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc_h += __shfl_down(acc_h, off);
    acc_o += __shfl_down(acc_o, off);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[0][wave] = acc_h;
    red[1][wave] = acc_o;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double h = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double o = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    atomicAdd(&loss[0], 0.5 * h / norm);
    if (is_offset) atomicAdd(&loss[1], 0.5 * o / norm);
  }
}

static unsigned grid_for(long total) {
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace udp

using namespace udp;

extern "C" int udp_warp_affine_ex(const uint8_t* frame, int fh, int fw, int row_stride_bytes, const double* mats,
                                  int n, int oh, int ow, const float* mean3, const float* std3, int flip_lr,
                                  int swap_rb, float* out, void* stream) {
  if (!frame || !mats || !mean3 || !std3 || !out) return fail(UDP_ERR_ARG, "udp_warp_affine: null pointer");
  if (fh <= 0 || fw <= 0 || row_stride_bytes < fw * 3 || n < 0 || oh <= 0 || ow <= 0)
    return fail(UDP_ERR_ARG, "udp_warp_affine: bad shape");
  if (n == 0) return UDP_OK;
  hipLaunchKernelGGL(warp_affine_kernel, dim3(grid_for((long)n * oh * ow)), dim3(256), 0, (hipStream_t)stream,
                     frame, fh, fw, row_stride_bytes, mats, n, oh, ow, mean3[0], mean3[1], mean3[2], std3[0],
                     std3[1], std3[2], flip_lr ? 1 : 0, swap_rb ? 1 : 0, out);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

extern "C" int udp_warp_affine(const uint8_t* frame, int fh, int fw, int row_stride_bytes, const double* mats,
                               int n, int oh, int ow, const float* mean3, const float* std3, float* out,
                               void* stream) {
  return udp_warp_affine_ex(frame, fh, fw, row_stride_bytes, mats, n, oh, ow, mean3, std3, 0, 0, out, stream);
}

extern "C" int udp_aid_apply(float* img, int n, int h, int w, const double* cutout, int n_patch, const int32_t* hs_grid,
                             const uint8_t* hs_mask, int mask_x, int mask_y, const float* mean3, const float* std3,
                             void* stream) {
  if (!img || !mean3 || !std3) return fail(UDP_ERR_ARG, "udp_aid_apply: null pointer");
  if (n < 0 || h <= 0 || w <= 0 || n_patch < 0 || (n_patch > 0 && !cutout)) return fail(UDP_ERR_ARG, "udp_aid_apply: bad shape");
  if (hs_grid && (!hs_mask || mask_x <= 0 || mask_y <= 0)) return fail(UDP_ERR_ARG, "udp_aid_apply: hide-and-seek mask missing");
  if (n == 0) return UDP_OK;
  hipLaunchKernelGGL(aid_apply_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, img, n, h, w,
                     cutout, n_patch, hs_grid, hs_mask, mask_x, mask_y, (0.f - mean3[0]) / std3[0],
                     (0.f - mean3[1]) / std3[1], (0.f - mean3[2]) / std3[2]);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

static int check_target_args(const char* who, const void* a, const void* b, const void* c, const void* d, int n,
                             int j, int img_w, int img_h, int hm_w, int hm_h) {
  if (!a || !b || !c || !d) return fail(UDP_ERR_ARG, "%s: null pointer", who);
  if (n < 0 || j <= 0 || img_w < 2 || img_h < 2 || hm_w < 2 || hm_h < 2) return fail(UDP_ERR_ARG, "%s: bad shape", who);
  return UDP_OK;
}

extern "C" int udp_target_gaussian(const float* joints, const float* vis, int n, int j, int img_w, int img_h,
                                   int hm_w, int hm_h, float sigma, float* target, float* weight, void* stream) {
  int rc = check_target_args("udp_target_gaussian", joints, vis, target, weight, n, j, img_w, img_h, hm_w, hm_h);
  if (rc) return rc;
  if (n == 0) return UDP_OK;
  hipLaunchKernelGGL(target_gaussian_kernel, dim3(grid_for((long)n * j * hm_h * hm_w)), dim3(256), 0,
                     (hipStream_t)stream, joints, vis, n, j, img_w, img_h, hm_w, hm_h, sigma, target, weight);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

extern "C" int udp_target_offset(const float* joints, const float* vis, int n, int j, int img_w, int img_h,
                                 int hm_w, int hm_h, float kpd, float* target, float* weight, void* stream) {
  int rc = check_target_args("udp_target_offset", joints, vis, target, weight, n, j, img_w, img_h, hm_w, hm_h);
  if (rc) return rc;
  if (n == 0) return UDP_OK;
  hipLaunchKernelGGL(target_offset_kernel, dim3(grid_for((long)n * j * hm_h * hm_w)), dim3(256), 0,
                     (hipStream_t)stream, joints, vis, n, j, img_w, img_h, hm_w, hm_h, kpd, target, weight);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}

extern "C" int udp_mse_loss(const float* pred, const float* target, const float* weight, int b, int j, int hw,
                            int is_offset, double* loss_out, float* grad, void* stream) {
  if (!pred || !target || !weight || !loss_out) return fail(UDP_ERR_ARG, "udp_mse_loss: null pointer");
  if (b <= 0 || j <= 0 || hw <= 0) return fail(UDP_ERR_ARG, "udp_mse_loss: bad shape");
  hipStream_t s = (hipStream_t)stream;
  UDP_HIP_CHECK(hipMemsetAsync(loss_out, 0, 2 * sizeof(double), s));
  const long total = (long)b * j * (is_offset ? 3 : 1) * hw;
  long blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mse_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, target, weight, b, j, hw,
                     is_offset ? 1 : 0, loss_out, grad);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}
