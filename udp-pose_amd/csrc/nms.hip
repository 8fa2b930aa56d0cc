// Keypoint rescoring + OKS-NMS for gfx950: the step after decode in evaluation.
// Replaces the per-image loop of deep_hrnet/lib/dataset/coco.py:321-356 (rescoring :326-341,
// oks_nms call :348-351) and deep_hrnet/lib/nms/nms.py:75-124 (oks_iou, oks_nms).
//
// One workgroup per image (persons of an image are contiguous, img_offsets[i]..img_offsets[i+1]):
// rescoring, rank by score (descending; ties: higher index first, the order of argsort()[::-1] on a
// stable sort), then the greedy sweep with all threads evaluating the OKS of the kept pose against the
// remaining ones.  dtypes follow the reference's NumPy arithmetic: keypoints fp32, squared distances
// summed in fp32, everything after the division by the variances in fp64.
// Compiled with -ffp-contract=off.
#include "common.h"

namespace udp {

constexpr int kMaxPersons = 1024;
constexpr int kMaxJoints = 64;

__global__ __launch_bounds__(256) void oks_nms_kernel(const float* __restrict__ kpts, const double* __restrict__ areas,
                                                      const double* __restrict__ box_scores,
                                                      const int32_t* __restrict__ offs, int J,
                                                      const double* __restrict__ vars, double in_vis_thre, int rescore,
                                                      double oks_thre, int use_vis, double oks_vis_thre, int soft,
                                                      double* __restrict__ scores_out, int32_t* __restrict__ keep_rank) {
  __shared__ double sc[kMaxPersons];
  __shared__ int order[kMaxPersons];
  __shared__ unsigned char dead[kMaxPersons];
  __shared__ float gk[kMaxJoints * 3];
  __shared__ double svar[kMaxJoints];
  const int a = offs[blockIdx.x], P = offs[blockIdx.x + 1] - a, t = threadIdx.x;
  for (int j = t; j < J; j += 256) svar[j] = vars[j];
  // rescoring (coco.py:326-341): mean of joint scores above in_vis_thre, accumulated in fp32, times the box score
  for (int p = t; p < P; p += 256) {
    double s = box_scores[a + p];
    if (rescore) {
      const float* k = kpts + (long)(a + p) * J * 3;
      float acc = 0.f;
      int valid = 0;
      for (int j = 0; j < J; ++j) {
        const float ts = k[3 * j + 2];
        if ((double)ts > in_vis_thre) {
          acc = acc + ts;
          ++valid;
        }
      }
      if (valid) acc = acc / (float)valid;
      s = (double)acc * s;
    }
    sc[p] = s;
    scores_out[a + p] = s;
    keep_rank[a + p] = -1;
    dead[p] = 0;
  }
  __syncthreads();
  for (int p = t; p < P; p += 256) {
    // rank = position in scores.argsort()[::-1] (nms.py:95).  NumPy sorts NaN last, so reversed it comes
    // FIRST: rank with NaN = +inf -- a total order, every order[] slot is written exactly once
    int r = 0;
    const double s = sc[p] != sc[p] ? (double)INFINITY : sc[p];
    for (int q = 0; q < P; ++q) {
      const double sq = sc[q] != sc[q] ? (double)INFINITY : sc[q];
      r += (sq > s) || (sq == s && q > p);
    }
    order[r] = p;
  }
  __syncthreads();
  if (soft) {
    // soft_oks_nms (nms.py:139-175): pick the best remaining pose, multiply every other remaining score by
    // exp(-oks^2 / thresh) (rescore(), 'gaussian'), repeat; at most 20 detections per image
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    for (int k = 0; k < P && k < 20; ++k) {
      double bv = -1.0;
      int bi = -1;
      for (int q = t; q < P; q += 256) {
        const double sq = sc[q] != sc[q] ? (double)INFINITY : sc[q];     // np.argmax picks a NaN first as well
        if (!dead[q] && (sq > bv || (sq == bv && q > bi))) {
          bv = sq;
          bi = q;
        }
      }
      red_v[t] = bv;
      red_i[t] = bi;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (t < o && (red_v[t + o] > red_v[t] || (red_v[t + o] == red_v[t] && red_i[t + o] > red_i[t]))) {
          red_v[t] = red_v[t + o];
          red_i[t] = red_i[t + o];
        }
        __syncthreads();
      }
      const int i = red_i[0];
      if (i < 0) break;                                  // uniform
      if (t == 0) {
        keep_rank[a + i] = k;
        dead[i] = 1;
      }
      for (int e = t; e < J * 3; e += 256) gk[e] = kpts[(long)(a + i) * J * 3 + e];
      __syncthreads();
      const double ag = areas[a + i];
      for (int d = t; d < P; d += 256) {
        if (dead[d]) continue;
        const float* kp = kpts + (long)(a + d) * J * 3;
        const double den = (ag + areas[a + d]) / 2 + 2.220446049250313e-16;
        double sum = 0.0;
        int cnt = 0;
        for (int j = 0; j < J; ++j) {
          if (use_vis && !((double)kp[3 * j + 2] > oks_vis_thre)) continue;
          const float dx = kp[3 * j] - gk[3 * j], dy = kp[3 * j + 1] - gk[3 * j + 1];
          const float d2 = dx * dx + dy * dy;
          sum += exp(-((double)d2 / svar[j] / den / 2));
          ++cnt;
        }
        const double iou = cnt ? sum / (double)cnt : 0.0;
        sc[d] = sc[d] * exp(-(iou * iou) / oks_thre);
      }
      __syncthreads();
    }
    return;
  }
  int kept = 0;
  for (int r = 0; r < P; ++r) {
    const int i = order[r];
    if (dead[i]) continue;          // uniform: dead[] was written before the last barrier
    if (t == 0) keep_rank[a + i] = kept;
    ++kept;
    for (int e = t; e < J * 3; e += 256) gk[e] = kpts[(long)(a + i) * J * 3 + e];
    __syncthreads();
    const double ag = areas[a + i];
    for (int q = r + 1 + t; q < P; q += 256) {
      const int d = order[q];
      if (dead[d]) continue;
      const float* k = kpts + (long)(a + d) * J * 3;
      const double den = (ag + areas[a + d]) / 2 + 2.220446049250313e-16;
      double sum = 0.0;
      int cnt = 0;
      for (int j = 0; j < J; ++j) {
        if (use_vis && !((double)k[3 * j + 2] > oks_vis_thre)) continue;
        const float dx = k[3 * j] - gk[3 * j], dy = k[3 * j + 1] - gk[3 * j + 1];
        const float d2 = dx * dx + dy * dy;
        const double e = (double)d2 / svar[j] / den / 2;
        sum += exp(-e);
        ++cnt;
      }
      const double iou = cnt ? sum / (double)cnt : 0.0;
      if (!(iou <= oks_thre)) dead[d] = 1;
    }
    __syncthreads();
  }
}

}  // namespace udp

using namespace udp;

extern "C" int udp_oks_nms(const float* kpts, const double* areas, const double* box_scores, const int32_t* img_offsets,
                           const int32_t* img_offsets_host, int n_images, int num_joints, const double* vars_dev,
                           double in_vis_thre, int rescore, double oks_thre, int use_vis, double oks_vis_thre, int soft,
                           double* scores_out, int32_t* keep_rank, void* stream) {
  if (!kpts || !areas || !box_scores || !img_offsets || !img_offsets_host || !vars_dev || !scores_out || !keep_rank)
    return fail(UDP_ERR_ARG, "udp_oks_nms: null pointer");
  if (n_images < 0 || num_joints <= 0 || num_joints > kMaxJoints) return fail(UDP_ERR_ARG, "udp_oks_nms: num_joints=%d", num_joints);
  for (int i = 0; i < n_images; ++i) {
    const int p = img_offsets_host[i + 1] - img_offsets_host[i];
    if (p < 0) return fail(UDP_ERR_ARG, "udp_oks_nms: image offsets must not decrease");
    if (p > kMaxPersons) return fail(UDP_ERR_UNSUPPORTED, "udp_oks_nms: %d persons in image %d (max %d)", p, i, kMaxPersons);
  }
  if (n_images == 0) return UDP_OK;
  hipLaunchKernelGGL(oks_nms_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, kpts, areas, box_scores,
                     img_offsets, num_joints, vars_dev, in_vis_thre, rescore ? 1 : 0, oks_thre, use_vis ? 1 : 0,
                     oks_vis_thre, soft ? 1 : 0, scores_out, keep_rank);
  UDP_HIP_CHECK(hipGetLastError());
  return UDP_OK;
}
