/*
 * udp_pose_hip.h -- C ABI of the MI355X-native UDP-Pose hot path (gfx950).
 *
 * One shared library (libudp_pose_hip.so), extern "C", plain pointers and sizes,
 * no torch types.  Every pointer is a DEVICE pointer unless the name ends in
 * `_host`.  Every entry point returns UDP_OK (0) or a negative error code and
 * never throws; udp_last_error() gives the message for the calling thread.
 * The caller owns all buffers; nothing is allocated after udp_hrnet_create().
 * All work is enqueued on the hipStream_t the caller passes (`stream`, a
 * hipStream_t cast to void*; NULL = the default stream) and is asynchronous.
 * A handle is single-stream; distinct handles are independent (one per GPU).
 *
 * The reference (realphongha/UDP-Pose) has no FFI layer: its seam is Python
 * call contracts.  Each entry below names the reference function (file:line
 * under /root/reference) whose arithmetic it replaces; INTEGRATION.md shows
 * the ctypes binding a maintainer would add on the reference side.
 */
#ifndef UDP_POSE_HIP_H
#define UDP_POSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UDP_POSE_ABI_VERSION 19

enum udp_status {
  UDP_OK = 0,
  UDP_ERR_ARG = -1,          /* bad argument (null pointer, shape, enum) */
  UDP_ERR_HIP = -2,          /* a HIP runtime call failed */
  UDP_ERR_UNSUPPORTED = -3,  /* shape / config outside what the kernels cover */
  UDP_ERR_WORKSPACE = -4     /* workspace too small */
};

/* Storage type of activations + weights.
 *   UDP_F32   fp32, exact fp32 MFMA (v_mfma_f32_16x16x4_f32): the reference's arithmetic.
 *   UDP_BF16  bf16 storage, fp32 accumulate: reduced precision (does NOT meet the 1e-3 parity contract).
 *   UDP_F16X2 split fp16: a value is the pair (hi, lo) with x ~= hi + lo * 2^-11 (22 significant bits),
 *             4 bytes per element laid out per pixel as [C hi][C lo]; products run as three fp16 MFMAs
 *             with fp32 accumulation.  The parity-grade throughput mode (fp32-level results, fp16 matrix
 *             pipe).  |x| >= 65520 overflows to NaN (never silently). */
enum udp_dtype { UDP_F32 = 0, UDP_BF16 = 1, UDP_F16X2 = 2 };

/* ABI version of the loaded library (== UDP_POSE_ABI_VERSION it was built with). */
int udp_abi_version(void);
/* UDP_F16X2 range guard.  Split-fp16 storage has fp16's range: a value of magnitude >= 65520 (or a NaN) about to be
 * stored raises a device-side flag in the kernel that produced it -- at the source, because a NaN does not survive
 * the next ReLU and a finiteness test of the heat-maps can miss the event.  Waits for `stream`, then returns 1 if
 * any split-fp16 store since the last reset left the range, 0 if none, < 0 on error; reset != 0 clears the flag.
 * The caller re-runs such a batch in UDP_F32 (udp_pose_amd.pose_engine does; the reference engine is fp32,
 * deep_hrnet/pose_engine.py:99-127). */
int udp_f16x2_overflow(void* stream, int reset);
/* Message of the last error on this thread ("" if none). */
const char* udp_last_error(void);

/* ------------------------------------------------------------------------- *
 * HRNet forward.  Replaces PoseHighResolutionNet.forward
 * (deep_hrnet/lib/models/pose_hrnet.py:436-471) incl. BasicBlock :29-59,
 * Bottleneck :62-100, HighResolutionModule.forward :260-273 with its fuse
 * layers :189-255 and the transitions :344-383; with flip_test != 0 also the
 * second forward on the W-mirrored batch of validate()
 * (deep_hrnet/lib/core/function.py:151-156).
 *
 * The network is handed over as a flat program of fused ops (one kernel launch
 * each) that the host builds from the reference's YAML + state_dict:
 * BatchNorm (eval) is folded into the conv weights/bias, so an op is
 *   out = act( conv(in) + bias [+ res] [+ sum_k nearest_up(up_k, 2^shift_k)] ).
 * Activations are NHWC in `dtype`; buffer b holds buf_elems[b] elements per
 * image and lives in the caller's workspace.
 * ------------------------------------------------------------------------- */
enum udp_op_kind {
  UDP_OP_STEM = 0, /* 3x3 s2 conv, Cin=3, reads the NCHW fp32 network input    */
  UDP_OP_CONV = 1, /* implicit-GEMM conv on MFMA, ks 1|3, stride 1|2           */
  UDP_OP_FUSE = 2, /* no conv: out = act(in [+ res] + sum_k nearest_up(up_k))  */
  UDP_OP_STEM7 = 3,   /* 7x7 s2 p3 conv, Cin=3, reads the NCHW fp32 network input (RSN top) */
  UDP_OP_MAXPOOL = 4, /* MaxPool2d(3, stride 2, pad 1) */
  UDP_OP_BILINEAR = 5, /* bilinear resize, align_corners=True, hin x win -> hout x wout */
  /* Polarized self-attention PSA_s (deep_hrnet/lib/models/PSA.py:190-269) of pose_hrnet_psa, per image;
   * w_off = fp32 block wq[C] | Wv[C/2][C] | W1[C/8][C/2] | b1 | ln_g | ln_b | W2[C][C/8] | b2[C] | Wg[C/2][C] */
  UDP_OP_PSA_POOL = 6,  /* in = x [C];   out = fp32 {sum_p softmax(wq.x)_p x_p, mean_p x_p}  (2C floats) */
  UDP_OP_PSA_MLP = 7,   /* in = POOL out; out = fp32 {channel mask m[C], gbar[C/2]}            (3C/2 floats) */
  UDP_OP_PSA_SCALE = 8, /* in = x, res = MLP out; out = x * m[c] */
  UDP_OP_BLOCK = 10,    /* fused BasicBlock, bf16, 32 channels: out = relu(conv3x3(relu(conv3x3(in)+b)) + b2 + in);
                           w_off/b_off = first conv, w2_off/b2_off = second conv (pose_hrnet.py:43-59) */
  UDP_OP_PSA_SP = 9     /* in = theta [C/2] (1x1 conv of the scaled map), res = scaled map [C], up_buf[0] = MLP out;
                           out = res * sigmoid(sum_j gbar_j softmax_HW(theta_j)) */
};

#define UDP_MAX_LANES 4
#define UDP_MAX_WAIT 8
#define UDP_BUF_NONE (-1)
#define UDP_BUF_OUTPUT (-2) /* out_buf: the NCHW fp32 heat-map output of the net */

typedef struct udp_conv_op {
  int32_t kind;            /* enum udp_op_kind */
  int32_t ks, stride;      /* kernel size 1|3 (pad = ks/2), stride 1|2 */
  int32_t relu;            /* apply ReLU in the epilogue */
  int32_t cin, cout;       /* real channel counts (cin multiple of 16 for UDP_OP_CONV) */
  int32_t cout_pad;        /* cout rounded up to a multiple of 32: rows of `weights`/`bias` */
  int32_t hin, win, hout, wout;
  int32_t in_buf, out_buf, res_buf; /* activation buffer ids; UDP_BUF_NONE / UDP_BUF_OUTPUT */
  int32_t n_up;            /* 0..3 nearest-upsampled addends */
  int32_t up_buf[3];
  int32_t up_shift[3];     /* up_k has (hout>>shift) x (wout>>shift) pixels, cout channels */
  int64_t w_off;           /* byte offset in the weight blob: [ks*ks][cout_pad][cin] in dtype
                              (UDP_OP_STEM: fp32 [27][cout]) */
  int64_t b_off;           /* byte offset of the fp32 bias [cout_pad] */
  /* Channel-slice views (elements; 0 pitch = dense): the op reads channels [in_coff, in_coff+cin) of
   * a tensor stored with in_pitch channels per pixel, writes [out_coff, out_coff+cout) of a tensor with
   * out_pitch channels, likewise for res.  torch.split / torch.cat of the RSN bottleneck
   * (RSN/exps/RSN18.coco/network.py:102-114) become views, never copies. */
  int32_t in_coff, in_pitch, out_coff, out_pitch, res_coff, res_pitch;
  int32_t lane;            /* 0..UDP_MAX_LANES-1: ops of different lanes may run concurrently
                              (HRNet branches); lane 0 runs on the caller's stream */
  int32_t n_wait;          /* cross-lane dependencies: this op starts after ops wait_op[0..n_wait) */
  int32_t wait_op[UDP_MAX_WAIT];
  int64_t w2_off, b2_off;  /* UDP_OP_BLOCK / chain_cout: the second conv's weights / bias in the blob */
  int32_t group;           /* != 0: consecutive UDP_OP_CONV ops with the same group id are independent of each
                              other (same-depth convs of different HRNet branches) and may share one launch */
  int32_t wfmt;            /* weight layout of a UDP_OP_CONV.  0: [tap][cout_pad][cin] in dtype (UDP_F16X2: per row the
                              cin hi values, then the cin lo values).  1 (UDP_F16X2 only): fragment-major for the
                              weight-stationary kernel -- 1 KiB blocks [tap][cin chunk of 32][cout pair of 32]
                              [block nb of 16][plane hi|lo], a block = 64 lanes x 8 fp16: lane kg*16 + li holds
                              w[tap][32*pair + 8*(li>>2) + 4*nb + (li&3)][32*chunk + 8*kg .. +7] (cin zero-padded to
                              a multiple of 32).  The stored numbers are the weights times 2^wexp: plane hi =
                              fp16(w * 2^wexp), plane lo = fp16(w * 2^wexp - hi) (the plain residual, NOT scaled by 2^11
                              as activations are).  udp_pose_amd.f16x2.pack_weights_ws builds it. */
  int32_t wexp;            /* wfmt 1: power-of-two scale of the stored weights, chosen so that max |w| * 2^wexp lies in
                              [2^13, 2^14) (|wexp| <= 40; 0 for an all-zero tensor).  The kernel keeps one fp32
                              accumulator of conv * 2^wexp and multiplies by 2^-wexp in its epilogue. */
  int32_t in_stuff2;       /* udp_conv2d_fused / _group, fp32 and bf16 storage: 1 = the input tensor is [n][hin/2][win/2][cin] and is read
                              as its zero-stuffed image (pixel (2i, 2j) = tensor pixel (i, j), zeros elsewhere): the input gradient of
                              a stride-2 conv = this stride-1 conv over the stuffed output gradient, without materialising it
                              (udp_zero_stuff2 + 4x the bytes); hin, win even.  0 elsewhere. */
  /* Second outputs (udp_hrnet_* programs, UDP_F16X2 convs with wfmt 1 and an NHWC output): besides `out`, the epilogue
   * writes out2_k = out + add2_k for k < n_out2 (0..2) -- the value AS STORED in `out` (22-bit split) plus channels
   * [add2_coff, add2_coff + cout) of another tensor of the output's resolution -- into channels [out2_coff, +cout) of a
   * third one.  The RSN bottleneck's element-wise sums between its 3x3 convs (RSN/exps/RSN18.coco/network.py:102-114:
   * `out_2_1 = conv(spx[1] + out_1_1)` ...) ride in the epilogue of the conv that produces their second operand instead of
   * being launches of their own; the numbers are those of UDP_OP_FUSE on the stored tensors, bit for bit.  With
   * n_out2 > 0, out_buf may be UDP_BUF_NONE: only the sums are stored. */
  int32_t n_out2;
  int32_t out2_buf[2], out2_coff[2], out2_pitch[2];
  int32_t add2_buf[2], add2_coff[2], add2_pitch[2];
  /* Chained 1x1 conv (udp_hrnet_* programs; chain_cout == 0: none).  A UDP_F16X2 1x1 stride-1 UDP_OP_CONV with wfmt 1,
   * cin 64 | 128, cout 256, a dense-or-sliced NHWC `out`, no up-sampled addends and no second outputs feeds
   * its result -- the value AS STORED in `out` -- straight into a second 1x1 conv + bias (+ ReLU if chain_relu) of
   * chain_cout (= 64) output channels, written densely to chain_buf: the Bottleneck chain of pose_hrnet.py:80-100
   * (conv3 + bn3 + shortcut + ReLU of one block, conv1 + bn1 + ReLU of the next) in one launch, the 4 * planes-channel
   * map written once and not read back.  The second conv's weights are fragment-major ([cout / 32 chunks][chain_cout / 32
   * pairs], scaled by 2^chain_wexp) at w2_off, its fp32 bias [chain_cout] at b2_off.  Both results equal those of the
   * two separate convs bit for bit. */
  int32_t chain_cout, chain_buf, chain_relu, chain_wexp;
} udp_conv_op;

typedef struct udp_hrnet udp_hrnet; /* opaque */

/* weights_dev: device blob (caller-owned, must outlive the handle).  ops and
 * buf_elems are host arrays, copied.  in_h/in_w: network input size. */
int udp_hrnet_create(const udp_conv_op* ops_host, int n_ops, const int64_t* buf_elems_host,
                     int n_bufs, const void* weights_dev, size_t weights_bytes, int dtype,
                     int in_h, int in_w, int out_channels, udp_hrnet** out);
/* Bytes of workspace udp_hrnet_forward needs for `n` input images (flip_test doubles it; with two sub-batch lanes:
 * both lanes' buffers plus, under flip_test, their heat-maps before they are copied into place). */
size_t udp_hrnet_workspace_bytes(const udp_hrnet* h, int n, int flip_test);
/* Sub-batch lanes udp_hrnet_forward(use_graph != 0) runs a batch of `n` images in: 2 = the batch is cut in two halves,
 * each replayed as its own hipGraph, the second on an internal stream beside the first (fork / join on the caller's
 * stream; results are those of one lane, images are independent); 1 otherwise.  Two lanes: UDP_F16X2 / UDP_BF16, inputs
 * up to 256x192, n >= 16 (environment UDP_POSE_LANES=1 / 2 forces one / two). */
int udp_hrnet_lanes(const udp_hrnet* h, int n);
/* in_nchw: fp32 [n,3,in_h,in_w].  heatmaps_nchw: fp32 [n*(flip_test?2:1), C, in_h/4, in_w/4];
 * with flip_test rows n..2n-1 are the raw outputs for the mirrored inputs (fuse them with
 * udp_flip_fuse).  use_graph != 0 replays a cached hipGraph of the launch sequence. */
int udp_hrnet_forward(udp_hrnet* h, const float* in_nchw, int n, int flip_test, void* workspace,
                      size_t workspace_bytes, float* heatmaps_nchw, int use_graph, void* stream);
/* Same launches as udp_hrnet_forward (eager, no graph) with a hipEvent pair around every launch on
 * `stream`; waits for completion and writes the elapsed milliseconds of op i to ms_per_op_host[i]
 * (udp_hrnet_num_launches entries; the time of a merged launch -- ops sharing a `group` -- is split over
 * its members in proportion to their FLOPs).  Measurement aid for bench.py's roofline section. */
int udp_hrnet_profile(udp_hrnet* h, const float* in_nchw, int n, int flip_test, void* workspace,
                      size_t workspace_bytes, float* heatmaps_nchw, float* ms_per_op_host, void* stream);
int udp_hrnet_destroy(udp_hrnet* h);
/* Number of kernel launches of one forward, and algorithmic FLOPs (2*MAC) per image. */
int udp_hrnet_num_launches(const udp_hrnet* h);
double udp_hrnet_flops_per_image(const udp_hrnet* h);

/* One fused conv launch on raw pointers (the operator the program above is made of; used by
 * the per-layer parity tests and kernel benchmarks).  `op` supplies kind (UDP_OP_CONV or
 * UDP_OP_FUSE), ks, stride, relu, cin, cout, cout_pad, hin, win, hout, wout, n_up, up_shift;
 * its buffer ids and blob offsets are ignored except out_buf == UDP_BUF_OUTPUT, which selects
 * the NCHW fp32 output form.  in/res/ups/out: NHWC `dtype`; weights [ks*ks][cout_pad][cin]
 * `dtype`; bias fp32 [cout_pad].  Replaces conv+BN(+add)(+ReLU), pose_hrnet.py:43-59.
 * UDP_OP_FUSE: out = act(in [+ res] + sum_k nearest_up(up_k, up_shift[k])), added in that order; up_shift 0 = an addend
 * of the output's own resolution (the training step's exchange-unit sums, pose_hrnet.py:266-272, have up to four terms);
 * weights / bias are ignored (may be NULL). */
int udp_conv2d_fused(const udp_conv_op* op_host, int dtype, int n, const void* in, const void* weights,
                     const float* bias, const void* res, const void* up0, const void* up1,
                     const void* up2, void* out, void* stream);
/* Training form of a plain conv (fp32 / bf16 NHWC, no addends, no ReLU): additionally leaves the statistics
 * pass of the BatchNorm2d that follows it (pose_hrnet.py:46, :83 ...: `bn(conv(x))` in train mode) in `bn_ws`:
 * one row of 2*cout doubles per workgroup tile -- sum x and sum x*x of the output AS STORED -- summed in the
 * conv epilogue instead of a separate pass over the tensor.  *bn_rows receives the row count
 * (<= udp_bn_rows_max()); feed both to udp_bn_train_fwd_from_sums. */
int udp_conv2d_fused_bn(const udp_conv_op* op_host, int dtype, int n, const void* in, const void* weights,
                        const float* bias, void* out, double* bn_ws, size_t bn_ws_doubles, int* bn_rows,
                        void* stream);

/* ------------------------------------------------------------------------- *
 * Flip-test fuse.  Replaces flip_back / flip_back_offset
 * (deep_hrnet/lib/utils/transforms.py:15-29 / :31-47) and
 * output = (output + output_flipped) * 0.5 (deep_hrnet/lib/core/function.py:161-171).
 * out[n,c,y,x] = 0.5f * (a[n,c,y,x] + sign[c] * b[n,src_ch[c],y,w-1-x]).
 * src_ch / sign: device arrays of length c (channel permutation and +-1).
 * out may alias a.
 * ------------------------------------------------------------------------- */
int udp_flip_fuse(const float* a, const float* b, const int32_t* src_ch, const float* sign,
                  int n, int c, int h, int w, float* out, void* stream);
/* Same, followed by a division: out = ((a + flip_back(b)) * 0.5f) / divisor.  The RSN test loop
 * divides the fused maps by 255 before decoding (RSN/exps/RSN18.coco.e1.se.36x8x132000_prm/test.py:174-185). */
int udp_flip_fuse_scaled(const float* a, const float* b, const int32_t* src_ch, const float* sign,
                         int n, int c, int h, int w, float divisor, float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * UDP decode.  Replaces get_final_preds (deep_hrnet/lib/core/inference.py:149-186):
 * get_max_preds :30-58, post (DARK/Taylor) :60-145, the offset branch :156-174 and
 * transform_preds :20-27.  heatmaps: fp32 [n, j*(offset?3:1), h, w] (not modified).
 * center, scale: fp64 [n,2]; cs_is_f32 != 0 says the caller's arrays were float32
 * (the reference then rounds scale*200, scale/(w-1) and scale*0.5 to fp32).
 * Outputs: preds fp64 [n,j,2] (image pixels), maxvals fp32 [n,j], preds_in fp64
 * [n,j,2] (preds_in_input_space), argidx int32 [n,j] (flat arg-max index; may be NULL).
 * ------------------------------------------------------------------------- */
int udp_decode_gaussian(const float* heatmaps, int n, int j, int h, int w, const double* center,
                        const double* scale, int cs_is_f32, int post_process, double* preds,
                        float* maxvals, double* preds_in, int32_t* argidx, void* stream);
int udp_decode_offset(const float* heatmaps, int n, int j, int h, int w, const double* center,
                      const double* scale, int cs_is_f32, float kpd, double* preds, float* maxvals,
                      double* preds_in, int32_t* argidx, void* stream);
/* Host helper: the 1-D Gaussian taps cv2.GaussianBlur(ksize, sigma=0) uses (fp32). */
int udp_gaussian_taps_host(int ksize, float* taps_host);

/* ------------------------------------------------------------------------- *
 * UDP data path.
 * udp_warp_affine: cv2.warpAffine(INTER_LINEAR, constant-0 border) on a uint8
 * HxWx3 frame + ToTensor + Normalize, one 2x3 dst->src matrix per crop
 * (deep_hrnet/pose_engine.py:69-85,40-43 with the inverse of
 * tools/infer_utils/utils.py:157-177; deep_hrnet/lib/dataset/JointsDataset.py:226-227
 * with get_warpmatrix :29-49).  mats: fp64 [n,6] dst->src.  out: fp32 [n,3,oh,ow].
 * ------------------------------------------------------------------------- */
int udp_warp_affine(const uint8_t* frame, int fh, int fw, int row_stride_bytes, const double* mats,
                    int n, int oh, int ow, const float* mean3, const float* std3, float* out,
                    void* stream);
/* Same warp with the training loader's two source transforms folded in: flip_lr = the frame is read
 * mirrored left-right (data_numpy[:, ::-1, :], JointsDataset.py:218-219; same fixed-point taps),
 * swap_rb = cv2.COLOR_BGR2RGB (:195-196).  mats: the get_warpmatrix dst->src matrix (:226-227). */
int udp_warp_affine_ex(const uint8_t* frame, int fh, int fw, int row_stride_bytes, const double* mats,
                       int n, int oh, int ow, const float* mean3, const float* std3, int flip_lr,
                       int swap_rb, float* out, void* stream);
/* AID information dropping (Cutout / HideAndSeek, deep_hrnet/lib/utils/transforms.py:144-224) on
 * normalized crops img fp32 [n,3,h,w]: dropped pixels become (0-mean)/std.  cutout: fp64
 * [n,n_patch,4] = (cx, cy, rx, ry) per ellipse (rx <= 0: unused).  hs_grid int32 [n] (0 = off) and
 * hs_mask uint8 [n,mask_x,mask_y]: cell (i,j) drops rows [i*g,(i+1)*g) for i*g < w and columns
 * [j*g,(j+1)*g) for j*g < h -- the reference's x/y-swapped indexing (:172-177), kept.  NULL = off. */
int udp_aid_apply(float* img, int n, int h, int w, const double* cutout, int n_patch,
                  const int32_t* hs_grid, const uint8_t* hs_mask, int mask_x, int mask_y,
                  const float* mean3, const float* std3, void* stream);
/* generate_target (JointsDataset.py:291-385).  joints fp32 [n,j,2] in crop pixels,
 * vis fp32 [n,j].  gaussian: target [n,j,h,w]; offset: [n,3j,h,w].  weight [n,j]. */
int udp_target_gaussian(const float* joints, const float* vis, int n, int j, int img_w, int img_h,
                        int hm_w, int hm_h, float sigma, float* target, float* weight, void* stream);
int udp_target_offset(const float* joints, const float* vis, int n, int j, int img_w, int img_h,
                      int hm_w, int hm_h, float kpd, float* target, float* weight, void* stream);

/* ------------------------------------------------------------------------- *
 * JointsMSELoss / JointsMSELoss_offset forward + gradient
 * (deep_hrnet/lib/core/loss.py:15-39 / :41-76).  pred, target: fp32 [b,c,hw];
 * weight fp32 [b,j].  loss_out: fp64 [2] = (L_hm, L_offset) (L_offset = 0 for the
 * plain loss); grad: fp32 like pred (d(L_hm+L_offset)/d pred), may be NULL.
 * The two scalars are accumulated with fp64 atomic adds of per-workgroup partial sums, so
 * their last bits may differ from run to run (~1e-16 relative); the gradient is computed per
 * element and does not depend on them.
 * ------------------------------------------------------------------------- */
int udp_mse_loss(const float* pred, const float* target, const float* weight, int b, int j, int hw,
                 int is_offset, double* loss_out, float* grad, void* stream);

/* ------------------------------------------------------------------------- *
 * Keypoint rescoring + OKS-NMS: deep_hrnet/lib/dataset/coco.py:321-356 with
 * oks_iou / oks_nms of deep_hrnet/lib/nms/nms.py:75-124, all images in one launch.
 * Persons of image i are rows img_offsets[i] .. img_offsets[i+1] (both a device and a host copy of the
 * int32 [n_images+1] offsets are passed; at most 1024 persons per image).  kpts fp32 [P,J,3] (x, y,
 * score) in image pixels, areas / box_scores fp64 [P], vars fp64 [J] = (2*sigma_j)^2.
 * rescore != 0: score = mean(joint scores > in_vis_thre) * box_score (coco.py:326-341), else box_score.
 * use_vis != 0 reproduces oks_iou's in_vis_thre branch (only joints of the candidate above
 * oks_vis_thre count).  scores_out fp64 [P]; keep_rank int32 [P]: position in the keep list (selection
 * order = descending score) or -1 when suppressed (overlap > oks_thre with a kept pose).
 * soft != 0: soft_oks_nms (nms.py:139-175, TEST.SOFT_NMS): instead of suppressing, the remaining scores
 * are multiplied by exp(-oks^2 / oks_thre) after every pick; at most 20 picks per image.  scores_out still
 * holds the (re)scores before NMS, as the reference never writes the decayed scores back.
 * ------------------------------------------------------------------------- */
int udp_oks_nms(const float* kpts, const double* areas, const double* box_scores,
                const int32_t* img_offsets, const int32_t* img_offsets_host, int n_images, int num_joints,
                const double* vars_dev, double in_vis_thre, int rescore, double oks_thre, int use_vis,
                double oks_vis_thre, int soft, double* scores_out, int32_t* keep_rank, void* stream);

/* ------------------------------------------------------------------------- *
 * Training step (deep_hrnet/lib/core/function.py:38-77: model.train(), forward,
 * criterion, zero_grad / backward / step; optimizer lib/utils/utils.py:70-74).
 * Activations are NHWC `dtype` with the channel count rounded up to 16 (zero
 * filled); parameters, gradients and optimizer state are fp32 in the reference's
 * state_dict layouts.  The host (udp-pose_amd/train.py) keeps the tape.
 * ------------------------------------------------------------------------- */
/* conv weight [cout][cin][ks][ks] fp32 -> operand layouts of udp_conv2d_fused:
 *   w_fwd   [ks*ks][cout_pad=ceil32(cout)][cin_k=ceil16(cin)]                      (forward)
 *   w_dgrad [ks*ks][ceil32(cin)][ceil16(cout)], taps mirrored, Cin/Cout swapped  (input gradient;
 *           may be NULL).  conv2d backward w.r.t. the input of a stride-1 conv is
 *   udp_conv2d_fused(dy, w_dgrad); for stride 2 apply it to udp_zero_stuff2(dy). */
int udp_pack_conv_weights(const float* w, int cout, int cin, int ks, int dtype, void* w_fwd,
                          void* w_dgrad, void* stream);
/* The same for every conv of a model in one launch: a device array of n descriptors (shapes are
 * validated by the host that builds the table; ks in {1,3}). */
typedef struct udp_pack_desc {
  const float* w;
  void* w_fwd;
  void* w_dgrad; /* may be NULL */
  int32_t cout, cin, ks, reserved;
} udp_pack_desc;
int udp_pack_conv_weights_batch(const udp_pack_desc* descs_dev, int n, int dtype, void* stream);
/* out[n][2y][2x][c] = dy[n][y][x][c], zero elsewhere (out: [n][2h][2w][c]). */
int udp_zero_stuff2(const void* dy, int n, int h, int w, int c, int dtype, void* out, void* stream);
/* dW[co][ci][ky][kx] (+)= sum_{n,y,x} dy[n,y,x,co] * x[n, y*s+ky-ks/2, x*s+kx-ks/2, ci]
 * (torch.nn.functional.conv2d backward w.r.t. weight).  x: [n,hin,win,cin_k], dy: [n,hout,wout,cout_k].
 * workspace: split-K partials, at least ks*ks*cout*cin*4 bytes (udp_conv2d_wgrad_workspace_bytes
 * suggests a size that keeps the chip full). */
size_t udp_conv2d_wgrad_workspace_bytes(int cout, int cin, int ks);
int udp_conv2d_wgrad(const void* x, const void* dy, int n, int hin, int win, int cin_k, int hout,
                     int wout, int cout_k, int ks, int stride, int cout, int cin, int dtype, float* dw,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The weight gradients of up to 4 convs (the same-depth convs of the HRNet branches): the partial-sum kernels run
 * one per member, the fixed-order reduces of all members as ONE launch; results are those of udp_conv2d_wgrad per
 * member, bit for bit.  Every member needs its own workspace. */
typedef struct udp_wgrad_item {
  const void* x;
  const void* dy;
  float* dw;
  void* workspace;
  size_t workspace_bytes;
  int32_t n, hin, win, cin_k, hout, wout, cout_k, ks, stride, cout, cin, accumulate;
} udp_wgrad_item;
/* Diagnostic, host only (no GPU): the dispatch order of a merged weight-stationary launch whose member j has
 * tiles[j] pixel tiles x ncby[j] cout blocks and kernel variant code[j] (1, 2 or 4): for every flat workgroup
 * b < total the member, tile and cout block it computes, resolved as the kernel resolves them.  Returns total
 * (<= cap), -1 on error; *used_table = 1 when the one-load lookup table applies.  Tests check that every
 * (member, tile, cout block) is computed exactly once. */
int udp_debug_multi_order(const unsigned* tiles, const unsigned* ncby, const int* code, int n, unsigned cap,
                          unsigned* out_member, unsigned* out_tile, unsigned* out_cby, int* used_table);
int udp_conv2d_wgrad_group(const udp_wgrad_item* items, int n_items, int dtype, void* stream);
/* Up to 4 independent plain convs (same dtype and batch n) in as few launches as possible -- the same-depth convs
 * of the HRNet branches in the training step (pose_hrnet.py:253-256): members whose tile fits the merged kernel run
 * as ONE launch, the rest on their own; per-member results are those of udp_conv2d_fused / udp_conv2d_fused_bn.
 * res (optional) is accumulated as in udp_conv2d_fused; bn_ws != NULL (then no res / ReLU): BatchNorm partial rows
 * as udp_conv2d_fused_bn leaves them, their count in bn_rows. */
typedef struct udp_conv_item {
  const udp_conv_op* op;
  const void* in;
  const void* weights;
  const float* bias;
  const void* res;
  void* out;
  double* bn_ws;
  size_t bn_ws_doubles;
  int32_t bn_rows;     /* out */
  int32_t reserved;
} udp_conv_item;
int udp_conv2d_fused_group(udp_conv_item* items, int n_items, int dtype, int n, void* stream);
/* nn.BatchNorm2d in train mode over x [m = N*H*W rows][c]: batch mean / biased variance (fp64 sums),
 * running_mean/var <- (1-momentum)*running + momentum*batch (unbiased variance), either may be NULL;
 * y = [relu](xhat*gamma + beta [+ res]).  save_mean / save_invstd fp32 [c] feed the backward.
 * c must be a multiple of 4.  ws: udp_bn_workspace_doubles(c) doubles of scratch (per-block partial
 * sums, summed in a fixed order: results are run-to-run deterministic).  A workspace serves one
 * stream at a time. */
size_t udp_bn_workspace_doubles(int c);
int udp_bn_train_fwd(const void* x, int64_t m, int c, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* save_mean,
                     float* save_invstd, const void* res, int relu, void* y, int dtype, double* ws,
                     void* stream);
/* The same with the statistics pass already done: `ws` holds `rows` partial rows of 2*c doubles
 * ([row][0..c) = sum x, [row][c..2c) = sum x*x), as udp_conv2d_fused_bn leaves them.  rows <= udp_bn_rows_max(). */
int udp_bn_rows_max(void);
int udp_bn_train_fwd_from_sums(const void* x, int64_t m, int c, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* save_mean,
                               float* save_invstd, const void* res, int relu, void* y, int dtype, double* ws,
                               int rows, void* stream);
/* Backward of the above.  g = dy * (y_relu > 0) when y_relu != NULL (the ReLU that followed), else dy.
 * dgamma = sum g*xhat, dbeta = sum g, dx = gamma*invstd*(g - dbeta/m - xhat*dgamma/m);
 * g_out (optional) receives g -- the gradient of the residual input. */
int udp_bn_train_bwd(const void* x, const void* dy, const void* y_relu, int64_t m, int c,
                     const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma,
                     float* dbeta, void* dx, void* g_out, int dtype, double* ws, void* stream);
/* Up to 4 BatchNorms per call (the bn1 / bn2 of one block depth of every HRNet branch are independent,
 * pose_hrnet.py:253-256: train.py runs the branches in lock step): each pass is ONE launch over all tensors,
 * per-tensor arithmetic and results are those of the single-tensor entry points above, bit for bit.
 * Forward reads x, gamma, beta, running_*, res, relu, rows (> 0: `ws` already holds that many partial rows from
 * udp_conv2d_fused_bn; 0: the partial sums are computed here) and writes save_*, y.  Backward reads x, dy, y_relu,
 * gamma, save_* and writes dgamma, dbeta, dx, g_out (optional).  Every tensor needs its own `ws`
 * (udp_bn_workspace_doubles(c) doubles). */
typedef struct udp_bn_item {
  const void* x;
  const void* dy;
  const void* y_relu;
  const void* res;
  void* y;
  void* dx;
  void* g_out;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float* dgamma;
  float* dbeta;
  double* ws;
  int64_t m;
  int32_t c, rows, relu, reserved;
} udp_bn_item;
int udp_bn_train_fwd_multi(const udp_bn_item* items, int n, float eps, float momentum, int dtype, void* stream);
int udp_bn_train_bwd_multi(const udp_bn_item* items, int n, int dtype, void* stream);
/* acc[n,y,x,c] (init ? = : +=) src[n, y>>shift, x>>shift, c], optional ReLU: the sum nodes of
 * HighResolutionModule.forward (pose_hrnet.py:266-272) with nn.Upsample(mode='nearest'). */
int udp_ew_accumulate(void* acc, const void* src, int n, int h, int w, int c, int shift, int init,
                      int relu, int dtype, void* stream);
/* g = dy * (y > 0). */
int udp_relu_bwd(const void* dy, const void* y, void* g, int64_t count, int dtype, void* stream);
/* Backward of nearest upsampling by 2^shift: du[n,y,x,c] (+)= sum of the 2^shift x 2^shift block of g
 * (g: [n,h,w,c], du: [n,h>>shift,w>>shift,c]). */
int udp_upsample_bwd(const void* g, int n, int h, int w, int c, int shift, void* du, int accumulate,
                     int dtype, void* stream);
/* db[c] = sum over the m rows of g[row*c_pitch + c] (conv bias gradient). */
int udp_bias_grad(const void* g, int64_t m, int c_pitch, int c, float* db, int dtype, void* stream);
/* NCHW fp32 [n,c,h,w] -> NHWC `dtype` [n,h,w,c_pad], channels c..c_pad zero. */
int udp_nchw_to_nhwc(const float* src, int n, int c, int h, int w, int c_pad, void* dst, int dtype,
                     void* stream);
/* torch.optim.Adam (no weight decay, no amsgrad) over flat fp32 buffers; step counts from 1.
 * The gradient is read as g*grad_scale (1/world_size after a SUM all-reduce; 1 otherwise). */
int udp_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, void* stream);
/* The same for a captured training step (hipGraph replay): the two step-dependent scalars
 * coef = {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} -- exactly the values udp_adam_step uses, as
 * udp_adam_coefficients computes them on the host -- are read from device memory at run time, so the host only
 * refreshes those 8 bytes before each replay. */
int udp_adam_coefficients(float lr, float beta1, float beta2, int step, float* coef_host);
int udp_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t count, float beta1, float beta2,
                      float eps, const float* coef_dev, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UDP_POSE_HIP_H */
