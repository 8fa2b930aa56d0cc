// HRNet forward executor: runs the host-built program of fused conv launches
// (include/udp_pose_hip.h, udp_conv_op) on one HIP stream, optionally replaying
// it as a hipGraph.  Replaces PoseHighResolutionNet.forward,
// deep_hrnet/lib/models/pose_hrnet.py:436-471.
#include <cstdlib>
#include <vector>

#include "common.h"

namespace udp {
int describe_conv(ConvParams p, int dtype, int ks, int stride, Launch* out);
int describe_stem(const ConvParams& p, int dtype, Launch* out);
int describe_fuse(const ConvParams& p, int dtype, Launch* out);
int describe_stem7(const ConvParams& p, int dtype, Launch* out);
int describe_maxpool(const ConvParams& p, int dtype, Launch* out);
int describe_bilinear(const ConvParams& p, int dtype, Launch* out);
int describe_psa(const ConvParams& p, int dtype, int kind, Launch* out);
int describe_block(ConvParams p, int dtype, Launch* out);
int describe_conv_grouped(ConvParams p, int dtype, int ks, int stride, Launch* out);
int describe_multi(const Launch* members, int n, ConvMulti* m, Launch* out);
int run_launch(const Launch& l, hipStream_t s);
}  // namespace udp

using namespace udp;

struct GraphEntry {
  int n, flip;
  const void* in;
  void* ws;
  void* out;
  hipGraph_t graph;
  hipGraphExec_t exec;
};

struct udp_hrnet {
  std::vector<udp_conv_op> ops;
  std::vector<int64_t> buf_elems;  // per image, rounded up to 64 elements
  std::vector<int64_t> buf_off;    // prefix sums (elements per image)
  int64_t total_elems = 0;
  const char* weights = nullptr;
  size_t weights_bytes = 0;
  int dtype = UDP_F32;
  int in_h = 0, in_w = 0, out_channels = 0;
  double flops = 0.0;
  std::vector<GraphEntry> graphs;
  // branch-parallel execution: lane 0 = caller's stream, lanes 1.. = internal streams
  int n_lanes = 1;
  hipStream_t lane_stream[UDP_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> op_done;     // one per op (recorded only where another lane waits on it)
  std::vector<char> op_signals;        // op has a cross-lane consumer
  hipEvent_t ev_fork = nullptr, ev_join[UDP_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  // sub-batch lanes (udp_hrnet_forward): the second half of a batch runs on this stream beside the first
  hipStream_t split_stream = nullptr;
  hipEvent_t split_fork = nullptr, split_join = nullptr;
};

static size_t esize(int dtype) { return dtype == UDP_BF16 ? 2 : 4; }   // bytes per stored element (F16X2: hi + lo)

static int validate_op(const udp_hrnet* h, const udp_conv_op& o, int idx) {
  const int nb = (int)h->buf_elems.size();
  auto buf_ok = [&](int b, int64_t need) { return b >= 0 && b < nb && h->buf_elems[b] >= need; };
  const int64_t out_need = (int64_t)o.hout * o.wout * (o.out_pitch ? o.out_pitch : o.cout);
  if (o.kind < UDP_OP_STEM || o.kind > UDP_OP_BLOCK) return fail(UDP_ERR_ARG, "op %d: bad kind %d", idx, o.kind);
  if (o.lane < 0 || o.lane >= UDP_MAX_LANES || o.n_wait < 0 || o.n_wait > UDP_MAX_WAIT) return fail(UDP_ERR_ARG, "op %d: lane/n_wait", idx);
  for (int k = 0; k < o.n_wait; ++k)
    if (o.wait_op[k] < 0 || o.wait_op[k] >= idx) return fail(UDP_ERR_ARG, "op %d: wait_op %d must name an earlier op", idx, o.wait_op[k]);
  if (o.kind == UDP_OP_BLOCK) {
    const int64_t hw = (int64_t)o.hin * o.win;
    if (h->dtype != UDP_BF16 || o.cin != 32 || o.cout != 32 || o.hin != o.hout || o.win != o.wout || o.hin % 8)
      return fail(UDP_ERR_ARG, "op %d: fused BasicBlock needs bf16, 32 channels, height %% 8 == 0", idx);
    if (!buf_ok(o.in_buf, hw * 32) || !buf_ok(o.out_buf, hw * 32) || o.in_buf == o.out_buf)
      return fail(UDP_ERR_ARG, "op %d: fused BasicBlock buffers", idx);
    const size_t wbytes = (size_t)9 * 32 * 32 * 2;
    const int64_t offs[4] = {o.w_off, o.w2_off, o.b_off, o.b2_off};
    for (int k = 0; k < 4; ++k)
      if (offs[k] < 0 || (size_t)offs[k] + (k < 2 ? wbytes : 32 * 4) > h->weights_bytes || (offs[k] & 15))
        return fail(UDP_ERR_ARG, "op %d: fused BasicBlock weight/bias range outside the blob or misaligned", idx);
    return UDP_OK;
  }
  if (o.kind >= UDP_OP_PSA_POOL) {
    // polarized self-attention ops: per-image side buffers hold fp32 rows, counted in `dtype` elements
    const int64_t f = 4 / (int64_t)esize(h->dtype);
    const int64_t hw = (int64_t)o.hin * o.win;
    const int C = o.kind == UDP_OP_PSA_SP ? o.cout : o.cin;
    if (C <= 0 || hw <= 0 || C % 16) return fail(UDP_ERR_ARG, "op %d: PSA shape", idx);
    const size_t wbytes = (size_t)(C + (C / 2) * C + (C / 8) * (C / 2) + 3 * (C / 8) + C * (C / 8) + C + (C / 2) * C) * 4;
    bool ok = true;
    switch (o.kind) {
      case UDP_OP_PSA_POOL: ok = buf_ok(o.in_buf, hw * C) && buf_ok(o.out_buf, 2 * C * f); break;
      case UDP_OP_PSA_MLP: ok = buf_ok(o.in_buf, 2 * C * f) && buf_ok(o.out_buf, (C + C / 2) * f); break;
      case UDP_OP_PSA_SCALE: ok = buf_ok(o.in_buf, hw * C) && buf_ok(o.res_buf, (C + C / 2) * f) && buf_ok(o.out_buf, hw * C); break;
      default: ok = buf_ok(o.in_buf, hw * (C / 2)) && buf_ok(o.res_buf, hw * C) && o.n_up == 1 && buf_ok(o.up_buf[0], (C + C / 2) * f) && buf_ok(o.out_buf, hw * C);
    }
    if (!ok) return fail(UDP_ERR_ARG, "op %d: PSA buffers missing or too small", idx);
    if (o.kind <= UDP_OP_PSA_MLP && (o.w_off < 0 || (size_t)o.w_off + wbytes > h->weights_bytes || (o.w_off & 15)))
      return fail(UDP_ERR_ARG, "op %d: PSA parameter block outside the blob or misaligned", idx);
    return UDP_OK;
  }
  const bool is_stem = o.kind == UDP_OP_STEM || o.kind == UDP_OP_STEM7;
  const bool has_w = is_stem || o.kind == UDP_OP_CONV;
  const int ipitch = o.in_pitch ? o.in_pitch : o.cin, opitch = o.out_pitch ? o.out_pitch : o.cout;
  const int rpitch = o.res_pitch ? o.res_pitch : o.cout;
  if (o.in_coff < 0 || o.out_coff < 0 || o.res_coff < 0 || o.in_coff + o.cin > ipitch || o.out_coff + o.cout > opitch ||
      o.res_coff + o.cout > rpitch)
    return fail(UDP_ERR_ARG, "op %d: channel view outside its tensor", idx);
  if (o.cout <= 0 || o.hout <= 0 || o.wout <= 0 || o.cout_pad < o.cout || o.cout_pad % 32)
    return fail(UDP_ERR_ARG, "op %d: bad output shape / cout_pad", idx);
  if (o.out_buf == UDP_BUF_OUTPUT) {
    if (o.kind != UDP_OP_CONV || o.cout != h->out_channels || o.hout * 4 != h->in_h || o.wout * 4 != h->in_w)
      return fail(UDP_ERR_ARG, "op %d: output op must be a conv producing [C=%d,%d,%d]", idx, h->out_channels,
                  h->in_h / 4, h->in_w / 4);
    if (o.res_buf != UDP_BUF_NONE || o.n_up != 0) return fail(UDP_ERR_ARG, "op %d: output op takes no addends", idx);
  } else if (o.out_buf == UDP_BUF_NONE && o.n_out2 > 0 && o.kind == UDP_OP_CONV) {
    // only the second outputs are wanted (nothing reads the plain conv output): its stores are dropped
  } else if (!buf_ok(o.out_buf, out_need)) {
    return fail(UDP_ERR_ARG, "op %d: out_buf %d missing or too small", idx, o.out_buf);
  }
  if (is_stem) {
    const int want = o.kind == UDP_OP_STEM ? 3 : 7;
    if (o.ks != want || o.stride != 2 || o.cin != 3 || o.hin != h->in_h || o.win != h->in_w)
      return fail(UDP_ERR_ARG, "op %d: stem must be %dx%d s2 on the %dx%d input", idx, want, want, h->in_h, h->in_w);
  } else {
    if (!buf_ok(o.in_buf, (int64_t)o.hin * o.win * ipitch)) return fail(UDP_ERR_ARG, "op %d: in_buf %d missing or too small", idx, o.in_buf);
    if (o.in_buf == o.out_buf && o.kind == UDP_OP_CONV && o.ks != 1) return fail(UDP_ERR_ARG, "op %d: in-place 3x3 conv is not supported", idx);
  }
  if (o.kind == UDP_OP_MAXPOOL) {
    if (o.cin != o.cout || o.hout != (o.hin + 2 - 3) / 2 + 1 || o.wout != (o.win + 2 - 3) / 2 + 1)
      return fail(UDP_ERR_ARG, "op %d: maxpool 3x3 s2 p1 shape", idx);
  } else if (o.kind == UDP_OP_BILINEAR) {
    if (o.cin != o.cout) return fail(UDP_ERR_ARG, "op %d: bilinear resize keeps the channel count", idx);
  } else if (has_w) {
    if ((o.ks != 1 && o.ks != 3 && o.ks != 7) || (o.stride != 1 && o.stride != 2)) return fail(UDP_ERR_ARG, "op %d: ks/stride", idx);
    const int pad = o.ks / 2;
    if (o.hout != (o.hin + 2 * pad - o.ks) / o.stride + 1 || o.wout != (o.win + 2 * pad - o.ks) / o.stride + 1)
      return fail(UDP_ERR_ARG, "op %d: output size does not match input/stride", idx);
    if (o.wfmt != 0 && (o.wfmt != 1 || is_stem || h->dtype != UDP_F16X2))
      return fail(UDP_ERR_ARG, "op %d: wfmt %d (fragment-major weights need a UDP_F16X2 UDP_OP_CONV)", idx, o.wfmt);
    const size_t wbytes = is_stem ? (size_t)o.ks * o.ks * 3 * o.cout * 4
                          : o.wfmt == 1 ? (size_t)o.ks * o.ks * ((o.cin + 31) / 32) * (o.cout_pad / 32) * 4096
                                        : (size_t)o.ks * o.ks * o.cout_pad * o.cin * esize(h->dtype);
    if (o.w_off < 0 || (size_t)o.w_off + wbytes > h->weights_bytes || (o.w_off & 15))
      return fail(UDP_ERR_ARG, "op %d: weight range outside the blob or misaligned", idx);
    if (o.b_off < 0 || (size_t)o.b_off + (size_t)o.cout_pad * 4 > h->weights_bytes || (o.b_off & 15))
      return fail(UDP_ERR_ARG, "op %d: bias range outside the blob or misaligned", idx);
  } else if (o.hin != o.hout || o.win != o.wout || o.cin != o.cout) {
    return fail(UDP_ERR_ARG, "op %d: fuse op must keep the shape", idx);
  }
  if (o.lane < 0 || o.lane >= UDP_MAX_LANES || o.n_wait < 0 || o.n_wait > UDP_MAX_WAIT) return fail(UDP_ERR_ARG, "op %d: lane/n_wait", idx);
  for (int k = 0; k < o.n_wait; ++k)
    if (o.wait_op[k] < 0 || o.wait_op[k] >= idx) return fail(UDP_ERR_ARG, "op %d: wait_op %d must name an earlier op", idx, o.wait_op[k]);
  if (o.res_buf != UDP_BUF_NONE && !buf_ok(o.res_buf, (int64_t)o.hout * o.wout * rpitch)) return fail(UDP_ERR_ARG, "op %d: res_buf", idx);
  if (o.n_up < 0 || o.n_up > 3) return fail(UDP_ERR_ARG, "op %d: n_up", idx);
  for (int u = 0; u < o.n_up; ++u) {
    const int s = o.up_shift[u];
    if (s < 1 || s > 5 || (o.hout & ((1 << s) - 1)) || (o.wout & ((1 << s) - 1)))
      return fail(UDP_ERR_ARG, "op %d: up_shift %d does not divide %dx%d", idx, s, o.hout, o.wout);
    if (!buf_ok(o.up_buf[u], (int64_t)(o.hout >> s) * (o.wout >> s) * o.cout)) return fail(UDP_ERR_ARG, "op %d: up_buf %d", idx, u);
  }
  if (o.chain_cout) {
    if (o.kind != UDP_OP_CONV || o.wfmt != 1 || h->dtype != UDP_F16X2 || o.ks != 1 || o.stride != 1 || o.group || o.n_up || o.n_out2 ||
        o.out_buf < 0 || (o.cin != 64 && o.cin != 128) || o.cout != 256 || o.cout_pad != o.cout || o.chain_cout != 64)
      return fail(UDP_ERR_UNSUPPORTED, "op %d: a chained conv needs an ungrouped split-fp16 1x1 conv (wfmt 1), 64 | 128 -> 256 channels, and 64 chained outputs", idx);
    if (!buf_ok(o.chain_buf, (int64_t)o.hout * o.wout * o.chain_cout) || o.chain_buf == o.out_buf || o.chain_buf == o.in_buf || o.chain_buf == o.res_buf)
      return fail(UDP_ERR_ARG, "op %d: chain_buf %d missing, too small or aliased", idx, o.chain_buf);
    const size_t w2bytes = (size_t)(o.cout / 32) * (o.chain_cout / 32) * 4096;
    if (o.w2_off < 0 || (size_t)o.w2_off + w2bytes > h->weights_bytes || (o.w2_off & 15) || o.b2_off < 0 ||
        (size_t)o.b2_off + (size_t)o.chain_cout * 4 > h->weights_bytes || (o.b2_off & 15) || o.chain_wexp < -40 || o.chain_wexp > 40)
      return fail(UDP_ERR_ARG, "op %d: chained conv: weight / bias range outside the blob or misaligned", idx);
  }
  if (o.n_out2 < 0 || o.n_out2 > 2) return fail(UDP_ERR_ARG, "op %d: n_out2", idx);
  if (o.n_out2 && (o.kind != UDP_OP_CONV || o.wfmt != 1 || h->dtype != UDP_F16X2 || o.out_buf == UDP_BUF_OUTPUT || o.cout % 8 || o.ks != 3 || o.stride != 1 || o.group))
    return fail(UDP_ERR_UNSUPPORTED, "op %d: second outputs need an ungrouped 3x3 stride-1 split-fp16 conv with fragment-major weights, an NHWC output and cout %% 8 == 0", idx);
  for (int k = 0; k < o.n_out2; ++k) {
    const int op = o.out2_pitch[k] ? o.out2_pitch[k] : o.cout, ap = o.add2_pitch[k] ? o.add2_pitch[k] : o.cout;
    if (o.out2_coff[k] < 0 || o.out2_coff[k] + o.cout > op || o.add2_coff[k] < 0 || o.add2_coff[k] + o.cout > ap ||
        (o.out2_coff[k] | o.add2_coff[k] | op | ap) % 8)
      return fail(UDP_ERR_ARG, "op %d: second output %d: channel views", idx, k);
    if (!buf_ok(o.out2_buf[k], (int64_t)o.hout * o.wout * op) || !buf_ok(o.add2_buf[k], (int64_t)o.hout * o.wout * ap))
      return fail(UDP_ERR_ARG, "op %d: second output %d: buffers", idx, k);
  }
  return UDP_OK;
}

extern "C" int udp_abi_version(void) { return UDP_POSE_ABI_VERSION; }
extern "C" const char* udp_last_error(void) { return err_buf(); }

extern "C" int udp_f16x2_overflow(void* stream, int reset) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int flag = 0;
  int rc = conv_h2_overflow(s, reset, &flag);
  if (!rc) rc = conv_ws_h2_overflow(s, reset, &flag);
  if (!rc) rc = psa_h2_overflow(s, reset, &flag);
  return rc ? rc : flag;
}

extern "C" int udp_hrnet_create(const udp_conv_op* ops, int n_ops, const int64_t* buf_elems, int n_bufs,
                                const void* weights_dev, size_t weights_bytes, int dtype, int in_h, int in_w,
                                int out_channels, udp_hrnet** out) {
  if (!ops || n_ops <= 0 || !buf_elems || n_bufs <= 0 || !weights_dev || !out)
    return fail(UDP_ERR_ARG, "udp_hrnet_create: null/empty argument");
  if (dtype != UDP_F32 && dtype != UDP_BF16 && dtype != UDP_F16X2) return fail(UDP_ERR_ARG, "udp_hrnet_create: dtype %d", dtype);
  if (in_h <= 0 || in_w <= 0 || (in_h % 32) || (in_w % 32))
    return fail(UDP_ERR_UNSUPPORTED, "udp_hrnet_create: input %dx%d must be a multiple of 32", in_h, in_w);
  if (reinterpret_cast<uintptr_t>(weights_dev) & 15) return fail(UDP_ERR_ARG, "udp_hrnet_create: weights not 16-byte aligned");
  udp_hrnet* h = new udp_hrnet();
  h->dtype = dtype;
  h->in_h = in_h;
  h->in_w = in_w;
  h->out_channels = out_channels;
  h->weights = reinterpret_cast<const char*>(weights_dev);
  h->weights_bytes = weights_bytes;
  h->buf_off.resize(n_bufs);
  for (int b = 0; b < n_bufs; ++b) {
    if (buf_elems[b] <= 0) {
      delete h;
      return fail(UDP_ERR_ARG, "udp_hrnet_create: buffer %d has %lld elements", b, (long long)buf_elems[b]);
    }
    const int64_t e = (buf_elems[b] + 63) / 64 * 64;
    h->buf_elems.push_back(e);
    h->buf_off[b] = h->total_elems;
    h->total_elems += e;
  }
  bool has_out = false;
  for (int i = 0; i < n_ops; ++i) {
    const int rc = validate_op(h, ops[i], i);
    if (rc) {
      delete h;
      return rc;
    }
    has_out |= ops[i].out_buf == UDP_BUF_OUTPUT;
    if (ops[i].kind == UDP_OP_STEM || ops[i].kind == UDP_OP_STEM7 || ops[i].kind == UDP_OP_CONV)
      h->flops += 2.0 * ops[i].ks * ops[i].ks * ops[i].cin * ops[i].cout * ops[i].hout * ops[i].wout;
    if (ops[i].kind == UDP_OP_BLOCK) h->flops += 2 * 2.0 * 9 * 32 * 32 * ops[i].hout * ops[i].wout;
    if (ops[i].kind == UDP_OP_CONV && ops[i].chain_cout) h->flops += 2.0 * ops[i].cout * ops[i].chain_cout * ops[i].hout * ops[i].wout;
    h->ops.push_back(ops[i]);
  }
  if (!has_out || (h->ops[0].kind != UDP_OP_STEM && h->ops[0].kind != UDP_OP_STEM7) || h->ops[0].lane != 0 || h->ops.back().lane != 0 ||
      h->ops.back().out_buf != UDP_BUF_OUTPUT) {
    delete h;
    return fail(UDP_ERR_ARG, "udp_hrnet_create: program needs a stem op first and the output op last, both on lane 0");
  }
  h->op_signals.assign(n_ops, 0);
  for (int i = 0; i < n_ops; ++i) {
    if (ops[i].lane + 1 > h->n_lanes) h->n_lanes = ops[i].lane + 1;
    for (int k = 0; k < ops[i].n_wait; ++k) h->op_signals[ops[i].wait_op[k]] = 1;
  }
  h->op_done.assign(n_ops, nullptr);
  hipError_t e = hipSuccess;
  for (int l = 1; l < h->n_lanes && e == hipSuccess; ++l) e = hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking);
  for (int l = 1; l < h->n_lanes && e == hipSuccess; ++l) e = hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  for (int i = 0; i < n_ops && e == hipSuccess; ++i)
    if (h->op_signals[i]) e = hipEventCreateWithFlags(&h->op_done[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    udp_hrnet_destroy(h);
    return fail(UDP_ERR_HIP, "udp_hrnet_create: stream/event creation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return UDP_OK;
}

static size_t ws_bytes_one(const udp_hrnet* h, int n, int flip) {
  const size_t b = (size_t)h->total_elems * (size_t)(n * (flip ? 2 : 1)) * esize(h->dtype);
  return (b + 255) & ~(size_t)255;
}
static size_t out_bytes(const udp_hrnet* h, int images) {
  return (size_t)images * h->out_channels * (size_t)(h->in_h / 4) * (size_t)(h->in_w / 4) * sizeof(float);
}
// Sub-batch lanes: a batch of the split-fp16 mode is run as two halves, each its own hipGraph, the second on an
// internal stream BESIDE the first (fork / join by events on the caller's stream).  Measured with two independent
// graph replays on two streams (tools/dual_stream.py, 64 crops + mirrored copies): W32 256x192 7.39 -> 7.81-7.92 k
// images/s, RSN-18 9.67 -> 10.93 k, W48 384x288 1.55 -> 1.53 k; four lanes: 4.5 k.  The launches of one chain leave
// the chip idle between a kernel's last workgroups and its successor's first HBM round trip (42 % of the workgroup
// slot time of the dominant launch is outside its MFMA loop, NOTES.md); a second, independent chain fills those gaps.
// Images are independent (same result for an image whatever batch it arrives in -- tests/test_gpu_e2e.py), so the
// split changes no number.  Default: split-fp16 and bf16 modes (bf16: 13.2 -> 14.3 k), inputs up to 256x192, 16 crops
// or more; UDP_POSE_LANES=1 / 2 forces it off / on.
static bool split_ok(const udp_hrnet* h, int n) {
  if (n < 16) return false;
  const char* e = getenv("UDP_POSE_LANES");
  if (e) return atoi(e) >= 2;
  return h->dtype != UDP_F32 && (long)h->in_h * h->in_w <= 256L * 192L;   // (fp32 mode: 2.86 -> 2.70 k with two lanes)
}
extern "C" int udp_hrnet_lanes(const udp_hrnet* h, int n) { return h && n > 0 && split_ok(h, n) ? 2 : 1; }
extern "C" size_t udp_hrnet_workspace_bytes(const udp_hrnet* h, int n, int flip_test) {
  if (!h || n <= 0) return 0;
  if (!split_ok(h, n)) return ws_bytes_one(h, n, flip_test);
  const int n0 = (n + 1) / 2;
  // two lanes' workspaces + (flip test) their [normal | mirrored] heat-maps before they are copied into place
  return ws_bytes_one(h, n0, flip_test) + ws_bytes_one(h, n - n0, flip_test) + (flip_test ? out_bytes(h, 2 * n) + 512 : 0);
}

extern "C" int udp_hrnet_num_launches(const udp_hrnet* h) { return h ? (int)h->ops.size() : 0; }
extern "C" double udp_hrnet_flops_per_image(const udp_hrnet* h) { return h ? h->flops : 0.0; }

// Resolves buffers / weights of every op for this call and picks kernels + tiles.
static int describe_all(const udp_hrnet* h, const float* in, int n, int flip, char* ws, float* out,
                        std::vector<Launch>& ls) {
  const int B = n * (flip ? 2 : 1);
  const size_t es = esize(h->dtype);
  auto buf = [&](int b) -> char* { return ws + (size_t)h->buf_off[b] * B * es; };
  ls.resize(h->ops.size());
  for (size_t i = 0; i < h->ops.size(); ++i) {
    const udp_conv_op& o = h->ops[i];
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.N = B;
    p.Hin = o.hin;
    p.Win = o.win;
    p.Cin = o.cin;
    p.Hout = o.hout;
    p.Wout = o.wout;
    p.Cout = o.cout;
    p.CoutPad = o.cout_pad;
    p.relu = o.relu;
    p.wfmt = o.wfmt;
    p.wexp = o.wexp;
    p.flip_from = flip ? n : B;
    const bool is_stem = o.kind == UDP_OP_STEM || o.kind == UDP_OP_STEM7;
    p.in_pitch = o.in_pitch ? o.in_pitch : o.cin;
    p.in_coff = o.in_coff;
    p.out_pitch = o.out_pitch ? o.out_pitch : o.cout;
    p.out_coff = o.out_coff;
    p.res_pitch = o.res_pitch ? o.res_pitch : o.cout;
    p.res_coff = o.res_coff;
    p.in = is_stem ? reinterpret_cast<const void*>(in) : buf(o.in_buf);
    p.out_nchw_f32 = o.out_buf == UDP_BUF_OUTPUT;
    p.out = p.out_nchw_f32 ? reinterpret_cast<void*>(out) : o.out_buf == UDP_BUF_NONE ? nullptr : buf(o.out_buf);
    p.res = o.res_buf == UDP_BUF_NONE ? nullptr : buf(o.res_buf);
    p.nup = o.n_up;
    for (int u = 0; u < o.n_up; ++u) {
      p.up[u] = buf(o.up_buf[u]);
      p.up_shift[u] = o.up_shift[u];
    }
    p.nout2 = o.n_out2;
    for (int k = 0; k < o.n_out2; ++k) {
      p.out2[k] = buf(o.out2_buf[k]);
      p.add2[k] = buf(o.add2_buf[k]);
      p.out2_coff[k] = o.out2_coff[k];
      p.out2_pitch[k] = o.out2_pitch[k] ? o.out2_pitch[k] : o.cout;
      p.add2_coff[k] = o.add2_coff[k];
      p.add2_pitch[k] = o.add2_pitch[k] ? o.add2_pitch[k] : o.cout;
    }
    if (is_stem || o.kind == UDP_OP_CONV) {
      p.wgt = h->weights + o.w_off;
      p.bias = reinterpret_cast<const float*>(h->weights + o.b_off);
    }
    if (o.kind == UDP_OP_CONV && o.chain_cout) {     // conv_chain_kernel's use of the addend fields (validate_op: n_up == 0)
      p.wgt2 = h->weights + o.w2_off;
      p.bias2 = reinterpret_cast<const float*>(h->weights + o.b2_off);
      p.up[0] = buf(o.chain_buf);
      p.up_shift[0] = o.chain_cout;
      p.up_shift[1] = o.chain_wexp;
      p.up_shift[2] = o.chain_relu;
    }
    if (o.kind == UDP_OP_BLOCK) {
      p.wgt = h->weights + o.w_off;
      p.bias = reinterpret_cast<const float*>(h->weights + o.b_off);
      p.wgt2 = h->weights + o.w2_off;
      p.bias2 = reinterpret_cast<const float*>(h->weights + o.b2_off);
    } else if (o.kind >= UDP_OP_PSA_POOL) {
      p.wgt = h->weights + o.w_off;
      if (o.kind == UDP_OP_PSA_SCALE) p.res_pitch = o.cin + o.cin / 2;   // fp32 rows {m[C], gbar[C/2]}
    }
    int rc;
    switch (o.kind) {
      case UDP_OP_PSA_POOL:
      case UDP_OP_PSA_MLP:
      case UDP_OP_PSA_SCALE:
      case UDP_OP_PSA_SP: rc = describe_psa(p, h->dtype, o.kind, &ls[i]); break;
      case UDP_OP_BLOCK: rc = describe_block(p, h->dtype, &ls[i]); break;
      case UDP_OP_STEM: rc = describe_stem(p, h->dtype, &ls[i]); break;
      case UDP_OP_STEM7: rc = describe_stem7(p, h->dtype, &ls[i]); break;
      case UDP_OP_FUSE: rc = describe_fuse(p, h->dtype, &ls[i]); break;
      case UDP_OP_MAXPOOL: rc = describe_maxpool(p, h->dtype, &ls[i]); break;
      case UDP_OP_BILINEAR: rc = describe_bilinear(p, h->dtype, &ls[i]); break;
      default:
        if (o.chain_cout) {
          rc = describe_conv_chain(p, &ls[i]);
          break;
        }
        rc = o.group != 0 && getenv("UDP_POSE_NO_GROUPS") == nullptr ? describe_conv_grouped(p, h->dtype, o.ks, o.stride, &ls[i]) : 1;
        if (rc == 1) rc = describe_conv(p, h->dtype, o.ks, o.stride, &ls[i]);
    }
    if (rc) return rc;
  }
  return UDP_OK;
}

// Launch list: runs of groupable convs with the same group id become one conv_mfma_multi launch.
struct Node {
  size_t first;
  int n;
};
static std::vector<Node> make_nodes(const udp_hrnet* h, const std::vector<Launch>& L, bool* grouped) {
  std::vector<Node> nodes;
  *grouped = false;
  for (size_t i = 0; i < L.size();) {
    int n = 1;
    if (L[i].groupable && h->ops[i].group != 0)
      while (n < 4 && i + n < L.size() && L[i + n].groupable / 10 == L[i].groupable / 10 && h->ops[i + n].group == h->ops[i].group) ++n;
    nodes.push_back({i, n});
    *grouped |= n > 1;
    i += n;
  }
  return nodes;
}
static int run_node(const std::vector<Launch>& L, const Node& nd, hipStream_t s) {
  if (nd.n == 1) return run_launch(L[nd.first], s);
  ConvMulti m;
  Launch ml;
  const int rc = describe_multi(&L[nd.first], nd.n, &m, &ml);
  if (rc) return rc;
  void* args[] = {&m};
  UDP_HIP_CHECK(hipLaunchKernel(ml.fn, ml.grid, ml.block, args, ml.lds, s));
  return UDP_OK;
}

// Eager execution.  lanes != 0: ops run on their lane's stream (lane 0 = s) with event edges for
// the cross-lane dependencies the host planner listed, forked from / joined back into s.
// ev != null: serial on s with an event pair around every op (profiling).
static int enqueue_all(udp_hrnet* h, const std::vector<Launch>& L, hipStream_t s, hipEvent_t* ev, int lanes) {
  bool grouped = false;
  const std::vector<Node> nodes = make_nodes(h, L, &grouped);
  if (grouped) {
    // merged launches span lanes: one chain on s, the same launches the graph replays.  Profiling: the
    // event pair of a merged launch sits on its first op; udp_hrnet_profile splits the time over the members
    for (const Node& nd : nodes) {
      if (ev) UDP_HIP_CHECK(hipEventRecord(ev[2 * nd.first], s));
      const int rc = run_node(L, nd, s);
      if (rc) return rc;
      if (ev) UDP_HIP_CHECK(hipEventRecord(ev[2 * nd.first + 1], s));
    }
    return UDP_OK;
  }
  lanes = lanes && h->n_lanes > 1 && !ev && getenv("UDP_POSE_SERIAL") == nullptr;
  hipStream_t ls[UDP_MAX_LANES] = {s, s, s, s};
  if (lanes) {
    UDP_HIP_CHECK(hipEventRecord(h->ev_fork, s));
    for (int l = 1; l < h->n_lanes; ++l) {
      ls[l] = h->lane_stream[l];
      UDP_HIP_CHECK(hipStreamWaitEvent(ls[l], h->ev_fork, 0));
    }
  }
  for (size_t i = 0; i < h->ops.size(); ++i) {
    const udp_conv_op& o = h->ops[i];
    hipStream_t os = ls[o.lane];
    if (lanes)
      for (int k = 0; k < o.n_wait; ++k)
        if (h->ops[o.wait_op[k]].lane != o.lane) UDP_HIP_CHECK(hipStreamWaitEvent(os, h->op_done[o.wait_op[k]], 0));
    if (ev) UDP_HIP_CHECK(hipEventRecord(ev[2 * i], os));
    const int rc = run_launch(L[i], os);
    if (rc) return rc;
    if (ev) UDP_HIP_CHECK(hipEventRecord(ev[2 * i + 1], os));
    if (lanes && h->op_signals[i]) UDP_HIP_CHECK(hipEventRecord(h->op_done[i], os));
  }
  if (lanes)
    for (int l = 1; l < h->n_lanes; ++l) {
      UDP_HIP_CHECK(hipEventRecord(h->ev_join[l], ls[l]));
      UDP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[l], 0));
    }
  return UDP_OK;
}

// hipGraph of the same program: one kernel node per op; edges = previous op of the same lane +
// the planner's cross-lane dependencies, so independent HRNet branches overlap on the GPU.
static int build_graph(udp_hrnet* h, const std::vector<Launch>& L, hipGraph_t* graph, hipGraphExec_t* exec) {
  UDP_HIP_CHECK(hipGraphCreate(graph, 0));
  bool grouped = false;
  const std::vector<Node> nodes = make_nodes(h, L, &grouped);
  std::vector<hipGraphNode_t> node(nodes.size());
  std::vector<int> node_of_op(L.size(), -1);
  int last_on_lane[UDP_MAX_LANES] = {-1, -1, -1, -1};
  // merged launches span lanes: one chain (branch overlap through graph edges bought nothing anyway)
  const bool serial = grouped || getenv("UDP_POSE_SERIAL") != nullptr;
  for (size_t k = 0; k < nodes.size(); ++k) {
    const size_t i = nodes[k].first;
    const udp_conv_op& o = h->ops[i];
    std::vector<hipGraphNode_t> deps;
    if (serial) {
      if (k) deps.push_back(node[k - 1]);
    } else {
      if (last_on_lane[o.lane] >= 0) deps.push_back(node[last_on_lane[o.lane]]);
      for (int w = 0; w < o.n_wait; ++w) deps.push_back(node[node_of_op[o.wait_op[w]]]);
    }
    ConvParams p = L[i].p;
    ConvMulti m;
    Launch ml = L[i];
    void* args[1] = {&p};
    if (nodes[k].n > 1) {
      const int rc = describe_multi(&L[i], nodes[k].n, &m, &ml);
      if (rc) {
        (void)hipGraphDestroy(*graph);
        return rc;
      }
      args[0] = &m;
    }
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.func = const_cast<void*>(ml.fn);
    kp.gridDim = ml.grid;
    kp.blockDim = ml.block;
    kp.sharedMemBytes = ml.lds;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipError_t e = hipGraphAddKernelNode(&node[k], *graph, deps.data(), deps.size(), &kp);
    if (e != hipSuccess) {
      (void)hipGraphDestroy(*graph);
      return fail(UDP_ERR_HIP, "hipGraphAddKernelNode(op %zu): %s", i, hipGetErrorString(e));
    }
    for (int j = 0; j < nodes[k].n; ++j) node_of_op[i + j] = (int)k;
    last_on_lane[o.lane] = (int)k;
  }
  hipError_t e = hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(*graph);
    return fail(UDP_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  return UDP_OK;
}

static int check_forward_args(const char* who, const udp_hrnet* h, const float* in, int n, int flip, const void* ws,
                              size_t ws_bytes, const float* out) {
  if (!h || !in || !ws || !out) return fail(UDP_ERR_ARG, "%s: null pointer", who);
  if (n <= 0) return fail(UDP_ERR_ARG, "%s: n=%d (the reference engine also requires n >= 1)", who, n);
  if (ws_bytes < udp_hrnet_workspace_bytes(h, n, flip))
    return fail(UDP_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, udp_hrnet_workspace_bytes(h, n, flip));
  if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(UDP_ERR_ARG, "%s: workspace not 256-byte aligned", who);
  return UDP_OK;
}

extern "C" int udp_hrnet_profile(udp_hrnet* h, const float* in_nchw, int n, int flip_test, void* workspace,
                                 size_t workspace_bytes, float* heatmaps_nchw, float* ms_per_op, void* stream) {
  int rc = check_forward_args("udp_hrnet_profile", h, in_nchw, n, flip_test, workspace, workspace_bytes, heatmaps_nchw);
  if (rc) return rc;
  if (!ms_per_op) return fail(UDP_ERR_ARG, "udp_hrnet_profile: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  std::vector<Launch> L;
  rc = describe_all(h, in_nchw, n, flip_test ? 1 : 0, reinterpret_cast<char*>(workspace), heatmaps_nchw, L);
  if (rc) return rc;
  const size_t nops = h->ops.size();
  std::vector<hipEvent_t> ev(2 * nops, nullptr);
  for (auto& e : ev) {
    const hipError_t ce = hipEventCreate(&e);
    if (ce != hipSuccess) {
      for (auto& d : ev)
        if (d) (void)hipEventDestroy(d);
      return fail(UDP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(ce));
    }
  }
  rc = enqueue_all(h, L, s, ev.data(), 0);
  hipError_t se = hipStreamSynchronize(s);
  if (!rc && se == hipSuccess) {
    bool grouped = false;
    for (const Node& nd : make_nodes(h, L, &grouped)) {
      float t = -1.f;
      if (hipEventElapsedTime(&t, ev[2 * nd.first], ev[2 * nd.first + 1]) != hipSuccess) t = -1.f;
      // a merged launch: its time is shared by the members in proportion to their FLOPs
      double total = 0.0;
      auto flops = [&](size_t i) { const udp_conv_op& o = h->ops[i]; return (double)o.ks * o.ks * o.cin * o.cout * o.hout * o.wout; };
      for (int j = 0; j < nd.n; ++j) total += flops(nd.first + j);
      for (int j = 0; j < nd.n; ++j) ms_per_op[nd.first + j] = nd.n == 1 || t < 0 ? t : (float)(t * flops(nd.first + j) / total);
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc) return rc;
  if (se != hipSuccess) return fail(UDP_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(se));
  return UDP_OK;
}

static int forward_one(udp_hrnet* h, const float* in_nchw, int n, int flip_test, void* workspace, float* heatmaps_nchw,
                       int use_graph, hipStream_t s) {
  int rc;
  if (use_graph)
    for (size_t k = 0; k < h->graphs.size(); ++k) {
      const GraphEntry g = h->graphs[k];
      if (g.n == n && g.flip == flip_test && g.in == in_nchw && g.ws == workspace && g.out == heatmaps_nchw) {
        if (k + 1 != h->graphs.size()) {   // most recently used last
          h->graphs.erase(h->graphs.begin() + k);
          h->graphs.push_back(g);
        }
        UDP_HIP_CHECK(hipGraphLaunch(g.exec, s));
        return UDP_OK;
      }
    }
  std::vector<Launch> L;
  rc = describe_all(h, in_nchw, n, flip_test, reinterpret_cast<char*>(workspace), heatmaps_nchw, L);
  if (rc) return rc;
  if (!use_graph) return enqueue_all(h, L, s, nullptr, 1);
  // First call for this (shape, buffers): build the graph, then launch it.
  GraphEntry e;
  e.n = n;
  e.flip = flip_test;
  e.in = in_nchw;
  e.ws = workspace;
  e.out = heatmaps_nchw;
  rc = build_graph(h, L, &e.graph, &e.exec);
  if (rc) return rc;
  if (h->graphs.size() >= 8) {
    // evict the least recently used graph; its last replay may still be running (on the caller's stream or on the
    // second lane's) -> drain the device before the executable goes away
    UDP_HIP_CHECK(hipDeviceSynchronize());
    (void)hipGraphExecDestroy(h->graphs[0].exec);
    (void)hipGraphDestroy(h->graphs[0].graph);
    h->graphs.erase(h->graphs.begin());
  }
  h->graphs.push_back(e);
  UDP_HIP_CHECK(hipGraphLaunch(e.exec, s));
  return UDP_OK;
}

extern "C" int udp_hrnet_forward(udp_hrnet* h, const float* in_nchw, int n, int flip_test, void* workspace,
                                 size_t workspace_bytes, float* heatmaps_nchw, int use_graph, void* stream) {
  int rc = check_forward_args("udp_hrnet_forward", h, in_nchw, n, flip_test, workspace, workspace_bytes, heatmaps_nchw);
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  flip_test = flip_test ? 1 : 0;
  if (!use_graph || !split_ok(h, n)) return forward_one(h, in_nchw, n, flip_test, workspace, heatmaps_nchw, use_graph, s);
  // ---- two sub-batch lanes
  if (!h->split_stream) {
    UDP_HIP_CHECK(hipStreamCreateWithFlags(&h->split_stream, hipStreamNonBlocking));
    UDP_HIP_CHECK(hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
    UDP_HIP_CHECK(hipEventCreateWithFlags(&h->split_join, hipEventDisableTiming));
  }
  const int n0 = (n + 1) / 2, n1 = n - n0;
  struct FillHint {                        // tile choice of the lanes' graphs (built on their first call only)
    FillHint() { ws_set_fill_wgs(128); }
    ~FillHint() { ws_set_fill_wgs(512); }
  } fill_hint;
  char* ws0 = reinterpret_cast<char*>(workspace);
  char* ws1 = ws0 + ws_bytes_one(h, n0, flip_test);
  const size_t img_in = (size_t)3 * h->in_h * h->in_w;
  const size_t img_out = out_bytes(h, 1) / sizeof(float);
  hipStream_t s2 = h->split_stream;
  UDP_HIP_CHECK(hipEventRecord(h->split_fork, s));
  UDP_HIP_CHECK(hipStreamWaitEvent(s2, h->split_fork, 0));
  if (!flip_test) {
    rc = forward_one(h, in_nchw, n0, 0, ws0, heatmaps_nchw, 1, s);
    if (!rc) rc = forward_one(h, in_nchw + n0 * img_in, n1, 0, ws1, heatmaps_nchw + n0 * img_out, 1, s2);
  } else {
    // a lane's heat-maps are [its images | their mirrored copies]; the caller's layout is [all images | all mirrored]
    float* t0 = reinterpret_cast<float*>(ws1 + ws_bytes_one(h, n1, 1));
    float* t1 = t0 + 2 * n0 * img_out;
    rc = forward_one(h, in_nchw, n0, 1, ws0, t0, 1, s);
    if (!rc) {
      UDP_HIP_CHECK(hipMemcpyAsync(heatmaps_nchw, t0, n0 * img_out * sizeof(float), hipMemcpyDeviceToDevice, s));
      UDP_HIP_CHECK(hipMemcpyAsync(heatmaps_nchw + n * img_out, t0 + n0 * img_out, n0 * img_out * sizeof(float), hipMemcpyDeviceToDevice, s));
      rc = forward_one(h, in_nchw + n0 * img_in, n1, 1, ws1, t1, 1, s2);
    }
    if (!rc) {
      UDP_HIP_CHECK(hipMemcpyAsync(heatmaps_nchw + n0 * img_out, t1, n1 * img_out * sizeof(float), hipMemcpyDeviceToDevice, s2));
      UDP_HIP_CHECK(hipMemcpyAsync(heatmaps_nchw + (n + n0) * img_out, t1 + n1 * img_out, n1 * img_out * sizeof(float), hipMemcpyDeviceToDevice, s2));
    }
  }
  // join (also after an error in a lane: whatever was enqueued on the second stream is ordered before the caller's next work)
  UDP_HIP_CHECK(hipEventRecord(h->split_join, s2));
  UDP_HIP_CHECK(hipStreamWaitEvent(s, h->split_join, 0));
  return rc;
}

extern "C" int udp_hrnet_destroy(udp_hrnet* h) {
  if (!h) return UDP_OK;
  if (h->split_stream) {
    (void)hipStreamSynchronize(h->split_stream);
    (void)hipStreamDestroy(h->split_stream);
    (void)hipEventDestroy(h->split_fork);
    (void)hipEventDestroy(h->split_join);
  }
  for (auto& g : h->graphs) {
    (void)hipGraphExecDestroy(g.exec);
    (void)hipGraphDestroy(g.graph);
  }
  for (int l = 1; l < UDP_MAX_LANES; ++l) {
    if (h->lane_stream[l]) (void)hipStreamDestroy(h->lane_stream[l]);
    if (h->ev_join[l]) (void)hipEventDestroy(h->ev_join[l]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (auto e : h->op_done)
    if (e) (void)hipEventDestroy(e);
  delete h;
  return UDP_OK;
}

static int conv2d_fused_impl(const udp_conv_op* o, int dtype, int n, const void* in, const void* weights,
                             const float* bias, const void* res, const void* up0, const void* up1,
                             const void* up2, void* out, double* bn_ws, size_t bn_ws_doubles, int* bn_rows, void* stream);

extern "C" int udp_conv2d_fused(const udp_conv_op* o, int dtype, int n, const void* in, const void* weights,
                                const float* bias, const void* res, const void* up0, const void* up1,
                                const void* up2, void* out, void* stream) {
  return conv2d_fused_impl(o, dtype, n, in, weights, bias, res, up0, up1, up2, out, nullptr, 0, nullptr, stream);
}

extern "C" int udp_conv2d_fused_bn(const udp_conv_op* o, int dtype, int n, const void* in, const void* weights,
                                   const float* bias, void* out, double* bn_ws, size_t bn_ws_doubles, int* bn_rows,
                                   void* stream) {
  if (!bn_ws || !bn_rows) return fail(UDP_ERR_ARG, "udp_conv2d_fused_bn: null pointer");
  if (dtype == UDP_F16X2 || !o || o->kind != UDP_OP_CONV || o->out_buf == UDP_BUF_OUTPUT || o->n_up || o->relu ||
      (o->out_pitch && o->out_pitch != o->cout) || o->out_coff)
    return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_fused_bn: a plain fp32 / bf16 NHWC conv without addends, ReLU or channel views");
  return conv2d_fused_impl(o, dtype, n, in, weights, bias, nullptr, nullptr, nullptr, nullptr, out, bn_ws, bn_ws_doubles, bn_rows, stream);
}

static int conv2d_params(const udp_conv_op* o, int dtype, int n, const void* in, const void* weights,
                         const float* bias, const void* res, const void* up0, const void* up1,
                         const void* up2, void* out, ConvParams* pp) {
  ConvParams& p = *pp;
  if (!o || !in || !out) return fail(UDP_ERR_ARG, "udp_conv2d_fused: null pointer");
  if (dtype != UDP_F32 && dtype != UDP_BF16 && dtype != UDP_F16X2) return fail(UDP_ERR_ARG, "udp_conv2d_fused: dtype %d", dtype);
  if (n <= 0) return fail(UDP_ERR_ARG, "udp_conv2d_fused: n=%d", n);
  if (o->kind != UDP_OP_CONV && o->kind != UDP_OP_FUSE) return fail(UDP_ERR_ARG, "udp_conv2d_fused: kind %d", o->kind);
  if (o->kind == UDP_OP_CONV && (!weights || !bias)) return fail(UDP_ERR_ARG, "udp_conv2d_fused: conv needs weights and bias");
  if (o->cout <= 0 || o->cout_pad < o->cout || o->cout_pad % 32 || o->n_up < 0 || o->n_up > 3)
    return fail(UDP_ERR_ARG, "udp_conv2d_fused: bad cout/cout_pad/n_up");
  if (o->kind == UDP_OP_CONV) {
    if ((o->ks != 1 && o->ks != 3) || (o->stride != 1 && o->stride != 2)) return fail(UDP_ERR_ARG, "udp_conv2d_fused: ks/stride");
    const int pad = o->ks / 2;
    if (o->hout != (o->hin + 2 * pad - o->ks) / o->stride + 1 || o->wout != (o->win + 2 * pad - o->ks) / o->stride + 1)
      return fail(UDP_ERR_ARG, "udp_conv2d_fused: output size does not match input/stride");
  }
  const void* ups[3] = {up0, up1, up2};
  memset(&p, 0, sizeof(p));
  p.N = n;
  p.Hin = o->hin;
  p.Win = o->win;
  p.Cin = o->cin;
  p.Hout = o->hout;
  p.Wout = o->wout;
  p.Cout = o->cout;
  p.CoutPad = o->cout_pad;
  p.relu = o->relu;
  if (o->n_out2) return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_fused: second outputs (n_out2) exist in udp_hrnet programs only");
  if (o->chain_cout) return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_fused: chained convs (chain_cout) exist in udp_hrnet programs only");
  p.wfmt = o->wfmt;
  p.wexp = o->wexp;
  p.in_stuff2 = o->in_stuff2 ? 1 : 0;
  p.flip_from = n;
  p.in_pitch = o->in_pitch ? o->in_pitch : o->cin;
  p.in_coff = o->in_coff;
  p.out_pitch = o->out_pitch ? o->out_pitch : o->cout;
  p.out_coff = o->out_coff;
  p.res_pitch = o->res_pitch ? o->res_pitch : o->cout;
  p.res_coff = o->res_coff;
  p.in = in;
  p.out = out;
  p.out_nchw_f32 = o->out_buf == UDP_BUF_OUTPUT;
  p.res = res;
  p.wgt = weights;
  p.bias = bias;
  p.nup = o->n_up;
  for (int u = 0; u < o->n_up; ++u) {
    const int s = o->up_shift[u];
    // (UDP_OP_FUSE also takes same-resolution addends, shift 0: the training step's exchange-unit sums have up to four terms)
    if (!ups[u] || s < (o->kind == UDP_OP_FUSE ? 0 : 1) || s > 5 || (o->hout & ((1 << s) - 1)) || (o->wout & ((1 << s) - 1)))
      return fail(UDP_ERR_ARG, "udp_conv2d_fused: up %d", u);
    p.up[u] = ups[u];
    p.up_shift[u] = s;
  }
  return UDP_OK;
}

static int conv2d_fused_impl(const udp_conv_op* o, int dtype, int n, const void* in, const void* weights,
                             const float* bias, const void* res, const void* up0, const void* up1,
                             const void* up2, void* out, double* bn_ws, size_t bn_ws_doubles, int* bn_rows, void* stream) {
  ConvParams p;
  const int prc = conv2d_params(o, dtype, n, in, weights, bias, res, up0, up1, up2, out, &p);
  if (prc) return prc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Launch l;
  p.bn_ws = bn_ws;
  const int rc = o->kind == UDP_OP_FUSE ? describe_fuse(p, dtype, &l) : describe_conv(p, dtype, o->ks, o->stride, &l);
  if (rc) return rc;
  if (bn_ws) {
    // one partial row of 2*Cout doubles per tile (grid.x); the caller's workspace must hold them
    // (and udp_bn_train_fwd_from_sums takes at most udp_bn_rows_max() rows: the caller falls back to the separate
    // statistics pass on UDP_ERR_WORKSPACE either way)
    if ((size_t)l.grid.x * 2 * o->cout > bn_ws_doubles || l.grid.x > (unsigned)udp_bn_rows_max())
      return fail(UDP_ERR_WORKSPACE, "udp_conv2d_fused_bn: %u partial rows x %d doubles exceed the workspace or the %d-row limit",
                  l.grid.x, 2 * o->cout, udp_bn_rows_max());
    *bn_rows = (int)l.grid.x;
  }
  return run_launch(l, s);
}

// Up to 4 independent plain convs (same dtype and batch; the same-depth convs of different HRNet branches in the
// training step) in as few launches as possible: members whose tile fits the merged instantiation go into ONE
// conv_mfma_multi launch, the others run on their own.  Per-member results are those of udp_conv2d_fused /
// udp_conv2d_fused_bn (the K loop order does not depend on the tiling).
extern "C" int udp_conv2d_fused_group(udp_conv_item* items, int n_items, int dtype, int n, void* stream) {
  if (!items || n_items < 1 || n_items > 4) return fail(UDP_ERR_ARG, "udp_conv2d_fused_group: %d members (1..4)", n_items);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  std::vector<Launch> L((size_t)n_items);
  for (int j = 0; j < n_items; ++j) {
    udp_conv_item& it = items[j];
    const udp_conv_op* o = it.op;
    if (!o || o->kind != UDP_OP_CONV || o->n_up || o->out_buf == UDP_BUF_OUTPUT)
      return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_fused_group: member %d is not a plain NHWC conv", j);
    if (it.bn_ws && (dtype == UDP_F16X2 || o->relu || it.res || (o->out_pitch && o->out_pitch != o->cout) || o->out_coff))
      return fail(UDP_ERR_UNSUPPORTED, "udp_conv2d_fused_group: member %d: BatchNorm sums need a plain fp32 / bf16 conv", j);
    ConvParams p;
    int rc = conv2d_params(o, dtype, n, it.in, it.weights, it.bias, it.res, nullptr, nullptr, nullptr, it.out, &p);
    if (rc) return rc;
    p.bn_ws = it.bn_ws;
    rc = n_items > 1 && getenv("UDP_POSE_NO_GROUPS") == nullptr ? describe_conv_grouped(p, dtype, o->ks, o->stride, &L[j]) : 1;
    if (rc == 1) rc = describe_conv(p, dtype, o->ks, o->stride, &L[j]);
    if (rc) return rc;
    if (it.bn_ws) {
      if ((size_t)L[j].grid.x * 2 * o->cout > it.bn_ws_doubles || L[j].grid.x > (unsigned)udp_bn_rows_max())
        return fail(UDP_ERR_WORKSPACE, "udp_conv2d_fused_group: member %d: %u partial rows x %d doubles exceed the workspace or the %d-row limit",
                    j, L[j].grid.x, 2 * o->cout, udp_bn_rows_max());
      it.bn_rows = (int)L[j].grid.x;
    }
  }
  // merged members first (order inside a merged launch: as given, the caller lists the deepest-K member first)
  std::vector<Launch> G;
  std::vector<int> rest;
  for (int j = 0; j < n_items; ++j) {
    if (L[j].groupable && (G.empty() || L[j].groupable / 10 == G[0].groupable / 10))
      G.push_back(L[j]);
    else
      rest.push_back(j);
  }
  if (G.size() >= 2) {
    ConvMulti m;
    Launch ml;
    const int rc = describe_multi(G.data(), (int)G.size(), &m, &ml);
    if (rc) return rc;
    void* args[] = {&m};
    UDP_HIP_CHECK(hipLaunchKernel(ml.fn, ml.grid, ml.block, args, ml.lds, s));
  } else {
    for (const Launch& l : G) {
      const int rc = run_launch(l, s);
      if (rc) return rc;
    }
  }
  for (int j : rest) {
    const int rc = run_launch(L[j], s);
    if (rc) return rc;
  }
  return UDP_OK;
}
