"""Training step on MI355X: the host side of deep_hrnet/lib/core/function.py:38-77.

``HRNetTrainer`` owns the parameters (one flat fp32 buffer in ``named_parameters()`` order, so
the gradient of the whole model is ONE contiguous bucket for the RCCL all-reduce), runs the
train-mode forward of lib/models/pose_hrnet.py:436-471 op by op through the C ABI, keeps the tape
and walks it backwards (``loss.backward()``), then ``udp_adam_step`` (``optimizer.step()``).
torch is used for device memory only; every arithmetic step is a kernel of libudp_pose_hip.so.

Reference contract mirrored: ``JointsMSELoss`` / ``JointsMSELoss_offset`` (lib/core/loss.py),
``get_optimizer`` -> Adam(lr) (lib/utils/utils.py:60-76), ``train()`` (function.py:27-77) as
``train_epoch``; multi-GPU replaces nn.DataParallel's implicit gradient reduction by an explicit
mean all-reduce of the flat gradient (dist.allreduce_mean_).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .synth import hrnet_param_shapes

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _rup(x, m):
    return (x + m - 1) // m * m


def _is_param(k):
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))


class _Act:
    """An NHWC activation [n,h,w,ck] (ck = channels rounded up to 16) and its gradient."""
    __slots__ = ("buf", "n", "h", "w", "c", "ck", "grad", "needs_grad", "bn_rows", "bn_ws")

    def __init__(self, buf, n, h, w, c, ck, needs_grad=True):
        self.buf, self.n, self.h, self.w, self.c, self.ck = buf, n, h, w, c, ck
        self.grad = None
        self.needs_grad = needs_grad
        self.bn_rows = 0            # > 0: the conv that produced it left that many BatchNorm partial rows in bn_ws
        self.bn_ws = None           # the BatchNorm workspace those rows are in


class _Group:
    """Tape entry of a multi-tensor op: its backward runs once every member's gradient is there (the members are
    outputs of independent branches; in reverse tape order all their consumers precede the op)."""
    __slots__ = ("ys",)

    def __init__(self, ys):
        self.ys = ys

    @property
    def grad(self):
        return None if all(y.grad is None for y in self.ys) else True

    @grad.setter
    def grad(self, v):
        for y in self.ys:
            y.grad = v


class HRNetTrainer:
    def __init__(self, cfg, state_dict, device="cuda", dtype="f32", lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        from .model import _get
        self.extra = _get(cfg, "MODEL", "EXTRA")
        self.num_joints = int(_get(cfg, "MODEL", "NUM_JOINTS"))
        self.target_type = _get(cfg, "MODEL", "TARGET_TYPE")
        self.device = torch.device(device)
        self.dtype = dtype
        self._dt = _lib.UDP_BF16 if dtype == "bf16" else _lib.UDP_F32
        self._tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        shapes = hrnet_param_shapes(self.extra, self.num_joints, self.target_type)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        missing = [k for k in shapes if k not in sd and not k.endswith("num_batches_tracked")]
        if missing:
            raise RuntimeError("state_dict lacks %d tensors, e.g. %s" % (len(missing), missing[:3]))
        self._keys = [k for k in shapes if _is_param(k)]
        self._stat_keys = [k for k in shapes if k.endswith("running_mean") or k.endswith("running_var")]
        self._shapes = shapes
        self._off, n = {}, 0
        for k in self._keys + self._stat_keys:
            self._off[k] = n
            n += _rup(int(np.prod(shapes[k])), 4)
        self._n_param = self._off[self._stat_keys[0]] if self._stat_keys else n
        flat = torch.zeros(n, dtype=torch.float32)
        for k in self._keys + self._stat_keys:
            v = sd[k].detach().to(torch.float32).reshape(-1)
            flat[self._off[k]:self._off[k] + v.numel()] = v
        self.flat = flat.to(self.device)                 # parameters, then running statistics
        self.grad = torch.zeros(self._n_param, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)
        # packed conv operands, refreshed from the fp32 master weights every step
        self._convs = {}
        esz = 2 if dtype == "bf16" else 4
        ws_bytes = 0
        for k in self._keys:
            s = shapes[k]
            if len(s) != 4:
                continue
            cout, cin, ks = s[0], s[1], s[2]
            fwd = torch.empty(ks * ks * _rup(cout, 32) * _rup(cin, 16) * esz, dtype=torch.uint8, device=self.device)
            dg = None
            if k != "conv1.weight":
                dg = torch.empty(ks * ks * _rup(cin, 32) * _rup(cout, 16) * esz, dtype=torch.uint8, device=self.device)
            self._convs[k[:-len(".weight")]] = (cout, cin, ks, fwd, dg)
            ws_bytes = max(ws_bytes, _lib.lib().udp_conv2d_wgrad_workspace_bytes(cout, cin, ks))
        self._wgrad_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        # one more per branch slot: the weight gradients of a lock-step level keep their partials until the common reduce
        self._wgrad_wss = [self._wgrad_ws] + [torch.empty(ws_bytes, dtype=torch.uint8, device=self.device) for _ in range(3)]
        descs = (_lib.PackDesc * len(self._convs))()
        for i, (name, (cout, cin, ks, fwd, dg)) in enumerate(self._convs.items()):
            descs[i].w, descs[i].w_fwd = self._p(name + ".weight"), fwd.data_ptr()
            descs[i].w_dgrad = None if dg is None else dg.data_ptr()
            descs[i].cout, descs[i].cin, descs[i].ks = cout, cin, ks
        self._pack_table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
        self._zeros = torch.zeros(1024, dtype=torch.float32, device=self.device)      # zero bias rows
        # one BatchNorm workspace per branch slot: the branches of a module run in lock step (_blocks_lockstep), a
        # conv's epilogue leaves its partial rows in its slot until the multi-tensor BatchNorm call has read them
        bn_c = max([shapes[k][0] for k in self._keys if len(shapes[k]) == 1] + [64])     # widest BatchNorm of this net
        self._bn_wss = [torch.zeros(_lib.lib().udp_bn_workspace_doubles(bn_c), dtype=torch.float64, device=self.device)
                        for _ in range(4)]
        self._bn_ws = self._bn_wss[0]
        self.bn_multi = os.environ.get("UDP_POSE_NO_BN_MULTI") is None           # A/B knob
        self.conv_multi = os.environ.get("UDP_POSE_NO_CONV_MULTI") is None       # A/B knob (merged branch convs)
        self._loss = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._graphs, self._warm, self._coef = {}, set(), None            # train_step_graphed
        self.version = 0                      # bumped by every optimizer step (the owning model's staleness check)
        self._tape = []
        # weight gradients on a side stream beside the backward chain (_on_side); A/B knob UDP_POSE_NO_WGRAD_OVERLAP
        self.overlap_wgrad = os.environ.get("UDP_POSE_NO_WGRAD_OVERLAP") is None and torch.device(self.device).type == "cuda"
        self._side = torch.cuda.Stream(device=self.device) if self.overlap_wgrad else None
        self._side_keep, self._side_busy = [], False
        self.fuse_bn_stats = os.environ.get("UDP_POSE_NO_BN_FUSION") is None     # A/B knob
        self.fuse_sums = os.environ.get("UDP_POSE_NO_SUM_FUSION") is None        # exchange-unit sums: one launch (A/B knob)
        # gradient buckets for the all-reduce (~25 MB each): consecutive parameters of the flat
        # gradient; a bucket is reduced as soon as the backward has written its last gradient, so the exchange
        # of the late layers' gradients runs under the backward of the early ones (DDP's overlap, which the
        # reference gets from RSN/exps/RSN18.coco/train.py:46-48 / nn.DataParallel's reducer)
        self.bucket_elems = 6 * 1024 * 1024
        self._buckets, self._bucket_of = [], {}
        lo, cnt = 0, 0
        for i, k in enumerate(self._keys):
            self._bucket_of[k] = len(self._buckets)
            cnt += 1
            end = self._off[self._keys[i + 1]] if i + 1 < len(self._keys) else self._n_param
            if end - lo >= self.bucket_elems or i + 1 == len(self._keys):
                self._buckets.append((lo, end, cnt))
                lo, cnt = end, 0

    # ------------------------------------------------------------------ parameter views
    def _p(self, key):
        return self.flat.data_ptr() + 4 * self._off[key]

    def _g(self, key):
        return self.grad.data_ptr() + 4 * self._off[key]

    def param(self, key):
        n = int(np.prod(self._shapes[key]))
        return self.flat[self._off[key]:self._off[key] + n].view(*self._shapes[key])

    def grad_of(self, key):
        n = int(np.prod(self._shapes[key]))
        return self.grad[self._off[key]:self._off[key] + n].view(*self._shapes[key])

    def state_dict(self):
        """Reference-format state_dict (fp32, host) -- loadable by the reference module and by model.MODELS."""
        flat = self.flat.cpu()
        out = {}
        for k, s in self._shapes.items():
            if k.endswith("num_batches_tracked"):
                out[k] = torch.tensor(self.step_count, dtype=torch.int64)
            else:
                n = int(np.prod(s))
                out[k] = flat[self._off[k]:self._off[k] + n].view(*s).clone()
        return out

    # ------------------------------------------------------------------ primitive ops
    def _stream(self):
        return _lib.stream_ptr()

    def _on_side(self, fn, *keep):
        """Run the launches of ``fn`` on the side stream, ordered behind everything queued on the current stream so far
        (fork).  The weight gradients go there: nothing in the backward chain reads them -- only the bucket all-reduce /
        Adam, behind ``_join_side`` -- and at 32 images per GPU the chain's kernels (48-1024 workgroups, BatchNorm
        passes) leave most of the chip idle.  ``keep``: tensors the side launches read; they stay referenced until the
        join, so the allocator cannot hand their memory to the main stream's next kernels."""
        if not self.overlap_wgrad:
            fn()
            return
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            fn()
        self._side_keep.extend(keep)
        self._side_busy = True

    def _join_side(self):
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_keep.clear()
            self._side_busy = False

    def _new(self, n, h, w, c, needs_grad=True):
        ck = _rup(c, 16)
        return _Act(torch.empty(n * h * w * ck, dtype=self._tdt, device=self.device), n, h, w, c, ck, needs_grad)

    def _like(self, a):
        return torch.empty_like(a.buf)

    def _conv_op(self, ks, stride, cin, cout, hin, win, hout, wout, nchw=False):
        op = _lib.ConvOp()
        op.kind, op.ks, op.stride, op.relu = _lib.UDP_OP_CONV, ks, stride, 0
        op.cin, op.cout, op.cout_pad = cin, cout, _rup(cout, 32)
        op.hin, op.win, op.hout, op.wout = hin, win, hout, wout
        op.in_buf = op.res_buf = _lib.UDP_BUF_NONE
        op.out_buf = _lib.UDP_BUF_OUTPUT if nchw else 0
        return op

    def _conv(self, x, name, stride=1, bias_key=None, nchw_out=False, bn_stats=False, slot=0):
        L = _lib.lib()
        bn_ws = self._bn_wss[slot]
        cout, cin, ks, wf, wd = self._convs[name]
        if x.c != cin:
            raise ValueError("%s: expects %d input channels, got %d" % (name, cin, x.c))
        pad = ks // 2
        ho, wo = (x.h + 2 * pad - ks) // stride + 1, (x.w + 2 * pad - ks) // stride + 1
        bias = self._p(bias_key) if bias_key else self._zeros.data_ptr()
        if nchw_out:
            y = _Act(torch.empty(x.n, cout, ho, wo, dtype=torch.float32, device=self.device), x.n, ho, wo, cout,
                     _rup(cout, 16))
            op = self._conv_op(ks, stride, x.ck, cout, x.h, x.w, ho, wo, nchw=True)
        else:
            y = self._new(x.n, ho, wo, cout)
            op = self._conv_op(ks, stride, x.ck, y.ck, x.h, x.w, ho, wo)
        if bias_key:                                       # bias row padded to cout_pad
            b = torch.zeros(_rup(cout, 32), dtype=torch.float32, device=self.device)
            b[:cout] = self.param(bias_key)
            bias = b.data_ptr()
            y_keep = b
        else:
            y_keep = None
        if bn_stats and self.fuse_bn_stats and not nchw_out and y.c == y.ck:
            # the BatchNorm that follows reads its statistics from the conv epilogue's partial sums
            rows = C.c_int(0)
            rc = L.udp_conv2d_fused_bn(C.byref(op), self._dt, x.n, x.buf.data_ptr(), wf.data_ptr(), bias,
                                       y.buf.data_ptr(), bn_ws.data_ptr(), bn_ws.numel(), C.byref(rows),
                                       self._stream())
            if rc == -4:       # UDP_ERR_WORKSPACE: more tiles than partial rows fit -> separate statistics pass
                _lib.check(L.udp_conv2d_fused(C.byref(op), self._dt, x.n, x.buf.data_ptr(), wf.data_ptr(), bias, None,
                                              None, None, None, y.buf.data_ptr(), self._stream()))
            else:
                _lib.check(rc)
                y.bn_rows = rows.value
                y.bn_ws = bn_ws
        else:
            _lib.check(L.udp_conv2d_fused(C.byref(op), self._dt, x.n, x.buf.data_ptr(), wf.data_ptr(), bias, None, None,
                                          None, None, y.buf.data_ptr(), self._stream()))

        def backward():
            dy = y.grad                                   # NHWC [n,ho,wo,ck(cout)]

            def wgrad():
                _lib.check(L.udp_conv2d_wgrad(x.buf.data_ptr(), dy.data_ptr(), x.n, x.h, x.w, x.ck, ho, wo, y.ck, ks,
                                              stride, cout, cin, self._dt, self._g(name + ".weight"), 0,
                                              self._wgrad_ws.data_ptr(), self._wgrad_ws.numel(), self._stream()))
                if bias_key:
                    _lib.check(L.udp_bias_grad(dy.data_ptr(), x.n * ho * wo, y.ck, cout, self._g(bias_key), self._dt,
                                               self._stream()))
            self._on_side(wgrad, dy, x.buf)
            if not x.needs_grad:
                return
            src, hh, ww = dy, ho, wo
            stuffed = 0
            if stride == 2:
                # the input gradient of a stride-2 conv = a stride-1 conv over dy on the even grid of a 2x image; the
                # conv kernel reads dy AS that zero-stuffed image (udp_conv_op.in_stuff2) instead of a materialised
                # copy (udp_zero_stuff2: one more launch, 4x the bytes written and read; UDP_POSE_ZERO_STUFF=1 for A/B)
                hh, ww = 2 * ho, 2 * wo
                if os.environ.get("UDP_POSE_ZERO_STUFF"):
                    src = torch.empty(x.n * 4 * ho * wo * y.ck, dtype=self._tdt, device=self.device)
                    _lib.check(L.udp_zero_stuff2(dy.data_ptr(), x.n, ho, wo, y.ck, self._dt, src.data_ptr(), self._stream()))
                else:
                    stuffed = 1
            dop = self._conv_op(ks, 1, y.ck, x.ck, hh, ww, x.h, x.w)
            dop.in_stuff2 = stuffed
            res = x.grad
            if res is None:
                x.grad = self._like(x)
            _lib.check(L.udp_conv2d_fused(C.byref(dop), self._dt, x.n, src.data_ptr(), wd.data_ptr(),
                                          self._zeros.data_ptr(), None if res is None else res.data_ptr(), None, None,
                                          None, x.grad.data_ptr(), self._stream()))
        self._tape.append((y, backward, y_keep, [name + ".weight"] + ([bias_key] if bias_key else [])))
        return y

    def _bn(self, x, name, relu=True, res=None):
        L = _lib.lib()
        m, c = x.n * x.h * x.w, x.ck
        if x.c != x.ck:
            raise ValueError("%s: BatchNorm over %d channels (not a multiple of 16)" % (name, x.c))
        y = self._new(x.n, x.h, x.w, x.c)
        save = torch.empty(2 * c, dtype=torch.float32, device=self.device)
        args = (x.buf.data_ptr(), m, c, self._p(name + ".weight"), self._p(name + ".bias"), BN_EPS, BN_MOMENTUM,
                self._p(name + ".running_mean"), self._p(name + ".running_var"), save.data_ptr(), save.data_ptr() + 4 * c,
                None if res is None else res.buf.data_ptr(), int(relu), y.buf.data_ptr(), self._dt,
                (x.bn_ws if x.bn_rows else self._bn_ws).data_ptr())
        if x.bn_rows:
            _lib.check(L.udp_bn_train_fwd_from_sums(*args, x.bn_rows, self._stream()))
        else:
            _lib.check(L.udp_bn_train_fwd(*args, self._stream()))

        def backward():
            x.grad = self._like(x)
            g_out = None
            if res is not None and res.needs_grad:
                g_out = self._like(y)
            _lib.check(L.udp_bn_train_bwd(x.buf.data_ptr(), y.grad.data_ptr(), y.buf.data_ptr() if relu else None, m, c,
                                          self._p(name + ".weight"), save.data_ptr(), save.data_ptr() + 4 * c,
                                          self._g(name + ".weight"), self._g(name + ".bias"), x.grad.data_ptr(),
                                          None if g_out is None else g_out.data_ptr(), self._dt,
                                          self._bn_ws.data_ptr(), self._stream()))
            if g_out is not None:
                if res.grad is None:
                    res.grad = g_out
                else:
                    _lib.check(L.udp_ew_accumulate(res.grad.data_ptr(), g_out.data_ptr(), res.n, res.h, res.w, res.ck,
                                                   0, 0, 0, self._dt, self._stream()))
        self._tape.append((y, backward, save, [name + ".weight", name + ".bias"]))
        return y

    def _bn_multi(self, xs, names, relu=True, res=None):
        """The BatchNorms `names[b]` over `xs[b]` (one per branch, independent) as ONE multi-tensor call per pass
        (udp_bn_train_fwd_multi / _bwd_multi): results are those of `_bn` per branch, bit for bit."""
        L = _lib.lib()
        nb = len(xs)
        res = res or [None] * nb
        relus = list(relu) if isinstance(relu, (list, tuple)) else [relu] * nb
        ys = [self._new(x.n, x.h, x.w, x.c) for x in xs]
        saves = [torch.empty(2 * x.ck, dtype=torch.float32, device=self.device) for x in xs]
        items = (_lib.BnItem * nb)()
        for b, (x, y, sv) in enumerate(zip(xs, ys, saves)):
            if x.c != x.ck:
                raise ValueError("%s: BatchNorm over %d channels (not a multiple of 16)" % (names[b], x.c))
            it = items[b]
            it.x, it.y, it.m, it.c = x.buf.data_ptr(), y.buf.data_ptr(), x.n * x.h * x.w, x.ck
            it.gamma, it.beta = self._p(names[b] + ".weight"), self._p(names[b] + ".bias")
            it.running_mean, it.running_var = self._p(names[b] + ".running_mean"), self._p(names[b] + ".running_var")
            it.save_mean, it.save_invstd = sv.data_ptr(), sv.data_ptr() + 4 * x.ck
            it.res = None if res[b] is None else res[b].buf.data_ptr()
            it.relu, it.rows = int(relus[b]), x.bn_rows
            it.ws = (x.bn_ws if x.bn_rows else self._bn_wss[b]).data_ptr()
        _lib.check(L.udp_bn_train_fwd_multi(items, nb, BN_EPS, BN_MOMENTUM, self._dt, self._stream()))

        def backward():
            bi = (_lib.BnItem * nb)()
            g_outs = []
            for b, (x, y, sv) in enumerate(zip(xs, ys, saves)):
                x.grad = self._like(x)
                if y.grad is None:                     # a branch output nothing consumed: zero gradient
                    y.grad = torch.zeros_like(y.buf)
                g_out = self._like(y) if res[b] is not None and res[b].needs_grad else None
                g_outs.append(g_out)
                it = bi[b]
                it.x, it.dy, it.m, it.c = x.buf.data_ptr(), y.grad.data_ptr(), x.n * x.h * x.w, x.ck
                it.y_relu = y.buf.data_ptr() if relus[b] else None
                it.gamma = self._p(names[b] + ".weight")
                it.save_mean, it.save_invstd = sv.data_ptr(), sv.data_ptr() + 4 * x.ck
                it.dgamma, it.dbeta = self._g(names[b] + ".weight"), self._g(names[b] + ".bias")
                it.dx = x.grad.data_ptr()
                it.g_out = None if g_out is None else g_out.data_ptr()
                it.ws = self._bn_wss[b].data_ptr()
            _lib.check(L.udp_bn_train_bwd_multi(bi, nb, self._dt, self._stream()))
            for b, g_out in enumerate(g_outs):
                if g_out is None:
                    continue
                r = res[b]
                if r.grad is None:
                    r.grad = g_out
                else:
                    _lib.check(L.udp_ew_accumulate(r.grad.data_ptr(), g_out.data_ptr(), r.n, r.h, r.w, r.ck, 0, 0, 0,
                                                   self._dt, self._stream()))
        keys = [n + s for n in names for s in (".weight", ".bias")]
        self._tape.append((_Group(ys), backward, saves, keys))
        return ys

    def _convs_lockstep(self, xs, names):
        """The stride-1 convs `names[b]` on `xs[b]` (one per branch, independent; each followed by a BatchNorm) through
        udp_conv2d_fused_group: the members whose tile fits the merged kernel run as ONE launch, forward and input
        gradient alike (bf16 storage; fp32 members run one launch each).  Per-member results equal `_conv`'s."""
        L = _lib.lib()
        nb = len(xs)
        if not self.conv_multi:
            return [self._conv(xs[b], names[b], bn_stats=True, slot=b) for b in range(nb)]
        metas = [self._convs[nm] for nm in names]                  # (cout, cin, ks, wf, wd)
        ys, ops = [], []
        for x, (cout, cin, ks, wf, wd) in zip(xs, metas):
            if x.c != cin:
                raise ValueError("conv expects %d input channels, got %d" % (cin, x.c))
            y = self._new(x.n, x.h, x.w, cout)
            ys.append(y)
            ops.append(self._conv_op(ks, 1, x.ck, y.ck, x.h, x.w, x.h, x.w))
        order = list(reversed(range(nb)))                          # deepest-K member first in a merged launch
        items = (_lib.ConvItem * nb)()
        for it, b in zip(items, order):
            x, y, (cout, cin, ks, wf, wd) = xs[b], ys[b], metas[b]
            it.op, it.inp, it.weights = C.addressof(ops[b]), x.buf.data_ptr(), wf.data_ptr()
            it.bias, it.res, it.out = self._zeros.data_ptr(), None, y.buf.data_ptr()
            if self.fuse_bn_stats and y.c == y.ck:
                it.bn_ws, it.bn_ws_doubles = self._bn_wss[b].data_ptr(), self._bn_wss[b].numel()
        rc = L.udp_conv2d_fused_group(items, nb, self._dt, xs[0].n, self._stream())
        if rc == -4:           # UDP_ERR_WORKSPACE (more tiles than partial rows fit): the per-conv path sorts it out
            return [self._conv(xs[b], names[b], bn_stats=True, slot=b) for b in range(nb)]
        _lib.check(rc)
        for it, b in zip(items, order):
            if it.bn_ws:
                ys[b].bn_rows, ys[b].bn_ws = it.bn_rows, self._bn_wss[b]

        def backward():
            gi = (_lib.ConvItem * nb)()
            wi = (_lib.WgradItem * nb)()
            dops, k = [], 0
            for slot, b in enumerate(order):
                x, y, (cout, cin, ks, wf, wd) = xs[b], ys[b], metas[b]
                if y.grad is None:
                    y.grad = torch.zeros_like(y.buf)
                w = wi[slot]                           # partial sums one launch per member, their reduces one launch
                w.x, w.dy, w.dw = x.buf.data_ptr(), y.grad.data_ptr(), self._g(names[b] + ".weight")
                w.workspace, w.workspace_bytes = self._wgrad_wss[slot].data_ptr(), self._wgrad_wss[slot].numel()
                w.n, w.hin, w.win, w.cin_k, w.hout, w.wout, w.cout_k = x.n, x.h, x.w, x.ck, x.h, x.w, y.ck
                w.ks, w.stride, w.cout, w.cin, w.accumulate = ks, 1, cout, cin, 0
            self._on_side(lambda: _lib.check(L.udp_conv2d_wgrad_group(wi, nb, self._dt, self._stream())),
                          *[xs[b].buf for b in order], *[ys[b].grad for b in order])
            for b in order:
                x, y, (cout, cin, ks, wf, wd) = xs[b], ys[b], metas[b]
                if not x.needs_grad:
                    continue
                res = x.grad
                if res is None:
                    x.grad = self._like(x)
                dops.append(self._conv_op(ks, 1, y.ck, x.ck, x.h, x.w, x.h, x.w))
                it = gi[k]
                k += 1
                it.op, it.inp, it.weights = C.addressof(dops[-1]), y.grad.data_ptr(), wd.data_ptr()
                it.bias, it.out = self._zeros.data_ptr(), x.grad.data_ptr()
                it.res = None if res is None else res.data_ptr()
            if k:
                _lib.check(L.udp_conv2d_fused_group(gi, k, self._dt, xs[0].n, self._stream()))
        self._tape.append((_Group(ys), backward, ops, [nm + ".weight" for nm in names]))
        return ys

    def _blocks_lockstep(self, xs, ps):
        """BasicBlock `ps[b]` on branch b, all branches at once: the convs stay one launch each, the two BatchNorms
        of the block run as multi-tensor calls over the branches."""
        nb = len(xs)
        c1 = self._convs_lockstep(xs, [q + ".conv1" for q in ps])
        t = self._bn_multi(c1, [q + ".bn1" for q in ps])
        c2 = self._convs_lockstep(t, [q + ".conv2" for q in ps])
        return self._bn_multi(c2, [q + ".bn2" for q in ps], res=xs)

    def _conv_bn_multi(self, specs):
        """Independent conv + BatchNorm pairs `(x, conv, bn, stride, relu)`: convs one launch each, their
        BatchNorms four at a time as multi-tensor calls."""
        outs = []
        for c0 in range(0, len(specs), 4):
            chunk = specs[c0:c0 + 4]
            if len(chunk) == 1:
                x, cv, bn, st, relu = chunk[0]
                outs.append(self._bn(self._conv(x, cv, stride=st, bn_stats=True), bn, relu=relu))
                continue
            ys = [self._conv(x, cv, stride=st, bn_stats=True, slot=b) for b, (x, cv, _, st, _) in enumerate(chunk)]
            outs += self._bn_multi(ys, [bn for _, _, bn, _, _ in chunk], relu=[r for *_, r in chunk])
        return outs

    def _fuse_lockstep(self, xs, p, last):
        """The fuse layers of a module (pose_hrnet.py:224-272) level by level: every 1x1 (j > i) conv and the k-th
        stride-2 conv of every (i, j < i) chain only depend on the level before."""
        nb = len(xs)
        n_out = 1 if last else nb
        ups, cur = {}, {}
        specs, keys = [], []
        for i in range(n_out):
            for j in range(nb):
                q = "%s.fuse_layers.%d.%d" % (p, i, j)
                if j > i:
                    specs.append((xs[j], q + ".0", q + ".1", 1, False))
                    keys.append(("up", i, j))
                elif j < i:
                    specs.append((xs[j], q + ".0.0", q + ".0.1", 2, i - j != 1))
                    keys.append(("down", i, j))
        for key, t in zip(keys, self._conv_bn_multi(specs)):
            (ups if key[0] == "up" else cur)[key[1:]] = t
        for k in range(1, n_out):
            todo = [(i, j) for (i, j) in cur if i - j > k]
            specs = [(cur[(i, j)], "%s.fuse_layers.%d.%d.%d.0" % (p, i, j, k), "%s.fuse_layers.%d.%d.%d.1" % (p, i, j, k), 2,
                      k != i - j - 1) for i, j in todo]
            for ij, t in zip(todo, self._conv_bn_multi(specs)):
                cur[ij] = t
        outs = []
        for i in range(n_out):
            terms = []
            for j in range(nb):
                if j > i:
                    terms.append((ups[(i, j)], j - i))
                elif j == i:
                    terms.append((self._conv(xs[j], "%s.fuse_layers.%d.%d.0" % (p, i, j)) if last else xs[j], 0))
                else:
                    terms.append((cur[(i, j)], 0))
            outs.append(self._sum_relu(terms))
        return outs

    def _sum_relu(self, terms):
        """y = relu(sum_k up(term_k, shift_k)): HighResolutionModule.forward :266-272."""
        L = _lib.lib()
        t0 = terms[0][0]
        y = self._new(t0.n, t0.h << terms[0][1], t0.w << terms[0][1], t0.c)
        # ONE launch: y = relu(((t0 + t1) + t2) + ...), the terms added in the reference's order (udp_conv2d_fused, UDP_OP_FUSE:
        # in + res + up to three addends, each up-sampled by its shift; shift 0 = same resolution); fp32 tensors: the numbers of
        # the term-by-term launches below, bit for bit
        second = len(terms) > 1 and terms[1][1] == 0           # `res` carries no shift
        rest = terms[2:] if second else terms[1:]
        if self.fuse_sums and terms[0][1] == 0 and len(rest) <= 3:
            op = self._conv_op(1, 1, y.ck, y.ck, y.h, y.w, y.h, y.w)
            op.kind, op.relu, op.n_up = _lib.UDP_OP_FUSE, 1, len(rest)
            for k, (t, s) in enumerate(rest):
                op.up_shift[k] = s
            up_ptrs = [t.buf.data_ptr() for t, _ in rest] + [None] * (3 - len(rest))
            _lib.check(L.udp_conv2d_fused(C.byref(op), self._dt, y.n, terms[0][0].buf.data_ptr(), None, None,
                                          terms[1][0].buf.data_ptr() if second else None, up_ptrs[0], up_ptrs[1], up_ptrs[2],
                                          y.buf.data_ptr(), self._stream()))
        else:
            for k, (t, s) in enumerate(terms):
                _lib.check(L.udp_ew_accumulate(y.buf.data_ptr(), t.buf.data_ptr(), y.n, y.h, y.w, y.ck, s, int(k == 0),
                                               int(k == len(terms) - 1), self._dt, self._stream()))

        def backward():
            g = self._like(y)
            _lib.check(L.udp_relu_bwd(y.grad.data_ptr(), y.buf.data_ptr(), g.data_ptr(), g.numel(), self._dt,
                                      self._stream()))
            handed = not self.fuse_sums
            for t, s in terms:
                if not t.needs_grad:
                    continue
                have = t.grad is not None
                if s == 0 and not have and not handed:
                    # the first same-resolution term without a gradient yet takes g itself instead of a copy of it: every
                    # other term of this sum reads g inside this function, later writers of t.grad come later on the tape
                    t.grad = g
                    handed = True
                    continue
                if not have:
                    t.grad = self._like(t)
                if s == 0:
                    _lib.check(L.udp_ew_accumulate(t.grad.data_ptr(), g.data_ptr(), t.n, t.h, t.w, t.ck, 0, int(not have),
                                                   0, self._dt, self._stream()))
                else:
                    _lib.check(L.udp_upsample_bwd(g.data_ptr(), y.n, y.h, y.w, y.ck, s, t.grad.data_ptr(), int(have),
                                                  self._dt, self._stream()))
        self._tape.append((y, backward, None, []))
        return y

    # ------------------------------------------------------------------ the HRNet graph (pose_hrnet.py)
    def _basic(self, x, p):
        t = self._bn(self._conv(x, p + ".conv1", bn_stats=True), p + ".bn1")
        return self._bn(self._conv(t, p + ".conv2", bn_stats=True), p + ".bn2", res=x)

    def _bottleneck(self, x, p):
        a = self._bn(self._conv(x, p + ".conv1", bn_stats=True), p + ".bn1")
        b = self._bn(self._conv(a, p + ".conv2", bn_stats=True), p + ".bn2")
        r = x
        if (p + ".downsample.0") in self._convs:
            r = self._bn(self._conv(x, p + ".downsample.0", bn_stats=True), p + ".downsample.1", relu=False)
        return self._bn(self._conv(b, p + ".conv3", bn_stats=True), p + ".bn3", res=r)

    def _module(self, xs, p, num_blocks, last):
        nb = len(xs)
        xs = list(xs)
        if self.bn_multi and 1 < nb <= 4 and len(set(num_blocks[:nb])) == 1:
            for k in range(num_blocks[0]):
                xs = self._blocks_lockstep(xs, ["%s.branches.%d.%d" % (p, b, k) for b in range(nb)])
        else:
            for b in range(nb):
                for k in range(num_blocks[b]):
                    xs[b] = self._basic(xs[b], "%s.branches.%d.%d" % (p, b, k))
        if self.bn_multi and nb > 1:
            return self._fuse_lockstep(xs, p, last)
        outs = []
        for i in range(1 if last else nb):
            terms = []
            for j in range(nb):
                q = "%s.fuse_layers.%d.%d" % (p, i, j)
                if j > i:
                    terms.append((self._bn(self._conv(xs[j], q + ".0", bn_stats=True), q + ".1", relu=False), j - i))
                elif j == i:
                    terms.append((self._conv(xs[j], q + ".0") if last else xs[j], 0))
                else:
                    t = xs[j]
                    for k in range(i - j):
                        t = self._bn(self._conv(t, "%s.%d.0" % (q, k), stride=2, bn_stats=True), "%s.%d.1" % (q, k),
                                     relu=(k != i - j - 1))
                    terms.append((t, 0))
            outs.append(self._sum_relu(terms))
        return outs

    def _transition(self, ys, name, n_cur):
        xs = []
        for i in range(n_cur):
            q = "%s.%d" % (name, i)
            if i < len(ys):
                xs.append(self._bn(self._conv(ys[i], q + ".0", bn_stats=True), q + ".1") if (q + ".0") in self._convs else ys[i])
            else:
                y = ys[-1]
                for k in range(i + 1 - len(ys)):
                    y = self._bn(self._conv(y, "%s.%d.0" % (q, k), stride=2, bn_stats=True), "%s.%d.1" % (q, k))
                xs.append(y)
        return xs

    def forward(self, x):
        """Train-mode forward.  x: fp32 [N,3,H,W] on the device.  Returns fp32 [N,C,H/4,W/4]."""
        L = _lib.lib()
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or not x.is_contiguous():
            raise ValueError("expected contiguous fp32 [N,3,H,W]")
        n, _, h, w = x.shape
        if h % 32 or w % 32:
            raise ValueError("input %dx%d must be a multiple of 32" % (h, w))
        self._tape = []
        _lib.check(L.udp_pack_conv_weights_batch(self._pack_table.data_ptr(), len(self._convs), self._dt, self._stream()))
        a = self._new(n, h, w, 3, needs_grad=False)
        _lib.check(L.udp_nchw_to_nhwc(x.data_ptr(), n, 3, h, w, a.ck, a.buf.data_ptr(), self._dt, self._stream()))
        a = self._bn(self._conv(a, "conv1", stride=2, bn_stats=True), "bn1")
        a = self._bn(self._conv(a, "conv2", stride=2, bn_stats=True), "bn2")
        for k in range(4):
            a = self._bottleneck(a, "layer1.%d" % k)
        ys = [a]
        for st in (2, 3, 4):
            cfg = self.extra["STAGE%d" % st]
            xs = self._transition(ys, "transition%d" % (st - 1), cfg["NUM_BRANCHES"])
            for mi in range(cfg["NUM_MODULES"]):
                xs = self._module(xs, "stage%d.%d" % (st, mi), cfg["NUM_BLOCKS"],
                                  st == 4 and mi == cfg["NUM_MODULES"] - 1)
            ys = xs
        self._out = self._conv(ys[0], "final_layer", bias_key="final_layer.bias", nchw_out=True)
        return self._out.buf

    def _backward_walk(self, dheat, pre=None):
        """The tape in reverse.  Yields the list of gradient buckets a backward node has just completed (the node wrote
        their last gradient), and finally the buckets no node completed (a bucket with an unused parameter: its gradient
        stays as is).  ``pre()`` is called before anything is launched after a yield (segmented graph capture opens its
        next segment there).  Every parameter gradient is overwritten (implicit zero_grad)."""
        L = _lib.lib()
        if pre:
            pre()
        o = self._out
        g = torch.empty(o.n * o.h * o.w * o.ck, dtype=self._tdt, device=self.device)
        _lib.check(L.udp_nchw_to_nhwc(dheat.data_ptr(), o.n, o.c, o.h, o.w, o.ck, g.data_ptr(), self._dt, self._stream()))
        o.grad = g
        left = [c for _, _, c in self._buckets]
        for y, bwd, _, keys in reversed(self._tape):
            done = []
            if y.grad is not None:
                if pre:
                    pre()
                bwd()
                for k in keys:
                    b = self._bucket_of[k]
                    left[b] -= 1
                    if left[b] == 0:
                        done.append(b)
            y.grad = None
            if done:
                self._join_side()             # the bucket's weight gradients were written on the side stream
                yield done
        self._tape = []
        self._join_side()
        rest = [b for b, n_left in enumerate(left) if n_left > 0]
        if rest:
            yield rest

    def backward(self, dheat, world_size=1):
        """dheat: fp32 [N,C,h,w] = d loss / d heat-maps.  Fills self.grad (zero_grad is implicit: every
        parameter gradient is overwritten).  ``world_size > 1``: every gradient bucket is SUM-all-reduced
        (asynchronously, on the communication stream of torch.distributed) as soon as its last gradient has been
        written, overlapping the rest of the backward; on return all buckets are reduced."""
        works = []
        self.reduce_order = []                # bucket indices in the order their all-reduce was issued
        for done in self._backward_walk(dheat):
            if world_size > 1:
                works += [self._reduce_bucket(b) for b in done]
        for w in works:
            w.wait()

    def _reduce_bucket(self, b):
        from .dist import allreduce_sum_async
        lo, hi, _ = self._buckets[b]
        self.reduce_order.append(b)
        if os.environ.get("UDP_POSE_FAKE_ALLREDUCE"):        # timing diagnosis only (tools/bench_train.py): no exchange
            class _Done:
                def wait(self):
                    return True
            return _Done()
        return allreduce_sum_async(self.grad[lo:hi])

    def loss_and_grad(self, heat, target, target_weight):
        """criterion(output, target, target_weight) (loss.py:15-76) and its gradient w.r.t. output."""
        b, c = heat.shape[:2]
        off = self.target_type == "offset"
        j = c // 3 if off else c
        d = torch.empty_like(heat)
        _lib.check(_lib.lib().udp_mse_loss(heat.data_ptr(), target.data_ptr(), target_weight.data_ptr(), b, j,
                                           heat.shape[2] * heat.shape[3], int(off), self._loss.data_ptr(), d.data_ptr(),
                                           self._stream()))
        self._loss_grad = d
        return self._loss, d

    def adam_step(self, grad_scale=1.0):
        """optimizer.step(): torch.optim.Adam(lr) on the flat parameter buffer (utils.py:70-74).
        ``grad_scale``: 1/world_size after the SUM all-reduce of the gradient."""
        self.step_count += 1
        self.version += 1                                   # owners compare it with the version they last read
        _lib.check(_lib.lib().udp_adam_step(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                            self.exp_avg_sq.data_ptr(), self._n_param, self.lr, self.betas[0],
                                            self.betas[1], self.eps, self.step_count, float(grad_scale),
                                            self._stream()))

    def _adam_step_dev(self, grad_scale=1.0):
        _lib.check(_lib.lib().udp_adam_step_dev(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                                self.exp_avg_sq.data_ptr(), self._n_param, self.betas[0], self.betas[1],
                                                self.eps, self._coef.data_ptr(), float(grad_scale), self._stream()))

    def train_step_graphed(self, x, target, target_weight, world_size=1):
        """train_step with the whole step (weight packing, forward, criterion, backward, Adam: ~1900
        launches issued from the Python tape) captured ONCE per input shape as a hipGraph (torch.cuda.graph: stream
        capture of our launches on torch's capture stream, activations from the graph's private pool) and replayed:
        the host then costs one graph launch per step instead of one ctypes call per kernel.  The first step of a
        shape runs eagerly (lazy one-time initialisation must not be captured), the second captures and replays.
        Arithmetic and results are those of train_step, bit for bit; the learning rate may change between steps
        (it only enters through Adam's two scalars, uploaded before every replay), betas / eps are part of the key.
        At most two shapes stay captured (a full and a ragged last batch).

        ``world_size > 1`` (one process per GPU, config 3): the step is captured in SEGMENTS that end where a gradient
        bucket receives its last gradient -- forward + criterion + the backward of the last layers; the backward
        between two bucket boundaries; ...; and Adam on its own.  A replayed step is: replay segment 0, issue the
        asynchronous SUM all-reduce of its bucket(s) from the host (torch.distributed orders it behind the segment on
        its communication stream: RCCL over xGMI, or gloo), replay segment 1 at once -- the exchange runs under it, as
        DistributedDataParallel's reducer overlaps the reference's backward -- and so on; the work handles are waited
        for (stream order, no host block under RCCL) before the Adam graph with grad_scale 1 / world_size.  All
        segments share one memory pool and are always replayed in capture order.  Same arithmetic, same bucket order
        and the same results as train_step(world_size), bit for bit."""
        key = (tuple(x.shape), tuple(target.shape), tuple(target_weight.shape), self.betas, self.eps, int(world_size))
        ent = self._graphs.get(key)
        if ent is None:
            if key not in self._warm:                                # first step of this shape: eager
                self._warm.add(key)
                return self.train_step(x, target, target_weight, world_size=world_size)
            if self._coef is None:
                self._coef = torch.zeros(2, dtype=torch.float32, device=self.device)
                # pinned staging slots for the asynchronous upload of Adam's two scalars: the host runs steps ahead
                # of the GPU, a slot is rewritten only after the copy that read it has completed (its event)
                self._coef_host = [torch.zeros(2, dtype=torch.float32).pin_memory() for _ in range(4)]
                self._coef_done = [None] * 4
            gx, gt, gw = x.clone(), target.contiguous().clone(), target_weight.contiguous().clone()
            torch.cuda.synchronize()             # the last replay of an evicted graph may still be running
            while len(self._graphs) >= 2:
                self._graphs.pop(next(iter(self._graphs)))
            if world_size == 1:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    heat = self.forward(gx)
                    self.loss_and_grad(heat, gt, gw)
                    self.backward(self._loss_grad)
                    self._adam_step_dev()
                segs = [(graph, [])]
            else:
                segs = self._capture_segments(gx, gt, gw, 1.0 / world_size)
            ent = self._graphs[key] = (segs, gx, gt, gw)
        segs, gx, gt, gw = ent
        gx.copy_(x, non_blocking=True)
        gt.copy_(target, non_blocking=True)
        gw.copy_(target_weight, non_blocking=True)
        self.step_count += 1
        slot = self.step_count % 4
        if self._coef_done[slot] is not None:
            self._coef_done[slot].synchronize()
        _lib.check(_lib.lib().udp_adam_coefficients(self.lr, self.betas[0], self.betas[1], self.step_count,
                                                    self._coef_host[slot].data_ptr()))
        self._coef.copy_(self._coef_host[slot], non_blocking=True)
        self._coef_done[slot] = torch.cuda.Event()
        self._coef_done[slot].record()
        works = []
        self.reduce_order = []
        for graph, buckets in segs:
            if graph is None:                                        # every bucket is reduced: Adam may read the gradient
                for w in works:
                    w.wait()
                continue
            graph.replay()
            works += [self._reduce_bucket(b) for b in buckets]
        self.version += 1
        return self._loss

    def _capture_segments(self, gx, gt, gw, grad_scale):
        """The step as a list of (graph, buckets to all-reduce after it) in replay order, a (None, []) marker where
        the outstanding all-reduces must be waited for, and the Adam graph last."""
        pool = torch.cuda.graph_pool_handle()
        segs = []
        cur = []                                    # [graph, context] of the segment being captured, or empty

        def begin():
            if not cur:
                g = torch.cuda.CUDAGraph()
                ctx = torch.cuda.graph(g, pool=pool)
                ctx.__enter__()
                cur[:] = [g, ctx]

        def end(buckets):
            if cur:
                cur[1].__exit__(None, None, None)
                segs.append((cur[0], list(buckets)))
                cur[:] = []
            elif buckets:                           # buckets nothing was launched for since the last boundary
                segs[-1] = (segs[-1][0], segs[-1][1] + list(buckets))

        try:
            begin()
            heat = self.forward(gx)
            self.loss_and_grad(heat, gt, gw)
            for done in self._backward_walk(self._loss_grad, pre=begin):
                end(done)
            end([])                                 # nodes behind the last bucket boundary (they write no parameter gradient)
        except BaseException:
            if cur:
                cur[1].__exit__(None, None, None)
            raise
        segs.append((None, []))
        adam = torch.cuda.CUDAGraph()
        with torch.cuda.graph(adam, pool=pool):
            self._adam_step_dev(grad_scale)
        segs.append((adam, []))
        return segs

    def train_step(self, x, target, target_weight, world_size=1):
        """function.py:46-76 for one batch.  Returns the loss tensor fp64 [2] = (L_hm, L_offset) on device."""
        heat = self.forward(x)
        loss, d = self.loss_and_grad(heat, target.contiguous(), target_weight.contiguous())
        self.backward(d, world_size)                       # incl. the one exchange step, bucket by bucket
        self.adam_step(1.0 / world_size)                   # mean over ranks = grad_scale
        return loss
