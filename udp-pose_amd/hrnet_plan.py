"""Host-side compiler: reference YAML (MODEL.EXTRA) + state_dict -> fused op program.

Walks the same graph as PoseHighResolutionNet.forward
(deep_hrnet/lib/models/pose_hrnet.py:436-471; modules :260-273, fuse layers
:189-255, transitions :344-383, Bottleneck :62-100, BasicBlock :29-59) and
emits one ``udp_conv_op`` (include/udp_pose_hip.h) per fused launch:

    out = act( conv(in) + bias [+ res] [+ sum_k nearest_up(up_k)] )

* BatchNorm(eval) is folded into the conv: w' = w * gamma/sqrt(var+eps),
  b' = beta - mean*gamma/sqrt(var+eps), computed in fp64, stored fp32 / bf16.
* An exchange-unit output y_i = ReLU(sum_j f_ij(x_j)) (:267-272) becomes: the
  1x1 convs of the j>i terms write low-resolution temporaries; the last 3x3
  stride-2 conv of each j<i chain adds the running sum (identity term, the
  upsampled temporaries, earlier chains) in its epilogue, the final one applies
  the ReLU; output 0 has no conv term and is one element-wise launch (or, in the
  last stage-4 module, the epilogue of its 1x1 C->4C conv, :213-221).
* Activation buffers are assigned by a linear scan over tensor lifetimes.
"""
import os

import numpy as np
import torch

from . import _lib

BN_EPS = 1e-5


class _T:
    """An activation tensor (NHWC, per image) in SSA form."""
    __slots__ = ("id", "c", "h", "w")

    def __init__(self, i, c, h, w):
        self.id, self.c, self.h, self.w = i, c, h, w

    @property
    def elems(self):
        return self.c * self.h * self.w


class _V:
    """Channels [coff, coff + c) of tensor ``t`` (a channel-slice view: udp_conv_op.in_coff / res_coff + pitch)."""
    __slots__ = ("t", "coff", "c")

    def __init__(self, t, coff, c):
        self.t, self.coff, self.c = t, coff, c

    @property
    def h(self):
        return self.t.h

    @property
    def w(self):
        return self.t.w


def _round_up(x, m):
    return (x + m - 1) // m * m


DTYPES = ("f32", "bf16", "f16x2")
F16X2_LO_SCALE = 2048.0          # csrc/conv.hip kLoScale


def encode_weights(wp, dtype):
    """[taps][cout_pad][cin] fp32 -> bytes in the storage dtype.  f16x2: rows of [cin hi][cin lo] fp16 with
    w ~= hi + lo * 2^-11 (include/udp_pose_hip.h, UDP_F16X2); a folded weight beyond fp16's range is refused."""
    if dtype == "bf16":
        return wp.to(torch.bfloat16).contiguous().view(torch.uint8).numpy().tobytes()
    if dtype == "f16x2":
        if wp.numel() and float(wp.abs().max()) >= 32768.0:
            raise ValueError("f16x2 storage: a BatchNorm-folded weight of magnitude %g exceeds the fp16 range"
                             % float(wp.abs().max()))
        hi = wp.to(torch.float16)
        lo = ((wp - hi.to(torch.float32)) * F16X2_LO_SCALE).to(torch.float16)
        return torch.stack([hi, lo], dim=2).contiguous().view(torch.uint8).numpy().tobytes()
    return wp.contiguous().numpy().tobytes()


def storage_bytes(dtype):
    """Bytes per stored activation / weight element."""
    return 2 if dtype == "bf16" else 4


class HRNetProgram:
    """The compiled program: ops (ctypes array), buffer sizes, packed weights."""

    def __init__(self, state_dict, extra, in_h, in_w, dtype="f32"):
        if dtype not in DTYPES:
            raise ValueError("dtype must be one of %s" % (DTYPES,))
        if in_h % 32 or in_w % 32:
            raise ValueError("input %dx%d must be a multiple of 32" % (in_h, in_w))
        self.sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        self.extra = extra
        self.dtype = dtype
        self.in_h, self.in_w = in_h, in_w
        self.fuse_blocks = os.environ.get("UDP_POSE_NO_BLOCK_FUSION") is None
        self.block_major = dtype in ("bf16", "f16x2")       # modules emitted block by block over all branches
        self.group_convs = self.block_major and os.environ.get("UDP_POSE_NO_GROUPS") is None
        # split-fp16 convs on the weight-stationary kernel (fragment-major weights, udp_conv_op.wfmt = 1)
        # (UDP_POSE_WS=0: the LDS-staged conv_mfma_kernel<H2> instead -- A/B knob)
        self.use_ws = dtype == "f16x2" and os.environ.get("UDP_POSE_WS", "1") != "0"
        self._groups = 0
        self._tensors = []
        self._ops = []          # dicts with _T references
        self._blob = []         # list of (offset, np.ndarray uint8)
        self._blob_size = 0
        self._build()
        self._assign_buffers()

    # ------------------------------------------------------------------ weights
    def _put(self, arr_bytes):
        off = _round_up(self._blob_size, 256)
        self._blob.append((off, arr_bytes))
        self._blob_size = off + len(arr_bytes)
        return off

    def _fold(self, conv, bn):
        w = self.sd[conv + ".weight"].detach().to(torch.float64).cpu()
        cout = w.shape[0]
        bias = self.sd.get(conv + ".bias")
        b = bias.detach().to(torch.float64).cpu() if bias is not None else torch.zeros(cout, dtype=torch.float64)
        if bn is not None:
            g = self.sd[bn + ".weight"].detach().to(torch.float64).cpu()
            beta = self.sd[bn + ".bias"].detach().to(torch.float64).cpu()
            mean = self.sd[bn + ".running_mean"].detach().to(torch.float64).cpu()
            var = self.sd[bn + ".running_var"].detach().to(torch.float64).cpu()
            s = g / torch.sqrt(var + BN_EPS)
            w = w * s[:, None, None, None]
            b = (b - mean) * s + beta
        return w.to(torch.float32), b.to(torch.float32)

    def _pack_conv(self, conv, bn, ws=False, plus=None):
        w, b = self._fold(conv, bn)
        for other in ([plus] if isinstance(plus, tuple) else (plus or [])):      # conv(x_a) + conv'(x_b) = one conv over [x_a | x_b]
            w2, b2 = self._fold(*other)
            if w2.shape[0] != w.shape[0] or w2.shape[2:] != w.shape[2:]:
                raise ValueError("%s + %s: different geometry" % (conv, other[0]))
            w, b = torch.cat([w, w2], dim=1), b + b2
        cout, cin, kh, kw = w.shape
        cout_pad = _round_up(cout, 32)
        wp = torch.zeros(kh * kw, cout_pad, cin, dtype=torch.float32)
        wp[:, :cout] = w.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
        self._wexp = 0
        if ws:
            from .f16x2 import pack_weights_ws
            packed, self._wexp = pack_weights_ws(wp)           # scaled by 2^wexp: any finite magnitude fits
            wbytes = packed.numpy().tobytes()
        else:
            wbytes = encode_weights(wp, self.dtype)
        bp = torch.zeros(cout_pad, dtype=torch.float32)
        bp[:cout] = b
        return self._put(wbytes), self._put(bp.numpy().tobytes()), cout, cin, kh, cout_pad

    # ------------------------------------------------------------------ emission
    def _new(self, c, h, w):
        t = _T(len(self._tensors), c, h, w)
        self._tensors.append(t)
        return t

    def _conv(self, x, conv, bn, stride=1, relu=True, res=None, ups=(), to_output=False, group=0, in_coff=None,
              into=None, plus=None, chain=None):
        """One conv + folded BatchNorm (+ residual / upsampled addends, + ReLU).  ``in_coff``: read the ``cin`` channels
        of ``x`` that start there (a channel-slice view); ``into = (tensor, coff)``: write the output into that slice of
        an existing wider tensor; ``plus = (conv', bn')``: a second conv + BatchNorm of the same geometry whose input
        channels FOLLOW this conv's in ``x`` and whose result is summed in -- one conv over the concatenated channels
        with the weights side by side and the biases added; ``chain = (conv'', bn'')``: a 1x1 conv + BatchNorm + ReLU
        applied to this conv's result in the same launch (udp_conv_op.chain_cout) -- returns ``(out, chained out)``."""
        # (the output conv writes NCHW fp32 from the weight-stationary kernel too; UDP_POSE_HEAD_WS=0: the LDS-staged kernel)
        head_ws = to_output and stride == 1 and os.environ.get("UDP_POSE_HEAD_WS", "1") != "0"
        ws = self.use_ws and (not to_output or head_ws) and int(self.sd[conv + ".weight"].shape[2]) in (1, 3) and stride in (1, 2)
        w_off, b_off, cout, cin, ks, cout_pad = self._pack_conv(conv, bn, ws, plus)
        if isinstance(x, _V):                    # ``x`` / ``res`` may be channel-slice views (_V) of wider tensors
            if cin != x.c:
                raise ValueError("%s: weight expects %d input channels, view has %d" % (conv, cin, x.c))
            x, in_coff = x.t, x.coff
        res_view = res if isinstance(res, _V) else None
        if res_view is not None:
            res = res_view.t
        if cin != x.c and in_coff is None:
            raise ValueError("%s: weight expects %d input channels, tensor has %d" % (conv, cin, x.c))
        pad = ks // 2
        ho = (x.h + 2 * pad - ks) // stride + 1
        wo = (x.w + 2 * pad - ks) // stride + 1
        out = None if to_output else (into[0] if into else self._new(cout, ho, wo))
        op = dict(kind=_lib.UDP_OP_CONV, ks=ks, stride=stride, relu=int(relu), cin=cin, cout=cout,
                  cout_pad=cout_pad, hin=x.h, win=x.w, hout=ho, wout=wo, inp=x, out=out, res=res,
                  ups=list(ups), w_off=w_off, b_off=b_off, name=conv, group=group, wfmt=int(ws), wexp=self._wexp)
        if in_coff is not None:
            if in_coff + cin > x.c:
                raise ValueError("%s: channels %d..%d of a %d-channel tensor" % (conv, in_coff, in_coff + cin, x.c))
            op.update(in_coff=in_coff, in_pitch=x.c)
        if res_view is not None:
            if res_view.c != cout or (res.h, res.w) != (ho, wo):
                raise ValueError("%s: residual view does not match the output" % conv)
            op.update(res_coff=res_view.coff, res_pitch=res.c, res_c=cout)
        if into:
            if (out.h, out.w) != (ho, wo) or into[1] + cout > out.c:
                raise ValueError("%s: output slice does not fit its tensor" % conv)
            op.update(out_coff=into[1], out_pitch=out.c)
        if chain is not None:
            from .f16x2 import pack_weights_ws
            w2, b2 = self._fold(*chain)
            c2 = int(w2.shape[0])
            if not ws or ks != 1 or tuple(w2.shape[1:]) != (cout, 1, 1) or cout_pad != cout or c2 % 32:
                raise ValueError("%s -> %s: not a chain of split-fp16 1x1 convs" % (conv, chain[0]))
            packed, wexp2 = pack_weights_ws(w2.reshape(1, c2, cout))
            z = self._new(c2, ho, wo)
            op.update(chain_out=z, chain_cout=c2, chain_relu=1, chain_wexp=wexp2, w2_off=self._put(packed.numpy().tobytes()),
                      b2_off=self._put(b2.numpy().tobytes()), chain_name=chain[0])
        self._ops.append(op)
        return (out, op["chain_out"]) if chain is not None else out

    def _next_group(self):
        self._groups += 1
        return self._groups

    def _basic_block(self, x, q):
        """BasicBlock.forward (pose_hrnet.py:43-59; PSA variant pose_hrnet_psa.py:37,49)."""
        if self._can_fuse_block(x, q):
            return self._block(x, q)
        t = self._conv(x, q + ".conv1", q + ".bn1")
        if (q + ".deattn.conv_q_right.weight") in self.sd:
            t = self._psa(t, q + ".deattn")
        return self._conv(t, q + ".conv2", q + ".bn2", res=x)

    def _can_fuse_block(self, x, q):
        """The fused BasicBlock kernel (csrc/conv.hip basic_block_c32_kernel): bf16, 32 channels, rows in
        tiles of 8, the staged (8+4) x (W+2) input tile within 640 LDS rows, no attention inside."""
        return (self.dtype == "bf16" and self.fuse_blocks and x.c == 32 and x.h % 8 == 0 and 12 * (x.w + 2) <= 640
                and x.w in (48, 24, 16, 8) and (q + ".deattn.conv_q_right.weight") not in self.sd
                and tuple(self.sd[q + ".conv1.weight"].shape) == (32, 32, 3, 3)
                and tuple(self.sd[q + ".conv2.weight"].shape) == (32, 32, 3, 3))

    def _block(self, x, q):
        w1, b1, _, _, _, _ = self._pack_conv(q + ".conv1", q + ".bn1")
        w2, b2, _, _, _, _ = self._pack_conv(q + ".conv2", q + ".bn2")
        out = self._new(32, x.h, x.w)
        self._ops.append(dict(kind=_lib.UDP_OP_BLOCK, ks=3, stride=1, relu=1, cin=32, cout=32, cout_pad=32, hin=x.h,
                              win=x.w, hout=x.h, wout=x.w, inp=x, out=out, res=None, ups=[], w_off=w1, b_off=b1,
                              w2_off=w2, b2_off=b2, name=q + ".block"))
        return out

    def _psa(self, x, p):
        """PSA_s (PSA.py:190-269) as POOL -> MLP -> SCALE -> 1x1 conv (theta) -> SP; see csrc/psa.hip."""
        sd, C = self.sd, x.c
        if 256 % C or C % 16:
            raise ValueError("%s: PSA needs a channel count that divides 256 (got %d)" % (p, C))
        f32 = lambda k: sd[p + k].detach().to(torch.float32).cpu().reshape(-1)
        block = torch.cat([f32(".conv_q_right.weight"), f32(".conv_v_right.weight"), f32(".conv_up.0.weight"),
                           f32(".conv_up.0.bias"), f32(".conv_up.1.weight"), f32(".conv_up.1.bias"),
                           f32(".conv_up.3.weight"), f32(".conv_up.3.bias"), f32(".conv_q_left.weight")])
        want = C + C // 2 * C + C // 8 * (C // 2) + 3 * (C // 8) + C * (C // 8) + C + C // 2 * C
        if block.numel() != want:
            raise ValueError("%s: PSA parameter shapes do not match planes=%d" % (p, C))
        w_off = self._put(block.numpy().tobytes())
        f = 4 // storage_bytes(self.dtype)                               # fp32 side rows, counted in dtype elements
        base = dict(ks=1, stride=1, relu=0, cout_pad=_round_up(C, 32), hin=x.h, win=x.w, hout=x.h, wout=x.w,
                    res=None, ups=[], w_off=w_off, b_off=0)
        pooled = self._new(2 * C * f, 1, 1)
        self._ops.append(dict(base, kind=_lib.UDP_OP_PSA_POOL, cin=C, cout=C, inp=x, out=pooled, name=p + ".pool"))
        mask = self._new((C + C // 2) * f, 1, 1)
        self._ops.append(dict(base, kind=_lib.UDP_OP_PSA_MLP, cin=C, cout=C, inp=pooled, out=mask, name=p + ".mlp"))
        x1 = self._new(C, x.h, x.w)
        self._ops.append(dict(base, kind=_lib.UDP_OP_PSA_SCALE, cin=C, cout=C, inp=x, res=mask, out=x1,
                              name=p + ".scale"))
        theta = self._conv(x1, p + ".conv_v_left", None, relu=False)
        x2 = self._new(C, x.h, x.w)
        self._ops.append(dict(base, kind=_lib.UDP_OP_PSA_SP, cin=C // 2, cout=C, inp=theta, res=x1, ups=[(mask, 0)],
                              out=x2, name=p + ".sp"))
        return x2

    def _build(self):
        sd = self.sd
        H, W = self.in_h, self.in_w
        # stem conv1 (VALU kernel on the NCHW fp32 input): weights fp32 [ky][kx][ci][cout]
        w, b = self._fold("conv1", "bn1")
        if tuple(w.shape) != (64, 3, 3, 3):
            raise ValueError("conv1.weight must be [64,3,3,3]")
        w_off = self._put(w.permute(2, 3, 1, 0).contiguous().numpy().tobytes())
        b_off = self._put(b.numpy().tobytes())
        x = self._new(64, H // 2, W // 2)
        self._ops.append(dict(kind=_lib.UDP_OP_STEM, ks=3, stride=2, relu=1, cin=3, cout=64, cout_pad=64, hin=H,
                              win=W, hout=H // 2, wout=W // 2, inp=None, out=x, res=None, ups=[], w_off=w_off,
                              b_off=b_off, name="conv1"))
        # The first Bottleneck's projection shortcut (pose_hrnet.py:87-98: out = relu(bn3(conv3(t)) + bn_d(conv_d(x))))
        # is a second 1x1 conv of the same shape as conv3: with t and x side by side in one tensor the two are ONE
        # 1x1 conv over the concatenated channels -- the 4*planes-channel shortcut map is never written nor read back.
        # (UDP_POSE_NO_L1_CONCAT=1: the two convs and the residual read, for A/B)
        p0 = "layer1.0"
        concat = (os.environ.get("UDP_POSE_NO_L1_CONCAT") is None and (p0 + ".downsample.0.weight") in sd
                  and tuple(sd[p0 + ".downsample.0.weight"].shape[1:]) == (64, 1, 1)
                  and tuple(sd[p0 + ".conv3.weight"].shape[1:]) == (64, 1, 1)
                  and tuple(sd[p0 + ".conv2.weight"].shape[:2]) == (64, 64))
        # conv3 (+ shortcut + ReLU) of a Bottleneck and conv1 (+ ReLU) of the next one are both 1x1: chained in one launch
        # (udp_conv_op.chain_cout) the 256-channel map between them is written once and not read back.  Split-fp16 storage
        # on the weight-stationary kernels only.  (UDP_POSE_NO_L1_CHAIN=1: separate launches, for A/B)
        def nxt(k):
            q = "layer1.%d" % (k + 1)
            ok = (self.use_ws and os.environ.get("UDP_POSE_NO_L1_CHAIN") is None and k + 1 < 4
                  and (q + ".downsample.0.weight") not in sd and tuple(sd[q + ".conv1.weight"].shape) == (64, 256, 1, 1)
                  and tuple(sd["layer1.%d.conv3.weight" % k].shape) == (256, 64, 1, 1))
            return (q + ".conv1", q + ".bn1") if ok else None
        a = None                                                     # conv1 output of the block, when chained in
        if concat:
            cat = self._new(128, H // 4, W // 4)                     # [conv2 output t | block input x]
            self._conv(x, "conv2", "bn2", stride=2, into=(cat, 64))
            a = self._conv(cat, p0 + ".conv1", p0 + ".bn1", in_coff=64)
            self._conv(a, p0 + ".conv2", p0 + ".bn2", into=(cat, 0))
            x = self._conv(cat, p0 + ".conv3", p0 + ".bn3", plus=(p0 + ".downsample.0", p0 + ".downsample.1"), chain=nxt(0))
            x, a = x if isinstance(x, tuple) else (x, None)
        else:
            x = self._conv(x, "conv2", "bn2", stride=2)
        for k in range(1 if concat else 0, 4):                       # layer1 (:297, Bottleneck :80-100)
            p = "layer1.%d" % k
            if a is None:
                a = self._conv(x, p + ".conv1", p + ".bn1")
            bt = self._conv(a, p + ".conv2", p + ".bn2")
            r = x
            if (p + ".downsample.0.weight") in sd:
                r = self._conv(x, p + ".downsample.0", p + ".downsample.1", relu=False)
            x = self._conv(bt, p + ".conv3", p + ".bn3", res=r, chain=nxt(k))
            x, a = x if isinstance(x, tuple) else (x, None)
        ys = [x]
        for st in (2, 3, 4):
            cfg = self.extra["STAGE%d" % st]
            if cfg["BLOCK"] != "BASIC" or cfg.get("FUSE_METHOD", "SUM") != "SUM":
                raise ValueError("stage %d: only BASIC blocks with SUM fusion are supported" % st)
            nb = cfg["NUM_BRANCHES"]
            if nb != len(cfg["NUM_BLOCKS"]) or nb != len(cfg["NUM_CHANNELS"]):
                raise ValueError("NUM_BRANCHES(%d) <> NUM_BLOCKS/NUM_CHANNELS" % nb)   # pose_hrnet.py:121-139
            xs = self._transition(ys, "transition%d" % (st - 1), nb)
            for m in range(cfg["NUM_MODULES"]):
                last = (st == 4 and m == cfg["NUM_MODULES"] - 1)
                xs = self._module(xs, "stage%d.%d" % (st, m), cfg["NUM_BLOCKS"], last)
            ys = xs
        self._conv(ys[0], "final_layer", None, relu=False, to_output=True)
        self.out_channels = self._ops[-1]["cout"]

    def _transition(self, ys, name, n_cur):
        xs = []
        for i in range(n_cur):
            q = "%s.%d" % (name, i)
            if i < len(ys):
                if (q + ".0.weight") in self.sd:
                    xs.append(self._conv(ys[i], q + ".0", q + ".1"))
                else:
                    xs.append(ys[i])
            else:
                y = ys[-1]
                for k in range(i + 1 - len(ys)):
                    y = self._conv(y, "%s.%d.0" % (q, k), "%s.%d.1" % (q, k), stride=2)
                xs.append(y)
        return xs

    def _module(self, xs, p, num_blocks, last):
        nb = len(xs)
        xs = list(xs)
        n_out = 1 if last else nb
        # Exchange-unit output i >= 2 sums stride-2 convs of tensors that all live at resolution level i - 1: the
        # intermediates of the chains from branches j < i - 1 and the output of branch i - 1 itself (pose_hrnet.py:
        # 236-255).  Side by side in ONE tensor they are one conv over the concatenated channels (weights side by side,
        # biases added, one fp32 accumulation): cat[i] = (tensor, channel offset of branch i - 1's output in it); the
        # branch's last conv2 and the chains' last intermediates write their slices.  (UDP_POSE_NO_FUSE_CONCAT=1: one
        # conv per term, for A/B)
        cat = {}
        if (self.block_major and len(set(num_blocks[:nb])) == 1 and os.environ.get("UDP_POSE_NO_FUSE_CONCAT") is None):
            for i in range(2, n_out):
                lead = sum(xs[j].c for j in range(i - 1))
                if (lead + xs[i - 1].c) % 8 == 0 and lead % 8 == 0:          # 16-byte aligned slices in both planes
                    cat[i] = (self._new(lead + xs[i - 1].c, xs[i - 1].h, xs[i - 1].w), lead)
        if self.block_major and len(set(num_blocks[:nb])) == 1:
            # block-major order: conv1 of block k of every branch, then conv2 of every branch.  The convs of
            # one such row are independent; those the executor can merge carry a common group id
            for k in range(num_blocks[0]):
                qs = ["%s.branches.%d.%d" % (p, b, k) for b in range(nb)]
                plain = [b for b in range(nb) if not self._can_fuse_block(xs[b], qs[b])
                         and (qs[b] + ".deattn.conv_q_right.weight") not in self.sd]
                gid = (self._next_group(), self._next_group()) if self.group_convs and len(plain) > 1 else (0, 0)
                mids = {}
                order = list(reversed(plain)) if os.environ.get("UDP_POSE_GROUP_FWD") is None else plain
                for b in order:            # deepest-K members first: their workgroups run longest
                    mids[b] = self._conv(xs[b], qs[b] + ".conv1", qs[b] + ".bn1", group=gid[0])
                final = k == num_blocks[0] - 1
                for b in order:
                    dst = cat.get(b + 1) if final else None
                    y = self._conv(mids[b], qs[b] + ".conv2", qs[b] + ".bn2", res=xs[b], group=gid[1], into=dst)
                    xs[b] = _V(dst[0], dst[1], xs[b].c) if dst else y
                for b in range(nb):
                    if b not in plain:
                        xs[b] = self._basic_block(xs[b], qs[b])
                        if final:
                            cat.pop(b + 1, None)        # no slice writer for this branch: one conv per term
        else:
            for b in range(nb):
                for k in range(num_blocks[b]):
                    xs[b] = self._basic_block(xs[b], "%s.branches.%d.%d" % (p, b, k))
        outs = []
        # the 1x1 convs of all j > i terms only read the branch outputs: emitted first, in launch groups of <= 4
        pairs = [(i, j) for i in range(n_out) for j in range(i + 1, nb)]
        if os.environ.get("UDP_POSE_GROUP_FWD") is None:
            pairs.sort(key=lambda ij: (-ij[1], ij[0]))          # deepest-K (lowest-resolution source) first
        up_terms = {i: [] for i in range(n_out)}
        for c0 in range(0, len(pairs), 4):
            chunk = pairs[c0:c0 + 4]
            gid = self._next_group() if self.group_convs and len(chunk) > 1 else 0
            for i, j in chunk:
                q = "%s.fuse_layers.%d.%d" % (p, i, j)
                up_terms[i].append((self._conv(xs[j], q + ".0", q + ".1", relu=False, group=gid), j - i))
        for i in range(n_out):
            ups = up_terms[i]
            if i == 0:
                if last:
                    outs.append(self._conv(xs[0], "%s.fuse_layers.0.0.0" % p, None, ups=ups))
                else:
                    out = self._new(xs[0].c, xs[0].h, xs[0].w)
                    self._ops.append(dict(kind=_lib.UDP_OP_FUSE, ks=1, stride=1, relu=1, cin=xs[0].c, cout=xs[0].c,
                                          cout_pad=_round_up(xs[0].c, 32), hin=xs[0].h, win=xs[0].w, hout=xs[0].h,
                                          wout=xs[0].w, inp=xs[0], out=out, res=None, ups=ups, w_off=0, b_off=0,
                                          name=p + ".fuse0"))
                    outs.append(out)
                continue
            if i in cat:
                ct, _ = cat[i]
                off = 0
                for j in range(i - 1):                  # the chains: their last intermediate lands in its slice of ct
                    t = xs[j]
                    for k in range(i - j - 1):
                        q = "%s.fuse_layers.%d.%d.%d" % (p, i, j, k)
                        t = self._conv(t, q + ".0", q + ".1", stride=2, into=(ct, off) if k == i - j - 2 else None)
                    off += xs[j].c
                names = ["%s.fuse_layers.%d.%d.%d" % (p, i, j, i - j - 1) for j in range(i)]
                outs.append(self._conv(ct, names[0] + ".0", names[0] + ".1", stride=2, relu=True, res=xs[i], ups=ups,
                                       plus=[(n + ".0", n + ".1") for n in names[1:]]))
                continue
            res, t = xs[i], None
            for j in range(i):
                t = xs[j]
                for k in range(i - j):
                    q = "%s.fuse_layers.%d.%d.%d" % (p, i, j, k)
                    if k != i - j - 1:
                        t = self._conv(t, q + ".0", q + ".1", stride=2)
                    else:
                        t = self._conv(t, q + ".0", q + ".1", stride=2, relu=(j == i - 1), res=res, ups=ups)
                        res, ups = t, []
            outs.append(t)
        return outs

    # ------------------------------------------------------------------ lanes + buffers
    def _assign_buffers(self):
        """Linear-scan buffer assignment plus the cross-lane dependency lists.

        Lane = resolution level of the op's output (HRNet branch): ops of different lanes may run
        concurrently, ordered only by (a) producer -> consumer edges and (b) buffer reuse: the new
        writer of a physical buffer waits for the previous tenant's writer and readers.  Same-lane
        predecessors are ordered by the stream itself and are not listed."""
        h4 = self.in_h // 4
        for op in self._ops:
            lvl = 0
            while (h4 >> lvl) > op["hout"] and lvl < _lib.MAX_LANES - 1:
                lvl += 1
            op["lane"] = lvl
        self._ops[0]["lane"] = 0
        producer = {}
        readers = {}

        def reads_of(op):        # every tensor the op reads (add2: addends of its second outputs, udp_conv_op.n_out2)
            return [t for t in [op["inp"], op["res"]] + [u for u, _ in op["ups"]] + [a for a, _ in op.get("add2", [])]
                    if t is not None]

        for idx, op in enumerate(self._ops):
            if op["out"] is not None:
                if op["out"].id in producer:
                    readers.setdefault(op["out"].id, []).append(producer[op["out"].id])   # earlier slice writers
                producer[op["out"].id] = idx
            for t in [t for t, _ in op.get("out2", [])] + ([op["chain_out"]] if op.get("chain_out") is not None else []):
                producer[t.id] = idx
            for t in reads_of(op):
                readers.setdefault(t.id, []).append(idx)
        last_use = {tid: max(r) for tid, r in readers.items()}
        free = {}             # elems -> [(buffer id, previous tenant tensor id)]
        self.buf_elems = []
        phys = {}
        pending = []
        for idx, op in enumerate(self._ops):
            deps = set()
            for t in reads_of(op):
                if t.id in producer:
                    deps.add(producer[t.id])
            out = op["out"]
            if out is not None and out.id in phys:
                deps.add(producer[out.id])      # later slice of a concat buffer: ordered after its other writers
                out = None
            for out in (([out] if out is not None else []) + [t for t, _ in op.get("out2", [])]
                        + ([op["chain_out"]] if op.get("chain_out") is not None else [])):
                pool = free.get(out.elems, [])
                pick = None
                for k in range(len(pool) - 1, -1, -1):
                    b, old = pool[k]
                    hazard = {producer[old]} | set(readers.get(old, []))
                    cross = {d for d in (deps | hazard) if self._ops[d]["lane"] != op["lane"]}
                    if len(cross) <= _lib.MAX_WAIT:
                        pick = k
                        deps |= hazard
                        break
                if pick is not None:
                    phys[out.id] = pool.pop(pick)[0]
                else:
                    phys[out.id] = len(self.buf_elems)
                    self.buf_elems.append(out.elems)
            cross = sorted(d for d in deps if self._ops[d]["lane"] != op["lane"])
            if len(cross) > _lib.MAX_WAIT:
                raise RuntimeError("op %s has %d cross-lane dependencies (max %d)" % (op["name"], len(cross), _lib.MAX_WAIT))
            op["wait"] = cross
            # members of a launch group run concurrently: a buffer one of them reads for the last time
            # must not be handed to a later member of the same group
            g = op.get("group", 0)
            nxt = self._ops[idx + 1].get("group", 0) if idx + 1 < len(self._ops) else 0
            for t in reads_of(op):
                if last_use.get(t.id) == idx and t.id in phys:
                    pending.append((t.elems, phys[t.id], t.id))
                    last_use[t.id] = -1
            if g == 0 or nxt != g:
                for elems, b, tid in pending:
                    free.setdefault(elems, []).append((b, tid))
                pending = []
        self._phys = phys

    # ------------------------------------------------------------------ output
    def ops_array(self):
        arr = (_lib.ConvOp * len(self._ops))()
        for i, op in enumerate(self._ops):
            o = arr[i]
            for f in ("kind", "ks", "stride", "relu", "cin", "cout", "cout_pad", "hin", "win", "hout", "wout",
                      "w_off", "b_off"):
                setattr(o, f, op[f])
            o.in_buf = _lib.UDP_BUF_NONE if op["inp"] is None else self._phys[op["inp"].id]
            o.out_buf = (_lib.UDP_BUF_NONE if op.get("no_out") else _lib.UDP_BUF_OUTPUT) if op["out"] is None else self._phys[op["out"].id]
            o.res_buf = _lib.UDP_BUF_NONE if op["res"] is None else self._phys[op["res"].id]
            for f in ("in_coff", "in_pitch", "out_coff", "out_pitch", "res_coff", "res_pitch", "w2_off", "b2_off", "group", "wfmt", "wexp"):
                setattr(o, f, op.get(f, 0))
            if op.get("chain_out") is not None:
                o.chain_cout, o.chain_relu, o.chain_wexp = op["chain_cout"], op["chain_relu"], op["chain_wexp"]
                o.chain_buf = self._phys[op["chain_out"].id]
            o.n_out2 = len(op.get("out2", []))
            for k, ((t2, c2), (ta, ca)) in enumerate(zip(op.get("out2", []), op.get("add2", []))):
                o.out2_buf[k], o.out2_coff[k], o.out2_pitch[k] = self._phys[t2.id], c2, t2.c
                o.add2_buf[k], o.add2_coff[k], o.add2_pitch[k] = self._phys[ta.id], ca, ta.c
            o.lane = op["lane"]
            o.n_wait = len(op["wait"])
            for k, d in enumerate(op["wait"]):
                o.wait_op[k] = d
            o.n_up = len(op["ups"])
            for u, (t, s) in enumerate(op["ups"]):
                o.up_buf[u] = self._phys[t.id]
                o.up_shift[u] = s
        return arr

    def weight_blob(self):
        blob = np.zeros(_round_up(self._blob_size, 256), dtype=np.uint8)
        for off, b in self._blob:
            blob[off:off + len(b)] = np.frombuffer(b, dtype=np.uint8)
        return blob

    def describe(self):
        return [(op["name"], op["kind"], op["ks"], op["stride"], op["cin"], op["cout"], op["hout"], op["wout"])
                for op in self._ops]

    def macs_per_image(self):
        return sum(op["ks"] ** 2 * op["cin"] * op["cout"] * op["hout"] * op["wout"] * (2 if op["kind"] == _lib.UDP_OP_BLOCK else 1)
                   + op.get("chain_cout", 0) * op["cout"] * op["hout"] * op["wout"]
                   for op in self._ops
                   if op["kind"] in (_lib.UDP_OP_STEM, _lib.UDP_OP_CONV, _lib.UDP_OP_STEM7, _lib.UDP_OP_BLOCK))

    def activation_elems_per_image(self):
        """Layer-wise algorithmic traffic: every op reads its inputs once and writes its output once."""
        n = 0
        for op in self._ops:
            n += op["hin"] * op["win"] * op["cin"] + op["hout"] * op["wout"] * op["cout"]
            if op["res"] is not None:
                n += op["res"].elems // op["res"].c * op.get("res_c", op["res"].c)
            n += sum(t.elems for t, _ in op["ups"])
            if op.get("chain_out") is not None:
                n += op["chain_out"].elems
        return n
