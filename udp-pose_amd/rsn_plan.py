"""Host compiler for RSN-18 (RSN/exps/RSN18.coco/network.py) -> the same fused op program.

Graph restated from the reference: ResNet_top :125-137 (7x7 s2 conv+BN+ReLU, 3x3 s2 max-pool),
four levels of two RSN Bottlenecks :49-122, top-down Upsample_units :202-255 (only what
``outputs[-1][-1]`` needs: u_skip, bilinear align-corners up + up_conv, res_conv1/2 of the finest
level), single stage.

RSN's bottleneck splits a 1x1 conv's output into four groups of ``in_planes*26//64`` channels and
concatenates four branch outputs (:102-114).  Here both are **views**: each group is padded to a
multiple of 16 channels (26 -> 32, 52 -> 64, 104 -> 112, 208), the 1x1 conv's weight rows / the
closing 1x1 conv's weight columns are scattered to the padded positions with zeros in between, the
3x3 convs read / write channel slices in place, and the pad channels stay exactly 0 (zero weights,
zero bias, ReLU).  The ``a + b`` that precedes most 3x3 convs is one element-wise launch.
"""
import torch

import os

from . import _lib
from .hrnet_plan import HRNetProgram, _round_up, encode_weights


class RSNProgram(HRNetProgram):
    def __init__(self, state_dict, in_h, in_w, dtype="f32", chl_num=256):
        self.chl_num = chl_num
        super().__init__(state_dict, {}, in_h, in_w, dtype)

    # ---- weights -------------------------------------------------------------------------------
    def _fold_cbr(self, name):
        """conv (with bias) + BatchNorm(eval) folded in fp64 -> fp32 weight [cout,cin,k,k], bias [cout]."""
        sd = self.sd
        w = sd[name + ".conv.weight"].detach().to(torch.float64).cpu()
        b = sd[name + ".conv.bias"].detach().to(torch.float64).cpu()
        g = sd[name + ".bn.weight"].detach().to(torch.float64).cpu()
        beta = sd[name + ".bn.bias"].detach().to(torch.float64).cpu()
        mean = sd[name + ".bn.running_mean"].detach().to(torch.float64).cpu()
        var = sd[name + ".bn.running_var"].detach().to(torch.float64).cpu()
        s = g / torch.sqrt(var + 1e-5)
        return (w * s[:, None, None, None]).to(torch.float32), ((b - mean) * s + beta).to(torch.float32)

    def _pack(self, w, b, out_map=None, in_map=None, cout_t=None, cin_t=None, ws=False):
        """Pack to [k*k][cout_pad][cin_t]; out_map / in_map scatter real channels to padded positions.  ``ws``: the
        fragment-major split-fp16 layout of the weight-stationary kernels (udp_conv_op.wfmt = 1), as HRNetProgram packs
        its convs."""
        cout, cin, kh, kw = w.shape
        cout_t = cout_t or cout
        cin_t = cin_t or cin
        out_map = out_map if out_map is not None else list(range(cout))
        in_map = in_map if in_map is not None else list(range(cin))
        cout_pad = _round_up(cout_t, 32)
        wp = torch.zeros(kh * kw, cout_pad, cin_t, dtype=torch.float32)
        oi = torch.tensor(out_map)
        ii = torch.tensor(in_map)
        wp[:, oi[:, None], ii[None, :]] = w.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
        bp = torch.zeros(cout_pad, dtype=torch.float32)
        bp[oi] = b
        self._wexp = 0
        if ws:
            from .f16x2 import pack_weights_ws
            packed, self._wexp = pack_weights_ws(wp)
            wbytes = packed.numpy().tobytes()
        else:
            wbytes = encode_weights(wp, self.dtype)
        return self._put(wbytes), self._put(bp.numpy().tobytes()), cout_t, cin_t, kh, cout_pad

    # ---- emission ------------------------------------------------------------------------------
    def _op(self, kind, x, out, name, ks=1, stride=1, relu=0, cin=None, cout=None, cout_pad=None, res=None,
            w_off=0, b_off=0, in_coff=0, out_coff=0, res_coff=0, hout=None, wout=None, wfmt=0, wexp=0,
            out2=(), add2=(), no_out=False):
        cin = cin if cin is not None else x.c
        cout = cout if cout is not None else out.c
        self._ops.append(dict(kind=kind, ks=ks, stride=stride, relu=int(relu), cin=cin, cout=cout,
                              cout_pad=cout_pad or _round_up(cout, 32), hin=x.h if x is not None else self.in_h,
                              win=x.w if x is not None else self.in_w, hout=hout or (out.h if out is not None else 0),
                              wout=wout or (out.w if out is not None else 0), inp=x, out=out, res=res, ups=[],
                              w_off=w_off, b_off=b_off, name=name, in_coff=in_coff,
                              in_pitch=x.c if x is not None else 0, out_coff=out_coff,
                              out_pitch=out.c if out is not None else 0, res_coff=res_coff,
                              res_pitch=res.c if res is not None else 0, wfmt=wfmt, wexp=wexp,
                              out2=list(out2), add2=list(add2), no_out=no_out))

    def _conv_v(self, x, name, out=None, ks=None, stride=1, relu=True, res=None, in_coff=0, cin_view=None,
                out_coff=0, out_map=None, in_map=None, cout_t=None, cin_t=None, to_output=False, sums=(), keep=True):
        """``sums``: [(tensor, channel offset)] addends -- for each one the conv also writes (its stored output + that
        slice) into a new tensor (udp_conv_op.n_out2) and returns ``(out, [sum tensors])``; ``keep=False``: nothing reads
        the plain output, only the sums are stored."""
        w, b = self._fold_cbr(name)
        # split-fp16 3x3 / 1x1 convs on the weight-stationary kernels (UDP_POSE_RSN_WS=0: the LDS-staged kernel, A/B)
        # (the 3x3 head conv writes the NCHW fp32 output from the weight-stationary kernel too; UDP_POSE_HEAD_WS=0: LDS-staged)
        head_ws = to_output and stride == 1 and os.environ.get("UDP_POSE_HEAD_WS", "1") != "0"
        ws = (self.use_ws and (not to_output or head_ws) and int(w.shape[2]) in (1, 3) and stride in (1, 2)
              and os.environ.get("UDP_POSE_RSN_WS", "1") != "0")
        w_off, b_off, cout, cin, k, cout_pad = self._pack(w, b, out_map, in_map, cout_t, cin_t, ws=ws)
        if cin_view is not None and cin_view != cin:
            raise ValueError("%s: view has %d channels, weights expect %d" % (name, cin_view, cin))
        pad = k // 2
        ho = (x.h + 2 * pad - k) // stride + 1
        wo = (x.w + 2 * pad - k) // stride + 1
        if sums and not ws:
            raise ValueError("%s: second outputs need the weight-stationary split-fp16 conv" % name)
        if out is None and not to_output and keep:
            out = self._new(cout, ho, wo)
        s2 = [self._new(cout, ho, wo) for _ in sums]
        self._op(_lib.UDP_OP_CONV, x, out, name, ks=k, stride=stride, relu=relu, cin=cin, cout=cout, cout_pad=cout_pad,
                 res=res, w_off=w_off, b_off=b_off, in_coff=in_coff, out_coff=out_coff, hout=ho, wout=wo,
                 wfmt=int(ws), wexp=self._wexp, out2=[(t, 0) for t in s2], add2=list(sums), no_out=not keep)
        return (out, s2) if sums else out

    def _add(self, a, a_coff, b, b_coff, c, name):
        """out[c channels] = a[a_coff:a_coff+c] + b[b_coff:b_coff+c] (no activation)."""
        out = self._new(c, a.h, a.w)
        self._op(_lib.UDP_OP_FUSE, a, out, name, cin=c, cout=c, res=b, in_coff=a_coff, res_coff=b_coff)
        return out

    def _bottleneck(self, x, p, planes, stride):
        sd = self.sd
        in_planes = sd[p + ".conv_bn_relu1.conv.weight"].shape[1]
        bch = sd[p + ".conv_bn_relu1.conv.weight"].shape[0] // 4
        bp = _round_up(bch, 16)
        scatter = [k * bp + j for k in range(4) for j in range(bch)]       # real channel -> padded position
        if in_planes != x.c:
            raise ValueError("%s expects %d input channels, got %d" % (p, in_planes, x.c))
        s = self._conv_v(x, p + ".conv_bn_relu1", stride=stride, out_map=scatter, cout_t=4 * bp)
        cat = self._new(4 * bp, s.h, s.w)
        pad_in = list(range(bch))

        def c3(src, src_coff, name, out=None, out_coff=0, sums=(), keep=True):
            return self._conv_v(src, p + ".conv_bn_relu" + name, out=out, in_coff=src_coff, out_coff=out_coff,
                                out_map=pad_in, in_map=pad_in, cout_t=bp, cin_t=bp, sums=sums, keep=keep)

        if self.use_ws and os.environ.get("UDP_POSE_RSN_WS", "1") != "0" and os.environ.get("UDP_POSE_RSN_FUSE_ADDS", "1") != "0":
            # split-fp16: the six element-wise sums between the 3x3 convs (network.py:102-114) ride in the epilogue of the
            # conv that produces their second operand (second outputs, udp_conv_op.n_out2) instead of being launches of
            # their own -- 48 launches and a third of their traffic per forward; same numbers as the separate sums
            _, (t21,) = c3(s, 0, "2_1_1", out=cat, out_coff=0, sums=[(s, bp)])           # cat[0]; + spx[1]
            o21, (t31,) = c3(t21, 0, "2_2_1", sums=[(s, 2 * bp)])                        # out_2_1; + spx[2]
            c3(o21, 0, "2_2_2", out=cat, out_coff=bp)                                    # out_2_2 -> cat[1]
            _, (t32, t41) = c3(t31, 0, "2_3_1", sums=[(cat, bp), (s, 3 * bp)], keep=False)   # out_3_1 (+ cat[1]; + spx[3])
            o32 = c3(t32, 0, "2_3_2")                                                    # out_3_2
            c3(o32, 0, "2_3_3", out=cat, out_coff=2 * bp)                                # out_3_3 -> cat[2]
            _, (t42,) = c3(t41, 0, "2_4_1", sums=[(o32, 0)], keep=False)                 # out_4_1 + out_3_2
            _, (t43,) = c3(t42, 0, "2_4_2", sums=[(cat, 2 * bp)], keep=False)            # out_4_2 + cat[2]
            o43 = c3(t43, 0, "2_4_3")
            c3(o43, 0, "2_4_4", out=cat, out_coff=3 * bp)                                # out_4_4 -> cat[3]
            r = x
            if (p + ".downsample.conv.weight") in sd:
                r = self._conv_v(x, p + ".downsample", stride=stride, relu=False)
            return self._conv_v(cat, p + ".conv_bn_relu3", res=r, in_map=scatter, cin_t=4 * bp)

        c3(s, 0, "2_1_1", out=cat, out_coff=0)                                   # out_1_1 -> cat[0]
        o21 = c3(self._add(s, bp, cat, 0, bp, p + ".add21"), 0, "2_2_1")
        c3(o21, 0, "2_2_2", out=cat, out_coff=bp)                                # out_2_2 -> cat[1]
        o31 = c3(self._add(s, 2 * bp, o21, 0, bp, p + ".add31"), 0, "2_3_1")
        o32 = c3(self._add(o31, 0, cat, bp, bp, p + ".add32"), 0, "2_3_2")
        c3(o32, 0, "2_3_3", out=cat, out_coff=2 * bp)                            # out_3_3 -> cat[2]
        o41 = c3(self._add(s, 3 * bp, o31, 0, bp, p + ".add41"), 0, "2_4_1")
        o42 = c3(self._add(o41, 0, o32, 0, bp, p + ".add42"), 0, "2_4_2")
        o43 = c3(self._add(o42, 0, cat, 2 * bp, bp, p + ".add43"), 0, "2_4_3")
        c3(o43, 0, "2_4_4", out=cat, out_coff=3 * bp)                            # out_4_4 -> cat[3]
        r = x
        if (p + ".downsample.conv.weight") in sd:
            r = self._conv_v(x, p + ".downsample", stride=stride, relu=False)
        return self._conv_v(cat, p + ".conv_bn_relu3", res=r, in_map=scatter, cin_t=4 * bp)

    def _build(self):
        H, W = self.in_h, self.in_w
        w, b = self._fold_cbr("top.conv")
        if tuple(w.shape) != (64, 3, 7, 7):
            raise ValueError("top.conv.conv.weight must be [64,3,7,7]")
        w_off = self._put(w.permute(2, 3, 1, 0).contiguous().numpy().tobytes())     # [ky][kx][ci][cout]
        b_off = self._put(b.numpy().tobytes())
        x = self._new(64, H // 2, W // 2)
        self._ops.append(dict(kind=_lib.UDP_OP_STEM7, ks=7, stride=2, relu=1, cin=3, cout=64, cout_pad=64, hin=H, win=W,
                              hout=H // 2, wout=W // 2, inp=None, out=x, res=None, ups=[], w_off=w_off, b_off=b_off,
                              name="top.conv"))
        pooled = self._new(64, H // 4, W // 4)
        self._op(_lib.UDP_OP_MAXPOOL, x, pooled, "top.maxpool", ks=3, stride=2)
        x = pooled
        feats = []
        for layer, planes in zip(range(1, 5), (64, 128, 256, 512)):
            for blk in range(2):
                x = self._bottleneck(x, "stage0.downsample.layer%d.%d" % (layer, blk), planes,
                                     2 if (layer > 1 and blk == 0) else 1)
            feats.append(x)
        up = None
        out = None
        for ind, xin in enumerate(reversed(feats)):                       # x4, x3, x2, x1
            p = "stage0.upsample.up%d" % (ind + 1)
            if ind == 0:
                out = self._conv_v(xin, p + ".u_skip", relu=True)            # relu(u_skip(x)), no addend
            else:
                big = self._new(up.c, xin.h, xin.w)
                self._op(_lib.UDP_OP_BILINEAR, up, big, p + ".bilinear")
                t = self._conv_v(big, p + ".up_conv", relu=False)
                out = self._conv_v(xin, p + ".u_skip", relu=True, res=t)     # relu(u_skip(x) + up_conv(up))
            up = out
        r1 = self._conv_v(out, "stage0.upsample.up4.res_conv1", relu=True)
        # the closing bilinear resize to OUTPUT_SHAPE (:255) is the identity at the finest level
        if (out.h, out.w) != (H // 4, W // 4):
            raise ValueError("finest RSN level must be at 1/4 resolution")
        self._conv_v(r1, "stage0.upsample.up4.res_conv2", relu=False, to_output=True)
        self.out_channels = self._ops[-1]["cout"]
