"""Oracle: RSN-18 forward on CPU (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Functional restatement of RSN/exps/RSN18.coco/network.py driven by a reference-format state_dict:
  conv_bn_relu :14-46, RSN Bottleneck :49-122, ResNet_top :125-137,
  ResNet_downsample_module :140-199, Upsample_unit/module :202-296, RSN.forward :400-412
(single stage, inference: returns outputs[-1][-1] = the finest `res`).  Stock torch.nn.functional
ops in fp32.  Pinned against the reference module (tests/golden/rsn18_*.npz, oracle/gen_golden.py).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


class _Net:
    def __init__(self, sd, calibrate=False):
        self.sd, self.calibrate = sd, calibrate

    def cbr(self, x, name, stride=1, relu=True):
        w = self.sd[name + ".conv.weight"]
        y = F.conv2d(x, w, self.sd[name + ".conv.bias"], stride=stride, padding=(w.shape[2] - 1) // 2)
        if self.calibrate:
            self.sd[name + ".bn.running_mean"] = y.mean(dim=(0, 2, 3)).clone()
            self.sd[name + ".bn.running_var"] = y.var(dim=(0, 2, 3), unbiased=False).clone()
        y = F.batch_norm(y, self.sd[name + ".bn.running_mean"], self.sd[name + ".bn.running_var"],
                         self.sd[name + ".bn.weight"], self.sd[name + ".bn.bias"], training=False, eps=BN_EPS)
        return F.relu(y) if relu else y


def _bottleneck(net, x, p, stride):
    out = net.cbr(x, p + ".conv_bn_relu1", stride=stride)
    bch = out.shape[1] // 4
    s = torch.split(out, bch, 1)
    o11 = net.cbr(s[0], p + ".conv_bn_relu2_1_1")
    o21 = net.cbr(s[1] + o11, p + ".conv_bn_relu2_2_1")
    o22 = net.cbr(o21, p + ".conv_bn_relu2_2_2")
    o31 = net.cbr(s[2] + o21, p + ".conv_bn_relu2_3_1")
    o32 = net.cbr(o31 + o22, p + ".conv_bn_relu2_3_2")
    o33 = net.cbr(o32, p + ".conv_bn_relu2_3_3")
    o41 = net.cbr(s[3] + o31, p + ".conv_bn_relu2_4_1")
    o42 = net.cbr(o41 + o32, p + ".conv_bn_relu2_4_2")
    o43 = net.cbr(o42 + o33, p + ".conv_bn_relu2_4_3")
    o44 = net.cbr(o43, p + ".conv_bn_relu2_4_4")
    out = net.cbr(torch.cat((o11, o22, o33, o44), 1), p + ".conv_bn_relu3", relu=False)
    if (p + ".downsample.conv.weight") in net.sd:
        x = net.cbr(x, p + ".downsample", stride=stride, relu=False)
    return F.relu(out + x)


@torch.no_grad()
def rsn_forward(sd, x, output_shape=(64, 48), calibrate=False, taps=None):
    """network.py:400-412 for STAGE_NUM = 1.  x: [N,3,H,W] fp32 -> [N, C_out, *output_shape]."""
    net = _Net(sd, calibrate)
    x = net.cbr(x, "top.conv", stride=2)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps["top"] = x
    feats = []
    for layer in range(1, 5):
        for b in range(2):
            x = _bottleneck(net, x, "stage0.downsample.layer%d.%d" % (layer, b), 2 if (layer > 1 and b == 0) else 1)
        feats.append(x)
        if taps is not None:
            taps["layer%d" % layer] = x
    x1, x2, x3, x4 = feats
    h, w = output_shape
    sizes = [(h // 8, w // 8), (h // 4, w // 4), (h // 2, w // 2), (h, w)]
    up = None
    res = None
    for ind, xin in enumerate((x4, x3, x2, x1)):
        p = "stage0.upsample.up%d" % (ind + 1)
        out = net.cbr(xin, p + ".u_skip", relu=False)
        if ind > 0:
            u = F.interpolate(up, size=sizes[ind], mode="bilinear", align_corners=True)
            out = out + net.cbr(u, p + ".up_conv", relu=False)
        out = F.relu(out)
        r = net.cbr(net.cbr(out, p + ".res_conv1"), p + ".res_conv2", relu=False)
        res = F.interpolate(r, size=output_shape, mode="bilinear", align_corners=True)
        up = out
    return res
