"""Oracle: HRNet (UDP variant) forward on CPU (TEST INFRASTRUCTURE ONLY).

Functional restatement of deep_hrnet/lib/models/pose_hrnet.py driven directly
by a reference-format ``state_dict`` and the ``MODEL.EXTRA`` dict of the YAML:
  forward                        :436-471
  Bottleneck / BasicBlock        :62-100 / :29-59
  HighResolutionModule.forward   :260-273, fuse layers :189-255
  transitions                    :344-383
Stock torch.nn.functional ops in fp32 (conv2d, batch_norm(eval), relu,
interpolate(nearest)).  Pinned against the reference module itself on the same
weights and inputs (tests/golden/hrnet_*.npz, oracle/gen_golden.py).

``calibrate=True`` runs every BatchNorm on batch statistics and writes them
back into ``sd`` as running_mean / running_var (used once, in the build
container, to give synthetic weights realistic statistics).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


class _Net:
    def __init__(self, sd, calibrate=False, train=False):
        self.sd = sd
        self.calibrate = calibrate
        self.train = train

    def conv(self, x, name, stride=1, pad=None):
        w = self.sd[name + ".weight"]
        if pad is None:
            pad = (w.shape[2] - 1) // 2
        return F.conv2d(x, w, self.sd.get(name + ".bias"), stride=stride, padding=pad)

    def bn(self, x, name):
        if self.calibrate:
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            self.sd[name + ".running_mean"] = mean.clone()
            self.sd[name + ".running_var"] = var.clone()
        if self.train:
            # nn.BatchNorm2d in train mode: batch statistics, running stats updated in place with
            # momentum 0.1 (BN_MOMENTUM pose_hrnet.py:20 == the nn default used by fuse/transition BNs)
            return F.batch_norm(x, self.sd[name + ".running_mean"], self.sd[name + ".running_var"],
                                self.sd[name + ".weight"], self.sd[name + ".bias"],
                                training=True, momentum=0.1, eps=BN_EPS)
        return F.batch_norm(x, self.sd[name + ".running_mean"], self.sd[name + ".running_var"],
                            self.sd[name + ".weight"], self.sd[name + ".bias"],
                            training=False, eps=BN_EPS)

    def has(self, name):
        return (name + ".weight") in self.sd


def _psa_s(net, x, p):
    """PSA_s.forward (deep_hrnet/lib/models/PSA.py:190-269): spatial_pool then channel_pool."""
    sd = net.sd
    n, c, h, w = x.shape
    # spatial_pool :190-222
    v = F.conv2d(x, sd[p + ".conv_v_right.weight"]).view(n, c // 2, h * w)
    q = F.softmax(F.conv2d(x, sd[p + ".conv_q_right.weight"]).view(n, 1, h * w), dim=2)
    ctx = torch.matmul(v, q.transpose(1, 2)).unsqueeze(-1)                           # [N, C/2, 1, 1]
    ctx = F.conv2d(ctx, sd[p + ".conv_up.0.weight"], sd[p + ".conv_up.0.bias"])
    ctx = F.layer_norm(ctx, [c // 8, 1, 1], sd[p + ".conv_up.1.weight"], sd[p + ".conv_up.1.bias"], 1e-5)
    ctx = F.conv2d(F.relu(ctx), sd[p + ".conv_up.3.weight"], sd[p + ".conv_up.3.bias"])
    out = x * torch.sigmoid(ctx)
    # channel_pool :224-258
    g = F.conv2d(out, sd[p + ".conv_q_left.weight"])
    avg = F.adaptive_avg_pool2d(g, 1).view(n, c // 2, 1).permute(0, 2, 1)             # [N, 1, C/2]
    theta = F.softmax(F.conv2d(out, sd[p + ".conv_v_left.weight"]).view(n, c // 2, h * w), dim=2)
    ctx = torch.matmul(avg, theta).view(n, 1, h, w)
    return out * torch.sigmoid(ctx)


def _basic_block(net, x, p):
    out = F.relu(net.bn(net.conv(x, p + ".conv1"), p + ".bn1"))
    if (p + ".deattn.conv_q_right.weight") in net.sd:            # pose_hrnet_psa.py:49
        out = _psa_s(net, out, p + ".deattn")
    out = net.bn(net.conv(out, p + ".conv2"), p + ".bn2")
    return F.relu(out + x)


def _bottleneck(net, x, p):
    out = F.relu(net.bn(net.conv(x, p + ".conv1"), p + ".bn1"))
    out = F.relu(net.bn(net.conv(out, p + ".conv2"), p + ".bn2"))
    out = net.bn(net.conv(out, p + ".conv3"), p + ".bn3")
    res = x
    if net.has(p + ".downsample.0"):
        res = net.bn(net.conv(x, p + ".downsample.0"), p + ".downsample.1")
    return F.relu(out + res)


def _fuse_term(net, xj, p, i, j, last_module):
    """f_ij of pose_hrnet.py:198-252."""
    q = "%s.fuse_layers.%d.%d" % (p, i, j)
    if j > i:
        y = net.bn(net.conv(xj, q + ".0"), q + ".1")
        return F.interpolate(y, scale_factor=2 ** (j - i), mode="nearest")
    if j == i:
        if last_module:
            return net.conv(xj, q + ".0")          # 1x1 C -> 4C, no BN (:214-221)
        return xj
    y = xj
    for k in range(i - j):
        y = net.bn(net.conv(y, "%s.%d.0" % (q, k), stride=2), "%s.%d.1" % (q, k))
        if k != i - j - 1:
            y = F.relu(y)
    return y


def _hr_module(net, xs, p, num_blocks, last_module):
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        for k in range(num_blocks[b]):
            xs[b] = _basic_block(net, xs[b], "%s.branches.%d.%d" % (p, b, k))
    n_out = 1 if last_module else nb
    outs = []
    for i in range(n_out):
        y = _fuse_term(net, xs[0], p, i, 0, last_module)
        for j in range(1, nb):
            y = y + _fuse_term(net, xs[j], p, i, j, last_module)
        outs.append(F.relu(y))
    return outs


def _transition(net, ys, name, n_cur):
    n_pre = len(ys)
    xs = []
    for i in range(n_cur):
        q = "%s.%d" % (name, i)
        if i < n_pre:
            if net.has(q + ".0"):
                xs.append(F.relu(net.bn(net.conv(ys[i], q + ".0"), q + ".1")))
            else:
                xs.append(ys[i])
        else:
            y = ys[-1]
            for k in range(i + 1 - n_pre):
                y = F.relu(net.bn(net.conv(y, "%s.%d.0" % (q, k), stride=2), "%s.%d.1" % (q, k)))
            xs.append(y)
    return xs


@torch.no_grad()
def hrnet_forward(sd, extra, x, calibrate=False, taps=None):
    """pose_hrnet.py:436-471.  ``sd``: reference state_dict (fp32 tensors),
    ``extra``: MODEL.EXTRA dict, ``x``: [N,3,H,W] fp32.  Optional ``taps``
    dict receives named intermediate activations."""
    return _forward(_Net(sd, calibrate), extra, x, taps)


def hrnet_forward_train(sd, extra, x, taps=None):
    """Same graph under ``model.train()`` (function.py:38): BatchNorm on batch statistics, running
    statistics of ``sd`` updated in place; autograd is left on so the caller can back-propagate."""
    return _forward(_Net(sd, train=True), extra, x, taps)


def _forward(net, extra, x, taps=None):
    x = F.relu(net.bn(net.conv(x, "conv1", stride=2), "bn1"))
    x = F.relu(net.bn(net.conv(x, "conv2", stride=2), "bn2"))
    if taps is not None:
        taps["stem"] = x
    for k in range(4):
        x = _bottleneck(net, x, "layer1.%d" % k)
    if taps is not None:
        taps["layer1"] = x
    ys = [x]
    for s, tname in ((2, "transition1"), (3, "transition2"), (4, "transition3")):
        cfg = extra["STAGE%d" % s]
        assert cfg["BLOCK"] == "BASIC" and cfg["FUSE_METHOD"] == "SUM"
        xs = _transition(net, ys, tname, cfg["NUM_BRANCHES"])
        for m in range(cfg["NUM_MODULES"]):
            last = (s == 4 and m == cfg["NUM_MODULES"] - 1)
            xs = _hr_module(net, xs, "stage%d.%d" % (s, m), cfg["NUM_BLOCKS"], last)
        ys = xs
        if taps is not None:
            for b, y in enumerate(ys):
                taps["stage%d.%d" % (s, b)] = y
    pad = 1 if extra.get("FINAL_CONV_KERNEL", 1) == 3 else 0
    return net.conv(ys[0], "final_layer", pad=pad)
