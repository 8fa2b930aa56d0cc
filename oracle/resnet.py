"""Oracle: SimpleBaseline pose_resnet forward on CPU (TEST INFRASTRUCTURE ONLY).

Functional restatement of deep_hrnet/lib/models/pose_resnet.py (Bottleneck :64-100, PoseResNet
:105-208: 7x7 s2 stem, max-pool, four Bottleneck stages with the stride on the 3x3 conv, three
ConvTranspose2d(k=4,s=2,p=1)+BN+ReLU, final 1x1 conv) driven by the reference state_dict.
Used for BASELINE.json configs[0] (pose_resnet_50 256x192, batch 1, CPU plumbing + UDP decode).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(sd, x, name, calibrate):
    if calibrate:
        sd[name + ".running_mean"] = x.mean(dim=(0, 2, 3)).clone()
        sd[name + ".running_var"] = x.var(dim=(0, 2, 3), unbiased=False).clone()
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], training=False, eps=BN_EPS)


@torch.no_grad()
def pose_resnet_forward(sd, x, layers=(3, 4, 6, 3), num_deconv=3, calibrate=False):
    x = F.relu(_bn(sd, F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), "bn1", calibrate))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, nblk in enumerate(layers, start=1):
        for b in range(nblk):
            p = "layer%d.%d" % (li, b)
            stride = 2 if (li > 1 and b == 0) else 1
            out = F.relu(_bn(sd, F.conv2d(x, sd[p + ".conv1.weight"]), p + ".bn1", calibrate))
            out = F.relu(_bn(sd, F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1), p + ".bn2", calibrate))
            out = _bn(sd, F.conv2d(out, sd[p + ".conv3.weight"]), p + ".bn3", calibrate)
            res = x
            if (p + ".downsample.0.weight") in sd:
                res = _bn(sd, F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), p + ".downsample.1", calibrate)
            x = F.relu(out + res)
    for d in range(num_deconv):
        w = sd["deconv_layers.%d.weight" % (3 * d)]
        x = F.conv_transpose2d(x, w, sd.get("deconv_layers.%d.bias" % (3 * d)), stride=2, padding=1, output_padding=0)
        x = F.relu(_bn(sd, x, "deconv_layers.%d" % (3 * d + 1), calibrate))
    pad = (sd["final_layer.weight"].shape[2] - 1) // 2
    return F.conv2d(x, sd["final_layer.weight"], sd["final_layer.bias"], padding=pad)
