import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
outs = {}
for fuse in ("0", "1"):
    os.environ["UDP_POSE_RSN_FUSE_ADDS"] = fuse
    _, net = bench.build_net("f16x2", target_type="offset", model="rsn18")
    hp = bench.HotPath(net, 24, torch.device("cuda", 0), seed=3, target_type="offset")
    hp.step(); torch.cuda.synchronize()
    outs[fuse] = net.raw_forward(hp.xin, flip_test=True).clone()
print("fused vs separate sums: equal", bool(torch.equal(outs["0"], outs["1"])), "max diff", float((outs["0"] - outs["1"]).abs().max()), "finite", bool(torch.isfinite(outs["1"]).all()))
