#!/usr/bin/env python3
"""Per-layer-shape sums of arbitrary rocprofv3 --pmc counters of tools/one_forward.py.

    python tools/pmc_by_shape.py <pmc_dir> [bf16|f32] [top_n]
Dispatch order == op order; the last forward of the run is used.
"""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from udp_pose_amd import synth, hrnet_plan


def main():
    d = sys.argv[1]
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("conv_mfma_kernel", "stem_conv_kernel", "fuse_sum_kernel")):
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0)
    desc = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, dtype).describe()
    n = len(desc)
    assert len(ids) % n == 0, (len(ids), n)
    ids = ids[-n:]
    agg = collections.OrderedDict()
    for (name, kind, ks, st, cin, cout, ho, wo), i in zip(desc, ids):
        a = agg.setdefault((kind, ks, st, cin, cout, ho, wo), collections.defaultdict(float))
        a["launches"] += 1
        for k, v in per[i].items():
            a[k] += v
    key = "SQ_WAVE_CYCLES" if any("SQ_WAVE_CYCLES" in a for a in agg.values()) else None
    items = sorted(agg.items(), key=lambda kv: -kv[1].get(key, kv[1]["launches"]))[:top]
    for shape, a in items:
        w = a.get("SQ_WAVES", 0) or 1
        print("k%d ks%d s%d %d->%d %dx%d  launches=%d" % (shape + (int(a["launches"]),)))
        print("    " + "  ".join("%s/wave=%.0f" % (k.replace("SQ_", ""), v / w) for k, v in sorted(a.items())
                                 if k not in ("launches", "SQ_WAVES")) + "  waves/launch=%.0f" % (w / a["launches"]))


if __name__ == "__main__":
    main()
