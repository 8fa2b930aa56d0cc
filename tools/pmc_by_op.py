#!/usr/bin/env python3
"""Map rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE dispatch rows of tools/one_forward.py onto the ops of
the HRNet program (dispatch order == op order) and print HBM traffic per kernel class.

    rocprofv3 --pmc FETCH_SIZE -d <fetch_dir> --output-format csv -- python3 tools/one_forward.py bf16 --per-op
    rocprofv3 --pmc WRITE_SIZE -d <write_dir> --output-format csv -- python3 tools/one_forward.py bf16 --per-op
    UDP_POSE_NO_GROUPS=1 python tools/pmc_by_op.py <fetch_dir> <write_dir> [bf16|f32] > profiles/rNN_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read (MI355X_MICROARCH.md, HBM section), so it is doubled here.
"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from udp_pose_amd import synth, hrnet_plan


def rows(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    out = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in ("conv_mfma_", "conv_ws_", "conv_chain_", "basic_block_c32", "stem_conv_kernel", "stem_mfma_k", "fuse_sum_kernel")):
            out.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out.sort()
    return [v for _, v in out]


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0)
    prog = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, dtype)
    desc = prog.describe()
    n = len(desc)
    fe, wr = rows(fetch_dir, "FETCH_SIZE"), rows(write_dir, "WRITE_SIZE")
    assert len(fe) % n == 0 and len(wr) % n == 0, (len(fe), len(wr), n)
    fe, wr = fe[-n:], wr[-n:]                    # the last forward of the run
    esz = 2 if dtype == "bf16" else 4            # bytes per stored element (f16x2: hi + lo)
    b = 128
    classes = {}
    for (name, kind, ks, st, cin, cout, ho, wo), op, f, w in zip(desc, prog._ops, fe, wr):
        key = {0: "stem", 2: "fuse_sum", 10: "basic_block_c32"}.get(kind, "conv%dx%d_s%d_nb%d" % (ks, ks, st, 4 if cout % 64 == 0 else 2))
        alg = (ho * st * wo * st * cin * (4 if kind == 0 else esz) + ho * wo * cout * esz) * b
        if op.get("chain_out") is not None:      # chained 1x1 conv (udp_conv_op.chain_cout): its output is written too
            alg += op["chain_out"].elems * esz * b
        extra = ((op["res"].elems if op.get("res") is not None else 0) + sum(t.elems for t, _ in op.get("ups", []))) * esz * b
        c = classes.setdefault(key, dict(launches=0, fetch_bytes=0.0, write_bytes=0.0, algorithmic_in_out_bytes=0.0,
                                         algorithmic_bytes_with_addends=0.0))
        c["launches"] += 1
        c["fetch_bytes"] += 2.0 * f * 1024.0
        c["write_bytes"] += w * 1024.0
        c["algorithmic_in_out_bytes"] += alg
        c["algorithmic_bytes_with_addends"] += alg + extra
    for c in classes.values():
        c["hbm_bytes_per_launch"] = (c["fetch_bytes"] + c["write_bytes"]) / c["launches"]
        c["traffic_over_algorithmic"] = round((c["fetch_bytes"] + c["write_bytes"]) / c["algorithmic_bytes_with_addends"], 3)
    import hashlib
    from udp_pose_amd import _lib
    with open(_lib.LIB_PATH, "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()
    print(json.dumps({"dtype": dtype, "images_per_forward": b, "lib_sha256": sha, "note": "FETCH_SIZE doubled (gfx950 correction); "
                      "separate --pmc passes; bytes per forward of 2N=128 images", "classes": classes}, indent=1))


if __name__ == "__main__":
    main()
