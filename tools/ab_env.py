#!/usr/bin/env python3
"""A/B of environment knobs inside ONE GPU session: python tools/ab_env.py [--model w32] "" "UDP_POSE_WS_T=12" "A=1,B=2" ...
Runs bench.py (headline mode only) once per setting and prints images/s, ms/step and the dominant class's launch time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
model = "w32"
if args and args[0] == "--model":
    model, args = args[1], args[2:]
for setting in args or [""]:
    env = dict(os.environ)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
                        "--no-other-modes", "--no-other-configs"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rf = d.get("roofline") or {}
        cl = rf.get("classes", {})
        print("%-44s %8.1f img/s  %7.3f ms  dominant %s: %.2f us x %s, frac %.3f | %s" % (
            setting or "(default)", d["value"], d["ms_per_step"], str(rf.get("kernel", ""))[-18:], rf.get("avg_launch_us", 0),
            rf.get("launches_per_step"), rf.get("frac", 0),
            " ".join("%s=%.2f" % (k.replace("conv", "c").replace("merged_", "m_"), v["ms"]) for k, v in cl.items())), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-44s FAILED %s\n%s" % (setting, e, r.stderr[-1500:]), flush=True)
