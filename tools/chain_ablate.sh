#!/bin/bash
# conv_chain_kernel A/B on the GPU box: tile variants (UDP_POSE_CHAIN_VAR) and timing-only ablations (UDP_POSE_CHAIN_DBG:
# 1 no weights, 2 no stores, 4 no residual, 8 no X -- zero-length descriptors, results WRONG); per-launch times of the layer1 ops
for v in ${VARS:-0 1 2 3}; do for d in ${DBGS:-0}; do
  UDP_POSE_CHAIN_VAR=$v UDP_POSE_CHAIN_DBG=$d UDP_POSE_LANES=1 timeout -k 10 200 python tools/profile_layers.py --dtype f16x2 --reps 3 > gpurun_out/layers_chain_v${v}_d$d.txt 2>&1
  echo "== var $v dbg $d"; grep -E "1 1 1 .*256  64x48" gpurun_out/layers_chain_v${v}_d$d.txt
done; done
