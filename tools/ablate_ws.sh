#!/bin/bash
# Diagnostic builds of the library with parts of the weight-stationary conv body removed (results are WRONG, timing only):
#   tools/ablate_ws.sh [bits...]  ->  udp-pose_amd/libudp_pose_hip_dbg<bits>.so ; run with UDP_POSE_LIB=<path>
# bits (UDP_WS_DBG): 1 no A-fragment prefetch, 2 no B-fragment reads, 4 no DMA after chunk 0, 16 one MFMA per block,
#   32 / 64 the one-chunk (Cin <= 32) convs store nothing / read no residual
cd "$(dirname "$0")/../udp-pose_amd/csrc" || exit 1
for d in ${@:-1 2 4 16}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUDP_WS_DBG=$d -c conv_ws.hip -o conv_ws_dbg$d.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv.o conv_ws_dbg$d.o hrnet.o decode.o data.o psa.o train.o nms.o -o ../libudp_pose_hip_dbg$d.so ) &
done
wait
