#!/bin/bash
# Diagnostic builds of the library with parts of conv_ws_h2_kernel removed (results are WRONG, timing only):
#   tools/ablate_ws.sh  ->  udp-pose_amd/libudp_pose_hip_dbg{1,2,4,7}.so ; run with UDP_POSE_LIB=<path>
cd "$(dirname "$0")/../udp-pose_amd/csrc" || exit 1
for d in 1 2 4 7; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUDP_WS_DBG=$d -c conv.hip -o conv_dbg$d.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_dbg$d.o hrnet.o decode.o data.o psa.o train.o nms.o -o ../libudp_pose_hip_dbg$d.so ) &
done
wait
