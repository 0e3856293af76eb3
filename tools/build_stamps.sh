#!/bin/bash
# Developer tool: diagnostic build of libmtd_hip.so with in-kernel time stamps (-DMTD_STAMPS) for tools/stamps.py
set -e
cd "$(dirname "$0")/../metadynamics-plugin_amd/csrc"
mkdir -p ../../tools/bin/obj_stamps
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=fast -DMTD_STAMPS -c $f -o ../../tools/bin/obj_stamps/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libmtd_hip_stamps.so ../../tools/bin/obj_stamps/*.o
