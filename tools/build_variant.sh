#!/bin/bash
# Developer tool: build libmtd_hip.so from the kernel sources of a git revision into tools/bin/libmtd_hip_<name>.so, for same-box
# A/B runs of two builds (LD_PRELOAD=tools/bin/libmtd_hip_<name>.so python tools/bench_mesh.py ...: the preloaded library's symbols win).
# usage: tools/build_variant.sh <git revision> <name>
set -e
REV=$1; NAME=$2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=/tmp/mtd_variant_$NAME
rm -rf $W && mkdir -p $W
git -C "$ROOT" archive $REV metadynamics-plugin_amd/csrc include | tar -x -C $W
make -C $W/metadynamics-plugin_amd/csrc -j8 -s
mkdir -p "$ROOT/tools/bin"
cp $W/metadynamics-plugin_amd/lib/libmtd_hip.so "$ROOT/tools/bin/libmtd_hip_$NAME.so"
rm -rf $W
ls -la "$ROOT/tools/bin/libmtd_hip_$NAME.so"
