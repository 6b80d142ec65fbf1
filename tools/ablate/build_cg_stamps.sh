#!/bin/bash
# Diagnostic build of the library with the cg kernel's timeline stamps (tools/ablate/run_cg_timeline.py):
# the cg objects are recompiled with -DRTK_CG_STAMPS, every other object is the product build's.
set -e
cd "$(dirname "$0")/../../r-tucker_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -Wno-unused-variable -DRTK_CG_STAMPS"
mkdir -p /tmp/cg_stamps
for sg in 0 1 2; do
  # only the fast-logistic object (sg 2) holds instantiations the timeline tool runs: KS = 13 alone keeps the build short
  hipcc $FLAGS -DRTK_CG_SG=$sg -c rtk_score_cg.hip -o /tmp/cg_stamps/rtk_score_cg_sg$sg.o &
done
wait
objs=$(ls obj/*.o | grep -v rtk_score_cg_sg)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/librtk_cg_stamps.so $objs /tmp/cg_stamps/rtk_score_cg_sg*.o -ldl
echo built tools/ablate/librtk_cg_stamps.so
