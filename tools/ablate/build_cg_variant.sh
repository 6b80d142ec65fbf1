#!/bin/bash
# A/B build of the library with extra flags on the cg kernel's objects only (every other object is the product
# build's):   tools/ablate/build_cg_variant.sh <name> <flags...>   ->  tools/ablate/librtk_cg_<name>.so
# (R_TUCKER_AMD_LIB=<that file> selects it; `stamps` = -DRTK_CG_STAMPS is the timeline build of run_cg_timeline.py)
set -e
name=$1; shift
cd "$(dirname "$0")/../../r-tucker_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -Wno-unused-variable $*"
mkdir -p /tmp/cg_$name
for sg in 0 1 2; do
  hipcc $FLAGS -DRTK_CG_SG=$sg -c rtk_score_cg.hip -o /tmp/cg_$name/rtk_score_cg_sg$sg.o &
done
wait
objs=$(ls obj/*.o | grep -v rtk_score_cg_sg)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/librtk_cg_$name.so $objs /tmp/cg_$name/rtk_score_cg_sg*.o -ldl
echo built tools/ablate/librtk_cg_$name.so
