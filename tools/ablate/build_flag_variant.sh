#!/bin/bash
# A/B build of the WHOLE library with extra compiler flags:  tools/ablate/build_flag_variant.sh <name> <flags...>
#   -> tools/ablate/librtk_cg_<name>.so   (R_TUCKER_AMD_LIB=<that file>, or tools/ab_lib.sh "<name> product")
set -e
name=$1; shift
cd "$(dirname "$0")/../../r-tucker_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function $*"
out=/tmp/flagvar_$name
mkdir -p $out
SRCS="rtk_abi rtk_gemm_f32 rtk_gemm_sf16 rtk_query rtk_query_bwd rtk_score_split rtk_score_ws rtk_score_bf16 rtk_rank rtk_bce rtk_chol rtk_comm
      rtk_score_cg:_sg0:-DRTK_CG_SG=0 rtk_score_cg:_sg1:-DRTK_CG_SG=1 rtk_score_cg:_sg2:-DRTK_CG_SG=2"
objs=()
for ent in $SRCS; do
  IFS=: read -r f suf extra <<< "$ent"
  o=$out/$f$suf.o
  objs+=($o)
  ( hipcc $FLAGS $extra -c $f.hip -o $o ) &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/librtk_cg_$name.so "${objs[@]}" -ldl
echo built tools/ablate/librtk_cg_$name.so
