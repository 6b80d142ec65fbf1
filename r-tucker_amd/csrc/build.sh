#!/bin/bash
# Build librtucker_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" obj
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function"
# "source" or "source:object-suffix:extra flag" (one source compiled into several objects)
SRCS="rtk_abi rtk_gemm_f32 rtk_gemm_sf16 rtk_query rtk_query_bwd rtk_score_split rtk_score_ws rtk_score_bf16 rtk_rank rtk_bce rtk_chol rtk_comm
      rtk_score_cg:_sg0:-DRTK_CG_SG=0 rtk_score_cg:_sg1:-DRTK_CG_SG=1 rtk_score_cg:_sg2:-DRTK_CG_SG=2"
pids=()
objs=()
# incremental: a source is recompiled when it, any header here or the public header is newer than its object
# (RTK_REBUILD=1 forces everything)
newest_hdr=$(ls -t *.h ../../include/*.h build.sh | head -1)
for ent in $SRCS; do
  IFS=: read -r f suf extra <<< "$ent"
  o=obj/$f$suf.o
  objs+=($o)
  if [ -z "$RTK_REBUILD" ] && [ $o -nt $f.hip ] && [ $o -nt "$newest_hdr" ]; then continue; fi
  ( hipcc $FLAGS $extra -c $f.hip -o $o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/librtucker_hip.so" "${objs[@]}" -ldl
echo "built $OUT/librtucker_hip.so"
