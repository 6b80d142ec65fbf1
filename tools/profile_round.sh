#!/bin/bash
# Collect a round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>
# -> gpurun_out/prof_<tag>/{c2,c3,c4,c5}_kernel_stats.csv + bench JSON lines (un-profiled and profiled runs).
# Counters (PMC) are collected by tools/pmc_traffic.sh / tools/pmc_sq.sh in their own passes.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A W=( [c2]=wn18rr_asym_r10x200_b512_f32 [c3]=fb15k237_sym_r200x200_b2048_bf16 [c4]=fb15k_asym_r200x200_b512_f32 [c5]=synthetic1m_shard125k_r256x512_b8192_bf16 )
declare -A STEPS=( [c2]=2000 [c3]=1000 [c4]=500 [c5]=100 )
for k in c2 c3 c4 c5; do
  timeout -k 10 300 python3 $ROOT/bench.py --workload ${W[$k]} --steps ${STEPS[$k]} --warmup 10 > $OUT/${k}_bench.json 2> $OUT/${k}_bench.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$k -o p -- python3 $ROOT/bench.py --workload ${W[$k]} --steps ${STEPS[$k]} --warmup 10 --no-cpu-baseline > $OUT/${k}_prof_bench.json 2> $OUT/${k}_prof.err
  cp /tmp/prof_${TAG}_$k/p_kernel_stats.csv $OUT/${k}_kernel_stats.csv
  rm -rf /tmp/prof_${TAG}_$k
  echo "$k done"
done
ls $OUT
