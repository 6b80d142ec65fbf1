#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>
# -> gpurun_out/prof_<tag>/{c2,c3,c5}_kernel_stats.csv, bench JSON lines, PMC passes for the C2 score kernel.
# Counters are collected in their own passes (kernel-trace only beside them).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A W=( [c2]=wn18rr_asym_r10x200_b512_f32 [c3]=fb15k237_sym_r200x200_b2048_bf16 [c5]=synthetic1m_shard125k_r256x512_b8192_bf16 )
declare -A STEPS=( [c2]=2000 [c3]=1000 [c5]=100 )
for k in c2 c3 c5; do
  timeout -k 10 300 python3 $ROOT/bench.py --workload ${W[$k]} --steps ${STEPS[$k]} --warmup 10 > $OUT/${k}_bench.json 2> $OUT/${k}_bench.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${k}_trace -o p -- python3 $ROOT/bench.py --workload ${W[$k]} --steps ${STEPS[$k]} --warmup 10 --no-cpu-baseline > $OUT/${k}_prof_bench.json 2> $OUT/${k}_prof.err
  cp $OUT/${k}_trace/p_kernel_stats.csv $OUT/${k}_kernel_stats.csv
  echo "$k done"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/c2_pmc_$c -o p -- python3 $ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/c2_pmc_$c.err
  echo "pmc $c done"
done
rm -rf $OUT/*_trace/p_kernel_trace.csv
ls $OUT
