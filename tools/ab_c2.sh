#!/bin/bash
# A/B of environment switches on the bench workload (run through gpurun from the repo root):
#   tools/ab_c2.sh <tag> "VAR1=x VAR2=y" "VAR3=z" ...      (one bench run per quoted variant; "" = defaults)
# -> gpurun_out/ab_<tag>.txt : ms/step and score-kernel ms per variant
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$TAG.txt
: > $OUT
WL=${WORKLOAD:-wn18rr_asym_r10x200_b512_f32}
STEPS=${STEPS:-2000}
for v in "$@"; do
  for rep in 1 2; do
    line=$(env $v python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup 200 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1)
    echo "[$v] rep$rep $(echo $line | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.5f kernel_ms %.5f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))')" | tee -a $OUT
  done
done
