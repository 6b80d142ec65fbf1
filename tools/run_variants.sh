#!/bin/bash
# The semantic variants of the normalised Riemannian step against the README recipe compressed x2,
# 150 epochs each, one gpurun call per variant (a call is limited to 20 minutes).  Run from the build container:
#   tools/run_variants.sh "coord regex both"
cd "$(dirname "$0")/.."
declare -A F=( [coord]="--variant-norm coordinate" [regex]="--variant-reg-excluded" [both]="--variant-norm coordinate --variant-reg-excluded" )
for tag in $1; do
  bash tools/gpusubmit.sh 1200 "timeout -k 10 1150 python tools/train_lease.py --compress 2 --max-epochs 150 --minutes 17 --test-every 50 --tag r04var_$tag ${F[$tag]} > gpurun_out/r04var_$tag.stdout 2>&1; echo rc=\$?"
done
