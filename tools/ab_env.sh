#!/bin/bash
# One box, alternating runs of bench.py under different environments:
#   tools/ab_env.sh "<bench args>" "ENV=a ENV2=b" "ENV=c" ...   (each quoted string = one variant; "-" = no extra env)
args=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e=""
    env $e timeout -k 10 200 python bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v] rep $rep: step %.2f us  per-batch %.2f  kernel(bracket) %.2f  back-to-back %.2f' % (d['ms_per_step']*1e3, d.get('per_batch_ms_per_step',0)*1e3, d['roofline']['kernel_ms']*1e3, d.get('score_kernel_back_to_back_ms',0)*1e3))"
  done
done
