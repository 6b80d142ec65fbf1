#!/bin/bash
# One box, alternating runs: bench.py (C2) under several RTK_CG_TUNE / kernel settings; prints step, kernel bracket, back-to-back.
#   tools/ab_cg.sh "0 1 2 3 ws" [steps]
steps=${2:-1500}
for rep in 1 2; do
  for v in $1; do
    if [ "$v" = ws ]; then export RTK_SCORE_KERNEL=ws; unset RTK_CG_TUNE; else unset RTK_SCORE_KERNEL; export RTK_CG_TUNE=$v; fi
    timeout -k 10 200 python bench.py --steps $steps --warmup 200 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('variant $v rep $rep: step %.2f us  per-batch %.2f  kernel(bracket) %.2f  back-to-back %.2f' % (d['ms_per_step']*1e3, d.get('per_batch_ms_per_step',0)*1e3, d['roofline']['kernel_ms']*1e3, d['score_kernel_back_to_back_ms']*1e3))"
  done
done
