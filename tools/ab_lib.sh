#!/bin/bash
# One box, alternating runs of bench.py (C2) with different builds of the library (R_TUCKER_AMD_LIB):
#   tools/ab_lib.sh "product pf4 ..." [steps]     (product = r-tucker_amd/lib, <name> = tools/ablate/librtk_cg_<name>.so)
steps=${2:-1000}
for rep in 1 2; do
  for v in $1; do
    if [ "$v" = product ]; then unset R_TUCKER_AMD_LIB; else export R_TUCKER_AMD_LIB=$PWD/tools/ablate/librtk_cg_$v.so; fi
    timeout -k 10 200 python bench.py --steps $steps --warmup 200 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib $v rep $rep: step %.2f us  per-batch %.2f  kernel(bracket) %.2f  back-to-back %.2f' % (d['ms_per_step']*1e3, d.get('per_batch_ms_per_step',0)*1e3, d['roofline']['kernel_ms']*1e3, d['score_kernel_back_to_back_ms']*1e3))"
  done
done
