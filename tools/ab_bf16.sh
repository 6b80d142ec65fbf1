#!/bin/bash
# A/B of the bf16 shard score kernel on one MI355X: parity tests, then the C5 shard bench for each variant.
cd "$(dirname "$0")/.."
W=synthetic1m_shard125k_r256x512_b8192_bf16
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -x -q -k "deep_k or c5" 2>&1 | tail -4
run() { echo "== $1"; shift; env "$@" python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('ms/step %.4f kernel_ms %.4f frac %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))"; }
run "deep-K kernel (default)" X=1
run "deep-K kernel (default), again" X=1
run "round-2 8-wave kernel" RTK_BF16_NO_W1=1
run "deep-K, unblocked query sweep" RTK_BF16_QB_KB=1000000
run "deep-K, 6 MB blocks" RTK_BF16_QB_KB=6144
run "deep-K, no nontemporal stores" RTK_NO_NT_STORES=1
