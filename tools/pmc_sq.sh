#!/bin/bash
# SQ issue / stall counters of the score kernel in two passes of <= 8 counters (own runs, kernel-trace only beside
# them).  Run through gpurun from the repo root:   tools/pmc_sq.sh <tag> [workload] [steps]  -> gpurun_out/sq_<tag>.json
set -e
TAG=$1
WL=${2:-wn18rr_asym_r10x200_b512_f32}
STEPS=${3:-25}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmcsq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - "$OUT" "$WL" "$ROOT/gpurun_out/sq_$TAG.json" <<'PY'
import csv, glob, json, sys, collections
out, wl, dst = sys.argv[1:4]
acc = collections.defaultdict(list)
dur = []
name = None
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "score_" in row["Kernel_Name"] and "kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            name = row["Kernel_Name"][:70]
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
res = {k: sum(v[5:]) / max(1, len(v[5:])) for k, v in acc.items()}
json.dump({"workload": wl, "kernel": name, "duration_us_mean": sum(dur) / max(1, len(dur)), "counters": res}, open(dst, "w"), indent=1, sort_keys=True)
print(open(dst).read())
PY
