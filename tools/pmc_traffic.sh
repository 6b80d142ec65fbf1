#!/bin/bash
# HBM-side traffic of the score kernel per launch, from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
# kernel-trace only beside them), as MI355X_MICROARCH.md prescribes.  Run through gpurun from the repo root:
#   tools/pmc_traffic.sh <tag> ["ENV=VAL ..."]      -> gpurun_out/traffic_<tag>.json
set -e
TAG=$1; ENVS=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmc_$TAG
mkdir -p $OUT
WL=${WORKLOAD:-wn18rr_asym_r10x200_b512_f32}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env $ENVS timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python3 $ROOT/bench.py --workload $WL --steps 25 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/$c.err
done
python3 - "$OUT" "$WL" "$ROOT/gpurun_out/traffic_$TAG.json" <<'PY'
import csv, glob, json, sys
out, wl, dst = sys.argv[1:4]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    vals, name = [], None
    for row in csv.DictReader(open(f)):
        if "score_" in row["Kernel_Name"] and "kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
            vals.append(float(row["Counter_Value"])); name = row["Kernel_Name"]
    vals = vals[5:]                      # skip the warm-up launches
    res[c] = (sum(vals) / len(vals), len(vals), name)
fetch = res["FETCH_SIZE"][0] * 1024 * 2      # KB -> bytes; gfx950 reports half the bytes of wide coalesced reads
write = res["WRITE_SIZE"][0] * 1024
json.dump({"workload": wl, "kernel": res["FETCH_SIZE"][2][:60], "dispatches": res["FETCH_SIZE"][1],
           "FETCH_SIZE_KB_raw": res["FETCH_SIZE"][0], "WRITE_SIZE_KB_raw": res["WRITE_SIZE"][0],
           "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
           "_comment": "mean per score-kernel launch; FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is; separate PMC passes"},
          open(dst, "w"), indent=1)
print(open(dst).read())
PY
