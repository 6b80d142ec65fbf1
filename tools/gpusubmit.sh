#!/bin/bash
# Submit a command to the GPU pool, resubmitting while the pool answers "no slot free" (exit 3: nothing ran,
# nothing was charged).  Any other outcome -- including a failed or timed-out command -- is final.
#   tools/gpusubmit.sh <timeout_s> '<command>'
T=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[gpusubmit] no slot (attempt $attempt), waiting 45 s" >&2
  sleep 45
done
exit 3
