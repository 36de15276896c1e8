#!/bin/bash
# frames/s of the three-stream bench against the batch size of a step (wave quantisation of the stage-3 / stage-4 tiles): tools/batch_sweep.sh <outdir> <steps> <batch> ...
O=${1:?outdir}; K=${2:?steps}; shift 2
mkdir -p $O
for b in "$@"; do
  OPD_BENCH_SUSTAINED=0 python bench.py --batch $b --steps $K --no-cpu-baseline --serial-steps 20 > $O/b$b.json 2> $O/b$b.err
  python - <<P
import json
d=json.load(open("$O/b$b.json")); s=d["serial"]
print("batch %2d  %8.1f frames/s  %.3f ms/step  | one stream, blocking: %8.1f frames/s  %.3f ms/step  stage_ms %s" % ($b, d["value"], d["ms_per_step"], s["frames_per_s"], s["ms_per_step"], d["stage_ms"]), flush=True)
P
done
