#!/bin/bash
# A/B of environment switches under the multi-stream bench on ONE box: tools/ab_env.sh <outdir> <steps> "NAME=VAL ..." "NAME=VAL ..." ...
# (each configuration twice, interleaved; prints frames/s, ms per step, the serial leg and the in-graph stage times)
O=${1:?outdir}; K=${2:?steps}; shift 2
mkdir -p $O
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env OPD_BENCH_SUSTAINED=0 $cfg python bench.py --steps $K --no-cpu-baseline --serial-steps 20 > $O/cfg${i}_$rep.json 2> $O/cfg${i}_$rep.err
    python - <<P
import json
d=json.load(open("$O/cfg${i}_$rep.json")); print("%-44s %8.1f frames/s  %.3f ms/step  serial %.3f ms  stage_ms %s" % ("$cfg", d["value"], d["ms_per_step"], d["serial"]["ms_per_step"], d.get("stage_ms")), flush=True)
P
  done
done
