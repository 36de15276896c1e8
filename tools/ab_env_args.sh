#!/bin/bash
# like tools/ab_env.sh with extra bench.py arguments: tools/ab_env_args.sh <outdir> "<bench args>" "NAME=VAL ..." ...
O=${1:?outdir}; A=${2:?bench args}; shift 2
mkdir -p $O
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env OPD_BENCH_SUSTAINED=0 $cfg python bench.py $A --no-cpu-baseline --serial-steps 20 > $O/cfg${i}_$rep.json 2> $O/cfg${i}_$rep.err
    python - <<P
import json
d=json.load(open("$O/cfg${i}_$rep.json")); print("%-44s %8.1f frames/s  %.3f ms/step  serial %.3f ms  stage_ms %s" % ("$cfg", d["value"], d["ms_per_step"], d["serial"]["ms_per_step"], d.get("stage_ms")), flush=True)
P
  done
done
