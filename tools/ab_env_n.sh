#!/bin/bash
# like tools/ab_env.sh with N interleaved repeats: tools/ab_env_n.sh <outdir> <steps> <repeats> "NAME=VAL ..." ...
O=${1:?outdir}; K=${2:?steps}; N=${3:?repeats}; shift 3
mkdir -p $O
for rep in $(seq 1 $N); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env OPD_BENCH_SUSTAINED=0 $cfg python bench.py --steps $K --no-cpu-baseline --serial-steps 0 > $O/cfg${i}_$rep.json 2> $O/cfg${i}_$rep.err
    python - <<P
import json
d=json.load(open("$O/cfg${i}_$rep.json")); print("%-44s %8.1f frames/s  %.3f ms/step" % ("$cfg", d["value"], d["ms_per_step"]), flush=True)
P
  done
done
