#!/bin/bash
# What does the decoder cost inside the graph-replayed forward?  Serial step time (one handle, one blocking call per step) with all six
# decoder layers against none (OPD_DBG_DEC_LAYERS: timing ablation, results wrong), fused and unfused decoder, alternating on one box.
O=${1:-gpurun_out/dec_cost}; mkdir -p $O
for rep in 1 2; do for f in 1 0; do for n in 6 0; do
OPD_FUSED_DEC=$f OPD_DBG_DEC_LAYERS=$n timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 --serial-steps 200 > $O/b.json 2> $O/b.err && python - $f $n $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]+'/b.json').read().strip().splitlines()[-1])
print('fused_dec', sys.argv[1], 'decoder layers', sys.argv[2], 'serial ms', d['serial']['ms_per_step'], 'three-stream fps', d['value'], flush=True)
PY
done; done; done
