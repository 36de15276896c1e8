set -o pipefail
R=$PWD; O=$R/gpurun_out/r04w; mkdir -p $O
for rep in 1 2; do
for cfg in "OPD_TAIL3=1" "OPD_TAIL3=0" "OPD_TAIL3=2"; do
env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --serial-steps 30 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r04w/bench.json').read().strip().splitlines()[-1])
print("$cfg", d['value'], d['serial']['ms_per_step'], 'stage3', d['stage_ms'][3])
PY
done
done
