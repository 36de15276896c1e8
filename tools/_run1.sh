set -o pipefail
R=$PWD; O=$R/gpurun_out/r04w; mkdir -p $O
for rep in 1 2 3; do
for cfg in "OPD_ENC_FRONT=1" "OPD_ENC_FRONT=0"; do
env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --serial-steps 10 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r04w/bench.json').read().strip().splitlines()[-1])
print("$cfg", d['value'], d['serial']['ms_per_step'], 'enc', d['stage_ms'][5])
PY
done
done
