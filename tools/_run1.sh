set -o pipefail
R=$PWD; O=$R/gpurun_out/r04u; mkdir -p $O
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04u/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], 'stage_ms', d['stage_ms'])
PY
done
OPD_FUSED_ENC_FFN=0 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench0.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04u/bench0.json').read().strip().splitlines()[-1])
print('two-launch FFN:', d['value'], d['serial'], 'stage_ms', d['stage_ms'])
PY
