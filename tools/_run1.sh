set -o pipefail
O=gpurun_out/r04r; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "stem" > $O/tests_k.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_k.log | cut -c1-200
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04r/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], 'stage_ms', d['stage_ms'])
PY
done
