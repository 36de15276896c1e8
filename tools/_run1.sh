set -o pipefail
O=gpurun_out/r04q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -q -m gpu > $O/tests_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -25 $O/tests_bf16.log | cut -c1-250
timeout -k 10 900 python -m pytest tests -q -m gpu --deselect tests/test_bf16_gpu.py > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log | cut -c1-200
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 --dtype bf16 > $O/bench_bf16.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04q/bench_bf16.json').read().strip().splitlines()[-1])
print(d['dtype'], d['value'], d['serial'], 'stage_ms', d['stage_ms'])
PY
