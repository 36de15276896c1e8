set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 120 python tools/_dbg_cross.py 2>&1 | grep -v amdgpu.ids | head -12 | cut -c1-300
timeout -k 10 300 python -m pytest tests/test_decoder_gpu.py -q -m gpu > $O/tests_dec.log 2>&1; echo "dec tests rc=$?"; tail -5 $O/tests_dec.log | cut -c1-220
timeout -k 10 700 python -m pytest tests -q -m gpu --deselect tests/test_decoder_gpu.py > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -32 $O/tests_gpu.log | cut -c1-200
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['serial'], d['stage_ms'])
PY
