set -o pipefail
O=gpurun_out/r04p; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm_ln" > $O/tests_k.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_k.log | cut -c1-200
timeout -k 10 120 python tools/bench_gemm_ln.py > $O/bench_gemm_ln.txt 2>&1; grep -v amdgpu $O/bench_gemm_ln.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], 'stage_ms', d['stage_ms'], 'rs', d.get('roofline_serial'))
PY
