set -o pipefail
R=$PWD; O=$R/gpurun_out/r04u; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "enc_ffn or gemm_ln_deep" > $O/tests_k.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests_k.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/bench_enc_ffn.py 2>&1 | tee $O/bench_enc_ffn.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04u/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], 'stage_ms', d['stage_ms'])
PY
