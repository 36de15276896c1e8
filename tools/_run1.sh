set -o pipefail
R=$PWD; O=$R/gpurun_out/r04w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_detector_gpu.py tests/test_workloads_gpu.py -q -m gpu -x > $O/tests_d.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests_d.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --serial-steps 100 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r04w/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], d['roofline']['graph_ms_per_step'])
PY
done
