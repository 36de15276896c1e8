set -o pipefail
R=$PWD; O=$R/gpurun_out/r04x; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests_all.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r04x/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], d['stage_ms'])
PY
done
