set -o pipefail
R=$PWD; O=$R/gpurun_out/r04t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_detector_gpu.py -q -m gpu -x > $O/tests_k.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests_k.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04t/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['serial'], 'stage_ms', d['stage_ms'])
PY
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --serial-steps 0 > $O/pmc_sq.log 2>&1 && cd $R && python tools/pmc_mfma.py $(ls -t $O/pmc_sq/*/*_counter_collection.csv | head -n 1) > $O/sq.txt && cut -c1-200 $O/sq.txt
