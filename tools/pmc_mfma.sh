#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/pmc_mfma; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $O/run.log 2>&1 && python3 $R/tools/pmc_mfma.py $O/*/*_counter_collection.csv | tee $O/summary.txt
