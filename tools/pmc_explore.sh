#!/bin/bash
# Exploratory counter passes over ONE fused bottleneck-tail shape (tools/one_btail.py): what the CU's memory path is doing.
#   usage: tools/pmc_explore.sh <tag> B H W C1 C3 stride      -> gpurun_out/pmcx/<tag>.txt
# One rocprofv3 run per counter group (no trace domains beside --kernel-trace); a group the profiler rejects is reported and skipped.
R=$PWD; O=$R/gpurun_out/pmcx; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
tag=$1; shift
CGRP=(
 "TA_BUSY_avr TA_BUSY_max GRBM_TA_BUSY GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"
 "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TCP_TA_TCP_STATE_READ_sum"
)
: > $O/$tag.txt
i=0
for g in "${CGRP[@]}"; do
  i=$((i+1)); rm -rf $O/run
  if timeout -k 10 120 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/run -- python3 $R/tools/one_btail.py "$@" > $O/$tag.g$i.log 2>&1; then
    python3 - "$O/run" >> $O/$tag.txt <<'PY'
import csv,glob,sys,collections
fs=glob.glob(sys.argv[1]+'/*/*_counter_collection.csv')
acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'btail' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(f"{k:44s} {sum(v[1:])/max(len(v)-1,1):16.0f}   ({len(v)} launches)", flush=True)
PY
  else
    echo "group $i rejected: $g" >> $O/$tag.txt; tail -2 $O/$tag.g$i.log >> $O/$tag.txt
  fi
done
rm -rf $O/run; cat $O/$tag.txt
