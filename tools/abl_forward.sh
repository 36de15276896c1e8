#!/bin/bash
# Timing ablations of the WHOLE forward under the multi-stream bench (results are wrong; the question is which resource the headline rate
# is sensitive to): OPD_DBG_BTAIL / OPD_DBG_GEMM set the kernels' dbg bits for every fused tail / implicit-GEMM launch; OPD_DBG_SKIP
# (bit i = segment i of stage_ms: stem, stages 1-4, encoder, decoder, post-process) leaves whole segments' launches out.
#   usage: tools/abl_forward.sh <outdir> [steps]
O=${1:?outdir}; K=${2:-600}
mkdir -p $O
run() {   # name, env...
  local n=$1; shift
  env OPD_BENCH_SUSTAINED=0 "$@" python bench.py --steps $K --no-cpu-baseline --serial-steps 10 > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.load(open("$O/$n.json")); print("%-28s %8.1f frames/s  %.3f ms/step  serial %.3f ms  stage_ms %s" % ("$n", d["value"], d["ms_per_step"], d["serial"]["ms_per_step"], d.get("stage_ms")), flush=True)
P
}
run base X=0
run rc0_ys0 OPD_TAIL_RC=0 OPD_Y_STRIDE2=0
run btail_nostore OPD_DBG_BTAIL=2
run btail_nores OPD_DBG_BTAIL=4
run btail_nostore_nores OPD_DBG_BTAIL=6
run btail_no3x3 OPD_DBG_BTAIL=1
run gemm_nostore OPD_DBG_GEMM=64
run gemm_nomfma OPD_DBG_GEMM=1
run gemm_nodma OPD_DBG_GEMM=2
run no_stage12 OPD_DBG_SKIP=6
run no_stage34 OPD_DBG_SKIP=24
run no_enc_dec OPD_DBG_SKIP=96
run only_stage12 OPD_DBG_SKIP=120
run only_stage34 OPD_DBG_SKIP=102
run only_enc_dec OPD_DBG_SKIP=30
run base2 X=0
