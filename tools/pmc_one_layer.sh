#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of single conv_gemm shapes (tools/one_layer.py) -> gpurun_out/pmc1/<tag>.txt
set -o pipefail
R=$PWD; O=$R/gpurun_out/pmc1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() {  # tag counter args...
  tag=$1; ctr=$2; shift 2
  timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/one_layer.py "$@" > $O/$tag.log 2>&1 || return 1
  python3 - "$O/$tag" "$tag" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'conv_gemm' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], ' '.join(f"{k}={sum(v[2:])/max(len(v)-2,1):.0f}" for k,v in sorted(acc.items())), flush=True)
PY
}

if [ $# -gt 0 ]; then
  # usage: pmc_one_layer.sh tag COUNTER one_layer-args...   (several triples separated by --)
  while [ $# -gt 0 ]; do
    a=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do a+=("$1"); shift; done; [ $# -gt 0 ] && shift
    run "${a[@]}" || exit 1
  done
  exit 0
fi
run n1024_res WRITE_SIZE 8 50 84 256 1024 1 1 1 &&
run n1024_nores WRITE_SIZE 8 50 84 256 1024 1 1 0 &&
run n512 WRITE_SIZE 8 50 84 256 512 1 1 0 &&
run n256 WRITE_SIZE 8 50 84 256 256 1 1 0 &&
run n2048_m8400 WRITE_SIZE 8 25 42 256 2048 1 1 0 &&
run n1024_m67200 WRITE_SIZE 16 50 84 256 1024 1 1 0
