set -o pipefail
R=$PWD; O=$R/gpurun_out/b1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
OPD_BENCH_SUSTAINED=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --batch 1 --streams 1 --steps 5 --warmup 1 --no-cpu-baseline --serial-steps 0 > $O/prof.log 2>&1 &&
cd $R && python tools/timeline.py $(ls -t $O/prof/*/*_kernel_trace.csv | head -n 1) --all > $O/timeline_batch1.txt && rm -rf $O/prof && tail -40 $O/timeline_batch1.txt
