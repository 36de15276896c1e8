#!/bin/bash
# Round-end evidence run on the GPU box (from the repo root), in two parts (a gpurun call lasts at most 20 minutes):
#   part 1: full GPU test suite, smoke, the default bench line with the CPU baseline, the driver's K = 20 form, rocprofv3 kernel
#           traces of the default command and of the one-batch-in-flight form, the HBM-traffic and SQ counter passes
#   part 2: the 2-rank gloo rehearsal of the self-launched bench, the other BASELINE shapes, host-boundary rates, kernel microbenches
# Everything lands under gpurun_out/final/; tools/collect_profiles.sh copies the summaries into profiles/.
set -o pipefail
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
part=${1:-1}
if [ "$part" = 1 ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1 && tail -2 $O/tests_gpu.log &&
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && cat $O/smoke.log &&
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err && cat $O/bench.json &&
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20.json 2>> $O/bench.err &&
cd /tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --serial-steps 0 > $O/prof_default.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_streams1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --serial-steps 0 > $O/prof_streams1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --serial-steps 0 > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --serial-steps 0 > $O/pmc_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --serial-steps 0 > $O/pmc_sq.log 2>&1 &&
cd $R &&
# summaries are made HERE: the raw traces exceed what a gpurun call copies back (64 MiB)
python tools/timeline.py $(ls -t $O/prof_streams1/*/*_kernel_trace.csv | head -n 1) --all > $O/bench_streams1_timeline.txt &&
python tools/overlap.py $(ls -t $O/prof_default/*/*_kernel_trace.csv | head -n 1) > $O/overlap_three_streams.txt &&
python tools/pmc_traffic.py $(ls -t $O/pmc_fetch/*/*_counter_collection.csv | head -n 1) $(ls -t $O/pmc_write/*/*_counter_collection.csv | head -n 1) $O/pmc_traffic.json &&
python tools/pmc_mfma.py $(ls -t $O/pmc_sq/*/*_counter_collection.csv | head -n 1) $O/pmc_sq.json > $O/pmc_sq_per_kernel.txt &&
cp $(ls -t $O/prof_default/*/*_kernel_stats.csv | head -n 1) $O/bench_default_kernel_stats.csv &&
cp $(ls -t $O/prof_streams1/*/*_kernel_stats.csv | head -n 1) $O/bench_streams1_kernel_stats.csv &&
cp gpurun_out/parity_table.json $O/parity_table.json &&
rm -rf $O/prof_default $O/prof_streams1 $O/pmc_fetch $O/pmc_write $O/pmc_sq && du -sh $O
else
OPD_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err && tail -1 $O/bench_gloo2.json | cut -c1-300 &&
timeout -k 10 300 python tools/host_rate.py 96 > $O/host_rate.txt 2>&1 && timeout -k 10 300 python tools/host_rate.py 96 720 1280 >> $O/host_rate.txt 2>&1 && cat $O/host_rate.txt &&
timeout -k 10 300 python bench.py --arch r101 --height 1066 --width 1920 --no-cpu-baseline --steps 300 > $O/bench_r101_1066x1920.json 2> $O/bench_r101.err && cut -c1-200 $O/bench_r101_1066x1920.json &&
timeout -k 10 300 python bench.py --height 1080 --width 1920 --batch 4 --no-cpu-baseline --steps 300 > $O/bench_r50_tile1080p_b4.json 2> $O/bench_tile.err && cut -c1-200 $O/bench_r50_tile1080p_b4.json &&
timeout -k 10 120 python tools/bench_attn.py > $O/bench_attn.txt 2>&1 &&
timeout -k 10 120 python tools/trace_attn.py > $O/trace_attn.txt 2>&1 &&
timeout -k 10 120 python tools/bench_gemm_ln.py > $O/bench_gemm_ln.txt 2>&1 &&
timeout -k 10 120 python tools/trace_gemm.py > $O/trace_gemm.txt 2>&1 &&
timeout -k 10 300 python tools/bench_layers.py 0 > $O/bench_layers.txt 2>&1 &&
timeout -k 10 300 python tools/bench_btail.py --ablate > $O/bench_btail.txt 2>&1 &&
timeout -k 10 200 python tools/bench_enc_ffn.py --ablate > $O/bench_enc_ffn.txt 2>&1 &&
timeout -k 10 200 python tools/bench_dec.py > $O/bench_dec.txt 2>&1 &&
timeout -k 10 120 tools/microbench/vmorder > $O/microbench_vmorder.txt 2>&1 &&
timeout -k 10 300 python tools/host_rate_ref_pattern.py > $O/host_rate_ref_pattern.txt 2>&1 &&
timeout -k 10 200 python tools/host_b1_probe.py 64 > $O/host_b1_probe.txt 2>&1 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 300 > $O/bench_bf16.json 2> $O/bench_bf16.err &&
timeout -k 10 120 tools/microbench/mfma_peak > $O/microbench_mfma_peak.txt 2>&1 &&
OPD_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --steps 300 --no-cpu-baseline --serial-steps 0 > $O/bench_forced_comm_1rank.json 2> $O/bench_forced_comm.err &&
OPD_BENCH_SUSTAINED=0 timeout -k 10 300 python bench.py --batch 1 --streams 1 --steps 400 --no-cpu-baseline > $O/bench_batch1.json 2> $O/bench_batch1.err &&
timeout -k 10 400 tools/abl_forward.sh $O/abl 400 > $O/abl_forward.txt 2>&1 &&
du -sh $O
fi
