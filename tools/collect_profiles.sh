#!/bin/bash
# Copy the summaries of tools/final_artifacts.sh (gpurun_out/final/) into profiles/ under this round's names.
# usage: tools/collect_profiles.sh r02
set -e
R=${1:?round tag, e.g. r02}
O=gpurun_out/final
P=profiles
tail -n 1 $O/bench.json > $P/${R}_bench_default_k200.json
tail -n 1 $O/bench_k20.json > $P/${R}_bench_default_k20.json
cp $(ls $O/prof_default/*/*_kernel_stats.csv | tail -n 1) $P/${R}_bench_default_kernel_stats.csv
cp $(ls $O/prof_streams1/*/*_kernel_stats.csv | tail -n 1) $P/${R}_bench_streams1_kernel_stats.csv
python tools/timeline.py $(ls $O/prof_streams1/*/*_kernel_trace.csv | tail -n 1) --all > $P/${R}_bench_streams1_timeline.txt
python tools/pmc_traffic.py $(ls $O/pmc_fetch/*/*_counter_collection.csv | tail -n 1) $(ls $O/pmc_write/*/*_counter_collection.csv | tail -n 1) $P/${R}_pmc_traffic.json
tail -n 1 $O/bench_gloo2.json > $P/${R}_bench_gloo2_selflaunch_rehearsal.json
tail -n 1 $O/bench_r101_1066x1920.json > $P/${R}_bench_r101_1066x1920.json
tail -n 1 $O/bench_r50_tile1080p_b4.json > $P/${R}_bench_r50_tile1080p_b4.json
cp $O/host_rate.txt $P/${R}_host_boundary_rate.txt
cp $O/trace_gemm.txt $P/${R}_trace_gemm.txt
cp $O/bench_layers.txt $P/${R}_bench_layers.txt
cp $O/bench_btail.txt $P/${R}_bench_btail.txt
cp $O/drift_toggles.txt $P/${R}_drift_toggles.txt
cp $O/smoke.log $P/${R}_smoke_parity.txt
cp gpurun_out/parity_table.json $P/${R}_parity_table.json
ls -la $P | grep ${R}_
