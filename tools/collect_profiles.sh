#!/bin/bash
# Copy the summaries of tools/final_artifacts.sh (gpurun_out/final/: made on the GPU box, the raw traces stay there) into profiles/ under this
# round's names.   usage: tools/collect_profiles.sh r05
set -e
R=${1:?round tag, e.g. r05}
O=gpurun_out/final
P=profiles
tail -n 1 $O/bench.json > $P/${R}_bench_default_k1000.json
tail -n 1 $O/bench_k20.json > $P/${R}_bench_default_k20.json
for f in bench_default_kernel_stats.csv bench_streams1_kernel_stats.csv bench_streams1_timeline.txt overlap_three_streams.txt pmc_traffic.json pmc_sq.json pmc_sq_per_kernel.txt parity_table.json; do
  [ -f $O/$f ] && cp $O/$f $P/${R}_$f
done
for f in bench_gloo2:bench_gloo2_selflaunch_rehearsal bench_r101_1066x1920:bench_r101_1066x1920 bench_r50_tile1080p_b4:bench_r50_tile1080p_b4 bench_bf16:bench_bf16 bench_forced_comm_1rank:bench_forced_comm_1rank bench_batch1:bench_batch1_streams1; do
  [ -f $O/${f%%:*}.json ] && tail -n 1 $O/${f%%:*}.json > $P/${R}_${f##*:}.json
done
for f in host_rate:host_boundary_rate trace_gemm:trace_gemm bench_layers:bench_layers bench_btail:bench_btail bench_btail3:bench_btail3 trace_btail3:trace_btail3 bench_attn:bench_attn trace_attn:trace_attn bench_gemm_ln:bench_gemm_ln bench_enc_ffn:bench_enc_ffn bench_dec:bench_dec_final microbench_vmorder:microbench_vmorder host_rate_ref_pattern:host_rate_ref_pattern host_b1_probe:host_b1_probe microbench_mfma_peak:microbench_mfma_peak abl_forward:abl_forward; do
  [ -f $O/${f%%:*}.txt ] && cp $O/${f%%:*}.txt $P/${R}_${f##*:}.txt
done
cp $O/smoke.log $P/${R}_smoke_parity.txt
ls -la $P | grep ${R}_
