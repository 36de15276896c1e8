set -o pipefail
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_workloads_gpu.py -q -m gpu -k "native_exchange or frame_split" > $O/tests_x.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests_x.log | cut -c1-220
bash tools/dec_cost.sh $O 2>&1 | grep -v amdgpu
