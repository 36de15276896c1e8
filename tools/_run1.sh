timeout -k 10 600 python -m pytest tests/test_workloads_gpu.py -q -m gpu -x -k "alternative_launch_plans" 2>&1 | tail -12 | cut -c1-250
