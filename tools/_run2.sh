timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "enc_ffn or gemm_ln_deep" 2>&1 | tail -3
timeout -k 10 200 python tools/bench_enc_ffn.py --ablate 2>&1
