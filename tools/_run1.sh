set -o pipefail
R=$PWD; O=$R/gpurun_out/r04v; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests_all.log | cut -c1-300
