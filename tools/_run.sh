set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "sharded or rehearsal or streams" > gpurun_out/t10.log 2>&1; rc=$?; tail -8 gpurun_out/t10.log; [ $rc = 0 ] || exit 1
