set -o pipefail
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
MASTER_ADDR=127.0.0.1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --rehearse-on-one-gpu --verify-gather > gpurun_out/rehearse.log 2>&1 || { tail -20 gpurun_out/rehearse.log; exit 1; }
tail -2 gpurun_out/rehearse.log | cut -c 1-600
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
cut -c 1-400 gpurun_out/bench_default.json
