set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t11.log 2>&1; rc=$?; tail -4 gpurun_out/t11.log; [ $rc = 0 ] || exit 1
for cfg in C D E; do
  timeout -k 10 120 python bench.py --config $cfg --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/b11_$cfg.json 2>gpurun_out/b11_err.log || { echo FAIL; tail -3 gpurun_out/b11_err.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/b11_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", round(d["value"],1), "Mrays/s ms/step", round(d["ms_per_step"],2), "trace", round(d["roofline"]["avg_launch_us"],1), "us frac", round(d["roofline"]["frac"],3), "shade", d["roofline"]["shade_kernel_avg_launch_us"])
PY
done
