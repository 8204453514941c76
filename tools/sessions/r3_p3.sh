#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
echo "== every shard r of N, kernel side (render + tone map + tile export), B C D E"
timeout -k 10 900 python tools/shard_scaling.py --config B,C,D,E --shards 1,2,4,8 --steps 10 > $OUT/shards_all.jsonl 2> $OUT/shards_all.err || { tail -5 $OUT/shards_all.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_all.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
echo "== D with SAH trees (same frames)"
timeout -k 10 300 python tools/shard_scaling.py --config D --shards 1,8 --steps 10 --tree 1 > $OUT/shards_D_sah.jsonl 2>> $OUT/shards_all.err; cut -c 1-200 $OUT/shards_D_sah.jsonl
echo "== compat integrator on E: all behaviours (63), all but the coloured shadow mask (59)"
for m in 63 59; do
timeout -k 10 300 python bench.py --config E --mode $m --steps 3 --warmup 1 --repeats 3 --min-seconds 0 --no-cpu-baseline --streams 1 > $OUT/bench_E_mode$m.json 2>> $OUT/bench_p3.err || { tail -3 $OUT/bench_p3.err; }
python3 -c "import json; d=json.load(open('$OUT/bench_E_mode$m.json')); r=d['roofline']; print('mode $m:', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step', d['config']['pipeline'], r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us', 'shade+shadow', round(r['shade_kernel_avg_launch_us'] or 0,1))"
done
echo "== device build of config D: with and without the host's validation of the downloaded tables"
python3 - <<'PY'
import os, time
from rayzath_amd import scenes
from rayzath_amd.engine import Context
from rayzath_amd.scene import camera_struct, flatten
w = scenes.CONFIGS["D"]["build"]()
flat, cam = flatten(w), camera_struct(w.camera)
for trust in ("", "1"):
    if trust: os.environ["HIPRZ_TRUST_DEVICE_TREES"] = "1"
    for tree in (1, 2):
        c = Context(0); c.set_tree(tree)
        t0 = time.perf_counter(); c.upload_scene(flat); dt = time.perf_counter() - t0
        c.upload_camera(cam)
        if tree == 2:
            t0 = time.perf_counter(); c.update_triangles(0, flat.tris, flat.tri_attrs); rf = time.perf_counter() - t0
            t0 = time.perf_counter(); c.update_instances(flat.instances); ri = time.perf_counter() - t0
        print(f"trust={trust or 0} tree={tree}: upload_scene {dt*1e3:.1f} ms" + (f", refit of all 301 400 triangles {rf*1e3:.2f} ms, update_instances {ri*1e3:.2f} ms" if tree == 2 else ""))
        print("   " + " | ".join(l.strip() for l in c.timings().splitlines() if "tree" in l))
        c.close()
PY
