#!/bin/bash
# session 12: device surface-area build — build times (with / without host proof), C under each tree, every shard of 8 under tree 3
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
echo "== device build of config D and E: Morton order (2) and binned surface area (3), with and without the host's proof of the downloaded tables"
python3 - <<'PY' > $OUT/device_build_sah.txt 2>&1 || { tail -5 $OUT/device_build_sah.txt; exit 1; }
import os, time
from rayzath_amd import scenes
from rayzath_amd.engine import Context
from rayzath_amd.scene import camera_struct, flatten
for cfg in ("D", "E"):
    w = scenes.CONFIGS[cfg]["build"]()
    flat, cam = flatten(w), camera_struct(w.camera)
    print("config", cfg, len(flat.tris), "triangles", len(flat.instances), "instances")
    for trust in ("", "1"):
        if trust: os.environ["HIPRZ_TRUST_DEVICE_TREES"] = "1"
        else: os.environ.pop("HIPRZ_TRUST_DEVICE_TREES", None)
        for tree in (1, 2, 3):
            for rep in range(2):
                c = Context(0); c.set_tree(tree)
                t0 = time.perf_counter(); c.upload_scene(flat); dt = time.perf_counter() - t0
                if rep: 
                    print(f"trust={trust or 0} tree={tree}: upload_scene {dt*1e3:.1f} ms")
                    print("   " + " | ".join(l.strip() for l in c.timings().splitlines() if "tree" in l))
                c.close()
PY
cat $OUT/device_build_sah.txt
echo "== C on each kind of tree"
for tree in 0 1 3; do
  timeout -k 10 200 python3 bench.py --config C --tree $tree --no-cpu-baseline --min-seconds 2 > $OUT/s12_bench_C_tree$tree.json 2> $OUT/s12_bench_C_tree$tree.err || { tail -5 $OUT/s12_bench_C_tree$tree.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/s12_bench_C_tree$tree.json')); print('C tree $tree', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step single', round(d['single_stream']['value'],1))"
done
echo "== every shard of 1 and 8 under tree 3"
timeout -k 10 600 python tools/shard_scaling.py --config C,D,E --shards 1,8 --steps 10 --tree 3 > $OUT/shards_device_sah.jsonl 2> $OUT/shards_device_sah.err || { tail -5 $OUT/shards_device_sah.err; exit 1; }
cut -c 1-330 $OUT/shards_device_sah.jsonl
