#!/bin/bash
# session 23: upload of config D with device-built trees, the host's proof on 8 threads
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 300 python -m pytest tests/test_device_build_gpu.py -x -q > $OUT/s23_tests.log 2>&1 || { tail -30 $OUT/s23_tests.log; exit 1; }
tail -1 $OUT/s23_tests.log
python3 - <<'PY' > $OUT/device_build_upload_D.txt 2>&1 || { tail -5 $OUT/device_build_upload_D.txt; exit 1; }
import os, time
from rayzath_amd import scenes
from rayzath_amd.engine import Context
from rayzath_amd.scene import camera_struct, flatten
w = scenes.CONFIGS["D"]["build"]()
flat = flatten(w)
print("config D", len(flat.tris), "triangles")
for trust in ("", "1"):
    if trust: os.environ["HIPRZ_TRUST_DEVICE_TREES"] = "1"
    else: os.environ.pop("HIPRZ_TRUST_DEVICE_TREES", None)
    for tree in (0, 1, 2, 3, 4):
        for rep in range(2):
            c = Context(0); c.set_tree(tree)
            t0 = time.perf_counter(); c.upload_scene(flat); dt = time.perf_counter() - t0
            if rep:
                print(f"trust={trust or 0} tree={tree}: upload_scene {dt*1e3:.1f} ms   " + " | ".join(l.strip() for l in c.timings().splitlines() if "tree" in l))
            c.close()
PY
cat $OUT/device_build_upload_D.txt
