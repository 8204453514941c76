#!/bin/bash
# session 22: hiprz_rebuild_trees — tests, then config D twisted: refitted trees against trees rebuilt on the device
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 500 python -m pytest tests/test_device_build_gpu.py -x -q > $OUT/s22_tests.log 2>&1 || { tail -30 $OUT/s22_tests.log; exit 1; }
tail -2 $OUT/s22_tests.log
python3 - <<'PY' > $OUT/rebuild_after_deformation_D.txt 2>&1 || { tail -5 $OUT/rebuild_after_deformation_D.txt; exit 1; }
import time, numpy as np
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
preset = scenes.CONFIGS["D"]
w = preset["build"]()
flat, cam = flatten(w), camera_struct(w.camera)
cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
c = Context(0); c.set_tree(3); c.upload_scene(flat); c.upload_camera(cam); c.set_config(cfg)
def step_ms(label):
    c.render(1)
    for _ in range(3): c.render(8)
    c.sync(); t0 = time.perf_counter()
    for _ in range(10): c.render(8)
    c.sync(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{label}: {ms:.3f} ms per step of 8 passes"); return ms
step_ms("device SAH trees as built")
# the big mesh twisted about y: vertices of every triangle record (v1, v2, v3 are absolute in the snapshot)
big = int(np.argmax([0] + [0]))  # (records carry no mesh id: twist all triangles of the biggest contiguous source range = the sphere)
tris, attrs = flat.tris.copy(), flat.tri_attrs.copy()
def twist(v, turns=3.0):
    a = (v[:, 1] * np.float32(turns)).astype(np.float32); co, si = np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)
    return np.stack([co * v[:, 0] + si * v[:, 2], v[:, 1], -si * v[:, 0] + co * v[:, 2]], 1).astype(np.float32)
n_big = 301400
sl = slice(len(tris) - n_big, len(tris)) if len(tris) >= n_big else slice(0, len(tris))
for k in ("v1", "v2", "v3"):
    tris[k][sl] = twist(tris[k][sl])
t0 = time.perf_counter(); c.update_triangles(0, tris, attrs); print(f"update_triangles (refit of all {len(tris)}): {(time.perf_counter()-t0)*1e3:.2f} ms")
refit = step_ms("after the twist, refitted trees")
for tree, name in ((2, "Morton order"), (3, "binned SAH")):
    t0 = time.perf_counter(); c.rebuild_trees(tree); dt = (time.perf_counter() - t0) * 1e3
    print(f"hiprz_rebuild_trees({name}): {dt:.2f} ms"); step_ms(f"after the twist, trees rebuilt on the device ({name})")
print(c.timings())
PY
cat $OUT/rebuild_after_deformation_D.txt | grep -v "^ *$" | head -30
