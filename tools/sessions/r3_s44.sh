#!/bin/bash
# session 44: the per-wave resident kernel on 4K frames of scenes without lights (129 600 waves) against the split pipeline
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for lim in 40000 400000; do
HIPRZ_WAVE_RESIDENT_MAX=$lim HIPRZ_TRUST_DEVICE_TREES=1 python3 - <<'PY'
import os, time
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
for name, build in (("C at 4K", lambda: scenes.cornell_sphere(3840, 2160, 80)), ("D at 4K", lambda: scenes.textured_sphere_scene(3840, 2160, 550))):
    w = build(); flat, cam = flatten(w), camera_struct(w.camera)
    c = Context(0); c.set_tree(4); c.upload_scene(flat); c.upload_camera(cam); c.set_config(RenderConfig(tracing=Tracing(8, 8)).struct())
    c.render(1)
    for _ in range(3): c.render(8)
    c.sync(); t0 = time.perf_counter()
    for _ in range(8): c.render(8)
    c.sync(); ms = (time.perf_counter() - t0) / 8 * 1e3
    print(f"limit {os.environ['HIPRZ_WAVE_RESIDENT_MAX']}: {name}: pipeline {c.pipeline()} {ms:.3f} ms per step, {8 * 3840 * 2160 / ms / 1e3:.1f} Mrays/s", flush=True)
    c.close()
PY
done
