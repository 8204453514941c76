#!/usr/bin/env python3
"""On the GPU box: the cooperative-triangle-phase walk (HIPRZ_COOP=1) against the plain front-to-back walk, bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rayzath_amd import scenes
from rayzath_amd.engine import Context, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

cases = ((scenes.cornell_sphere(160, 96, resolution=40), (1, 1)), (scenes.textured_sphere_scene(160, 96, resolution=60, map_size=64), (1, 1)),
         (scenes.living_room(128, 80, 16), (2, 2)), (scenes.textured_sphere_scene(320, 200, resolution=200, map_size=64), (1, 1)))
ok = True
for world, samples in cases:
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(*samples), Tracing(6, 4)).struct()
    out = []
    for coop in ("0", "1"):
        os.environ[sys.argv[1] if len(sys.argv) > 1 else "HIPRZ_COOP"] = coop
        c = Context(0)
        c.set_traversal_mode(3), c.set_lds_scene(0), c.set_walk_order(2)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        counters = c.render_counted(2)
        c.render(6), c.render(4)
        out.append((c.read_accum(), c.read_depth(), c.read_state(), counters))
        c.close()
    same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and all(np.array_equal(out[0][2][k], out[1][2][k]) for k in out[0][2])
    print("tris", len(flat.tris), "same frame:", same, "same counters:", out[0][3] == out[1][3], {k: (out[0][3][k], out[1][3][k]) for k in ("box_tests", "tri_tests")})
    ok = ok and same and out[0][3] == out[1][3]
sys.exit(0 if ok else 1)
