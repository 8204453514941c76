"""Run by tests/test_full_size_gpu.py in a child process: batches of passes replayed from a captured graph, interleaved with ANOTHER
user of the process's HIP state — torch allocating tensors between them.  A replay of a graph that contained the library radix
sort (hipcub) faulted in this situation (round 2); the sort is hand-written since (hiprz_sort.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
def P(*a): print(*a, flush=True)
S = set(os.environ.get("SYNCS", "").split(","))
torch.cuda.set_device(0)
preset = scenes.CONFIGS[os.environ.get("CFG", "C")]; w = preset["build"](); flat, cam = flatten(w), camera_struct(w.camera)
ctx = Context(0)
ctx.upload_scene(flat); ctx.upload_camera(cam); ctx.set_config(RenderConfig(tracing=Tracing(8, 8)).struct())
if os.environ.get("NOGRAPH"): ctx.set_graph(False)
if os.environ.get("NOSORT"): ctx.set_ray_sort(0)
if os.environ.get("PIPELINE"): ctx.set_pipeline(int(os.environ["PIPELINE"]))
if os.environ.get("CFG"): pass
def fence(): ctx.sync(); torch.cuda.synchronize()
ctx.render(1)
if "a" in S: ctx.sync()
ctx.render(8)
if "b" in S: ctx.sync()
ctx.tonemap()
fence(); P("warm ok")
ctx.kernel_time_ms()
if "c" not in S: a = ctx.read_accum()
fence()
for rep in range(3):
    fence()
    for k in range(5):
        ctx.render(8); ctx.tonemap()
    fence(); P("repeat ok", rep)
    if "d" not in S: t = torch.tensor([1.0], dtype=torch.float64, device="cuda"); t.item()
P("done", "graph captures", ctx.graph_captures())
