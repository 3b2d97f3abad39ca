"""tools/fly_soak.py [frames] -- the 1 M-triangle scene with a camera that moves every frame: reset, one sample, finalize (the
reference's Tick while a key is held, renderer.cpp:622-633).  PREVIEW=1: option preview_reads."""
import os
import sys
import time

sys.path.insert(0, ".")
from raytracer_amd import api  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
d = api.Device(W, H, preview_reads=int(os.environ.get("PREVIEW", "0")))
d.upload(scene.desc)
d.set_seeds(api.seed_table(W * H))
t0 = time.perf_counter()
for i in range(T):
    d.reset()
    d.set_camera(api.zoom_camera(api.default_camera(W, H, scene.scale), 1.0 + 0.001 * (i % 100)))
    d.render(api.PATHTRACING, i, 1, 1)
    d.finalize(1)
    if i % 50 == 49:
        t1 = time.perf_counter()
        print("frames %d-%d: %.2f ms per frame" % (i - 49, i, (t1 - t0) / 50 * 1e3), flush=True)
        t0 = t1
d.close()
