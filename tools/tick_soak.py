"""tools/tick_soak.py [ticks] -- many path-traced ticks in a row, with the time of every block of 20 (looks for a frame that never ends)"""
import sys, time
sys.path.insert(0, ".")
from raytracer_amd import api
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 1280, 720
r = api.Renderer(W, H, 0)
assert r.L.rth_scene_make_default(r.scene.h, api.ASSETS.encode(), api.REFERENCE_SAH) == 0
r.init()
r.set_mode(api.PATHTRACING)
t0 = time.perf_counter()
for i in range(T):
    r.tick()
    if i % 20 == 19:
        t1 = time.perf_counter()
        print("ticks %d-%d: %.2f ms per tick" % (i - 19, i, (t1 - t0) / 20 * 1e3), flush=True)
        t0 = t1
