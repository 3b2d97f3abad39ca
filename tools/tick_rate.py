"""tools/tick_rate.py [W H [ticks]] -- frame time of the interactive loop (Renderer::Tick: 1 spp + Finalize per frame,
renderer.cpp:524-556) on the default scene, per visualization mode.  Run on the GPU box."""
import sys
import time

sys.path.insert(0, ".")
from raytracer_amd import api

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40

r = api.Renderer(W, H, 0)
assert r.L.rth_scene_make_default(r.scene.h, api.ASSETS.encode(), api.REFERENCE_SAH) == 0
r.init()
for name, mode in (("RayTracing", api.RAYTRACING), ("PathTracing", api.PATHTRACING)):
    r.set_mode(mode)
    for _ in range(3):
        r.tick()
    t0 = time.perf_counter()
    for _ in range(T):
        r.tick()
    ms = (time.perf_counter() - t0) / T * 1e3
    print("%s %dx%d: %.2f ms per tick (%.1f fps), %d frames accumulated" % (name, W, H, ms, 1e3 / ms, r.frames))
