"""Would more than two lanes pay?  The bench's workload (configs[2], 64 spp per step) on H handles of one lane each, one host thread
per handle, the steps dealt round-robin -- against one handle with two lanes.  Timing only: the handles' accumulators are not summed
(inside the library a third lane would share the accumulator and the ticket chain).  GPU box only.
usage: python tools/n_lanes.py STEPS "H:slots[:lanes] ..." """
import os
import sys
import threading
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
configs = (sys.argv[2] if len(sys.argv) > 2 else "1:4:2 2:2 3:2 4:2").split()
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
seeds, cam = api.seed_table(W * H), api.default_camera(W, H, scene.scale)
for cfg in configs:
    f = [int(x) for x in cfg.split(":")]
    n, slots, lanes = f[0], f[1], (f[2] if len(f) > 2 else 1)
    devs = []
    for _ in range(n):
        d = api.Device(W, H, lanes=lanes, slots=slots)
        d.upload(scene.desc)
        d.set_seeds(seeds)
        d.set_camera(cam)
        d.render(api.PATHTRACING, 0, 64, 1)
        d.flush(), d.sync(), d.reset(), d.reset_stats()
        devs.append(d)

    def work(k):
        for i in range(k, steps, n):
            devs[k].render(api.PATHTRACING, i * 64, 64, 1)
        devs[k].flush()
        devs[k].sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    rays = sum(d.stats()["extend_rays"] + d.stats()["shadow_rays"] for d in devs)
    print("%d handle(s) x %d lane(s), %d slots: %.1f ms per step, %.0f Mrays/s" % (n, lanes, slots, dt * 1e3 / steps, rays / dt / 1e6), flush=True)
    for d in devs:
        d.close()
