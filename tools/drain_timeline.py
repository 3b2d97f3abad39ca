"""What the end of a render is made of: the bench's workload (configs[2], 20 steps of 64 spp, two lanes), then rtmi_flush with option
log_bounces: the library prints the survivors of every draining round with the host clock (it synchronises per round, so the
rounds' own durations show; the whole is a little slower than an undisturbed drain).  GPU box only.
usage: python tools/drain_timeline.py [steps] [lanes]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
d = api.Device(W, H, lanes=lanes)
d.upload(scene.desc)
d.set_seeds(api.seed_table(W * H))
d.set_camera(api.default_camera(W, H, scene.scale))
d.render(api.PATHTRACING, 0, 64, 1)
d.flush(), d.sync()
d.reset()
t0 = time.perf_counter()
for i in range(steps):
    d.render(api.PATHTRACING, i * 64, 64, 1)
d.sync_streams() if hasattr(d, "sync_streams") else None
t1 = time.perf_counter()
d.set_option("log_bounces", 1)
print("enqueued in %.1f ms; draining (host clock %.1f ms)" % ((t1 - t0) * 1e3, (time.monotonic() * 1e3) % 1e6), file=sys.stderr, flush=True)
d.flush()
d.sync()
t2 = time.perf_counter()
print("drain with per-round synchronisation: %.1f ms" % ((t2 - t1) * 1e3), file=sys.stderr)
print(d.stats()["dropped_paths"])
