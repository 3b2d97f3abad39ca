"""tools/two_lanes.py [steps] -- would two batches in flight at once (one's k_shade beside the other's walks) be faster?
Experiment without library changes: two handles on one GPU, each driven by its own host thread (ctypes releases the GIL),
each rendering half of every step's samples, against one handle rendering all of them.  1 M triangles, 1080p, 64 spp/step."""
import sys, time, threading
sys.path.insert(0, ".")
from raytracer_amd import api
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
seeds = api.seed_table(W * H)
cam = api.default_camera(W, H, scene.scale)

def make(**o):
    d = api.Device(W, H, **o)
    d.upload(scene.desc); d.set_seeds(seeds); d.set_camera(cam)
    return d

def run(devs, spp_each, label):
    for d in devs: d.reset(); d.render(api.PATHTRACING, 0, spp_each, 1)
    for d in devs: d.flush(); d.sync(); d.reset(); d.reset_stats()
    def work(d, k):
        for s in range(steps): d.render(api.PATHTRACING, (len(devs) * s + k) * spp_each, spp_each, 1)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(d, k)) for k, d in enumerate(devs)]
    for t in th: t.start()
    for t in th: t.join()
    t1 = time.perf_counter()
    for d in devs: d.flush()
    for d in devs: d.sync()
    t2 = time.perf_counter()
    rays = sum(d.stats()["extend_rays"] + d.stats()["shadow_rays"] for d in devs)
    print("%-44s %7.1f ms/step enqueue, %7.1f ms/step with the drain (%5.0f ms), %6.0f Mrays/s" % (label, (t1 - t0) * 1e3 / steps, (t2 - t0) * 1e3 / steps, (t2 - t1) * 1e3, rays / (t2 - t0) / 1e6), flush=True)

import os
n = int(os.environ.get("HANDLES", "2"))
opts = dict(lanes=int(os.environ.get("LANES", "1")))
d = make(**opts); run([d], 64, "one handle (lanes=%d), batches of 16 spp" % opts["lanes"]); d.close()
hs = [make(**opts) for _ in range(n)]; run(hs, 64 // n, "%d handles (lanes=%d), batches of 16 spp" % (n, opts["lanes"]))
for h in hs: h.close()
