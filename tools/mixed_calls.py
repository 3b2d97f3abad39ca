import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import api
W, H = 1920, 1080
scene = api.mesh_scene(100000, api.BINNED_SAH)
seeds = api.seed_table(W * H)
cam = api.default_camera(W, H, scene.scale)
def run(calls, **o):
    d = api.Device(W, H, **o)
    d.upload(scene.desc); d.set_seeds(seeds); d.set_camera(cam)
    b = 0
    for n in calls:
        d.render(api.PATHTRACING, b, n, 1); b += n
    img = d.accumulator()
    st = d.stats()
    d.close()
    return img.view(np.uint32), st
ref, st0 = run([67], lanes=1)
for calls, o in (([67], dict(lanes=2)), ([32, 32, 1, 1, 1], dict(lanes=2)), ([1, 33, 1, 32], dict(lanes=2)), ([5, 40, 22], dict(lanes=1)), ([64, 3], dict(lanes=2, batch=7)),
                 ([1, 1, 33, 1, 30, 1], dict(lanes=2, preview_reads=1)), ([1] * 12 + [55], dict(lanes=1, preview_reads=1))):
    img, st = run(calls, **o)
    print(calls, o, "bit-equal" if np.array_equal(img, ref) else "DIFFERENT %d" % int((img != ref).sum()), st["extend_rays"] == st0["extend_rays"], flush=True)
