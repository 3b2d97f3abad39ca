"""tools/checksum.py -- SHA-1 of the accumulator bits of two fixed renders (default scene 320x180 x 16 spp; 1M-triangle scene
1920x1080 x 4 spp) and of the primary hit maps: two builds that print the same lines compute the same bits."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
from raytracer_amd import api


def run(scene, W, H, spp, scale=1.0):
    d = api.Device(W, H)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.set_camera(api.default_camera(W, H, scale))
    obj, t = d.trace_primary()
    d.render(api.PATHTRACING, 0, spp, 1)
    acc = d.accumulator()
    d.close()
    return hashlib.sha1(acc.tobytes()).hexdigest()[:16], hashlib.sha1(obj.tobytes() + t.tobytes()).hexdigest()[:16]


print("default 320x180 x16:", *run(api.HostScene.default(), 320, 180, 16))
m = api.mesh_scene(1000000, api.BINNED_SAH)
print("mesh 1920x1080 x4:  ", *run(m, 1920, 1080, 4, m.scale))
