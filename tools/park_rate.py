"""tools/park_rate.py -- how many walks are parked and resumed (tail control of k_trace) on the bench scene"""
import sys
sys.path.insert(0, ".")
from raytracer_amd import api
W, H = 1920, 1080
m = api.mesh_scene(1000000, api.BINNED_SAH)
d = api.Device(W, H)
d.upload(m.desc)
d.set_seeds(api.seed_table(W * H))
d.set_camera(api.default_camera(W, H, m.scale))
d.render(api.PATHTRACING, 0, 64, 1)
d.accumulator()
st = d.stats()
print("walkers: extend %d shadow %d; parked %d (%.1f %% of walkers)" % (st["extend_walkers"], st["shadow_walkers"], st["suspended"], 100.0 * st["suspended"] / (st["extend_walkers"] + st["shadow_walkers"])))
