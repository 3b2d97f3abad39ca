"""tools/survival.py [spp] -- how many paths of one batch are alive at each bounce (1 M triangles, 1080p): the wavefront loop alone,
no tail, no carry, option log_bounces.  The integral over the late bounces is the work that the end of a render has to finish."""
import sys
sys.path.insert(0, ".")
from raytracer_amd import api
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
d = api.Device(W, H, carry=0, tail_paths=0, log_bounces=1)
d.upload(scene.desc)
d.set_seeds(api.seed_table(W * H))
d.set_camera(api.default_camera(W, H, scene.scale))
d.render(api.PATHTRACING, 0, spp, 1)
d.flush()
d.sync()
st = d.stats()
print("extend rays", st["extend_rays"], "walkers", st["extend_walkers"], "shadow rays", st["shadow_rays"], "walkers", st["shadow_walkers"])
d.close()
