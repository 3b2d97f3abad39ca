import sys
sys.path.insert(0, ".")
from raytracer_amd import api
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
d = api.Device(W, H, log_bounces=1, lanes=1)
d.upload(scene.desc); d.set_seeds(api.seed_table(W * H)); d.set_camera(api.default_camera(W, H, scene.scale))
d.render(api.PATHTRACING, 0, 48, 1)
d.flush(); d.sync()
d.close()
