import os, sys
os.environ["RTMI_LOG_BOUNCES"] = "1"
sys.path.insert(0, os.getcwd())
from raytracer_amd import api
W, H = 1920, 1080
s = api.mesh_scene(1000000)
d = api.Device(W, H)
d.upload(s.desc); d.set_seeds(api.seed_table(W * H)); d.set_camera(api.default_camera(W, H, s.scale))
d.render(api.PATHTRACING, 0, 48, 1)
d.sync()
a = d.accumulator()
print("done", a.mean())
