"""tools/drain_one.py [tail_team] -- one 16-spp render of the 1M-triangle scene at 1080p, twice, then the drain (for tools/drain_trace.sh)"""
import sys
sys.path.insert(0, ".")
from raytracer_amd import api
team = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # 0: k_tail only, 1: + k_tail_team, 2: + k_tail_wave (the default)
W, H = 1920, 1080
scene = api.mesh_scene(1000000, api.BINNED_SAH)
d = api.Device(W, H, tail_team=int(team >= 1), tail_wave=int(team >= 2))
d.upload(scene.desc)
d.set_seeds(api.seed_table(W * H))
d.set_camera(api.default_camera(W, H, scene.scale))
for rep in range(2):
    d.reset()
    d.render(api.PATHTRACING, rep * 16, 16, 1)
    d.sync()
    d.flush()
    d.sync()
d.close()
