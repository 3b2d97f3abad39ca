#!/usr/bin/env python3
"""tools/bounce_profile.py -- live path count per bounce of one 4-spp batch of the bench scene (RTMI_LOG_BOUNCES)"""
import os, sys
os.environ["RTMI_LOG_BOUNCES"] = "1"
os.environ["RTMI_TAIL_PATHS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api
W, H = 1920, 1080
s = api.mesh_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
d = api.Device(W, H)
d.upload(s.desc); d.set_seeds(api.seed_table(W * H)); d.set_camera(api.default_camera(W, H, s.scale))
d.render(api.PATHTRACING, 0, 4, 1)
d.sync()
print(d.stats()["bounce_launches"])
