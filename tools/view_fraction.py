"""tools/view_fraction.py -- share of camera rays whose nearest hit is a triangle, for a few (zoom, tube radius) views (bench.py --zoom/--tube)"""
import sys
sys.path.insert(0, ".")
from raytracer_amd import api
W, H = 1920, 1080
for tube in (0.075, 0.15):
    scene = api.mesh_scene(1000000, api.BINNED_SAH, tube=tube)
    info = scene.info()
    na = info["quads"] + info["spheres"] + info["cubes"] + info["planes"]
    d = api.Device(W, H)
    d.upload(scene.desc)
    for zoom in (1.0, 0.5, 0.35, 0.25, 0.15):
        d.set_camera(api.zoom_camera(api.default_camera(W, H, scene.scale), zoom))
        print("tube %.3f zoom %.2f: mesh fraction %.3f" % (tube, zoom, float((d.trace_primary()[0] >= na).mean())), flush=True)
    d.close()
