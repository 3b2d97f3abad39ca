"""tools/drain_probe.py [tris W H spp] -- the end of a render: how long the caller waits for the last paths after the bulk.

Renders `spp` samples of the mesh scene, then times rtmi_flush + rtmi_synchronize (the drain: the late paths that are
still alive get a batch of their own, then the k_tail / k_tail_team rounds), for the one-lane-per-path tail and for the
four-lane teams, and prints where a path-bounce of the draining rounds spends its time (option tail_profile: 100 MHz
ticks summed over the lanes that lead a path).  Run on the GPU box.
"""
import sys
import time

sys.path.insert(0, ".")
from raytracer_amd import api

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 16

scene = api.mesh_scene(tris, api.BINNED_SAH) if tris > 0 else api.HostScene.default()
seeds = api.seed_table(W * H)
cam = api.default_camera(W, H, scene.scale)
for label, opts in (("k_tail (one lane per path)", dict(tail_team=0, tail_wave=0)), ("k_tail_team (four lanes per path)", dict(tail_team=1, tail_wave=0)),
                    ("k_tail_team + k_tail_wave (default)", dict()),
                    ("k_tail, profiled", dict(tail_team=0, tail_wave=0, tail_profile=1)), ("k_tail_team, profiled", dict(tail_team=1, tail_wave=0, tail_profile=1)),
                    ("k_tail_wave, profiled", dict(tail_team=0, tail_wave=1, wave_below=1 << 14, tail_profile=1))):
    d = api.Device(W, H, **opts)
    d.upload(scene.desc)
    d.set_seeds(seeds)
    d.set_camera(cam)
    times = []
    for rep in range(3):
        d.reset()
        d.sync()
        t0 = time.perf_counter()
        d.render(api.PATHTRACING, rep * spp, spp, 1)
        d.sync()  # the bulk and the batches' own first tail rounds are done; the late paths wait in the carry buffer
        d.reset_stats()
        t1 = time.perf_counter()
        d.flush()
        d.sync()
        t2 = time.perf_counter()
        times.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    st = d.stats()
    p = st["tail_profile"]
    print("%-36s render+sync %6.1f ms, then flush+sync %6.1f ms (runs: %s)" % (label, times[-1][0], times[-1][1], " ".join("%.0f" % t[1] for t in times)))
    if p[4]:
        us = lambda x: x / 100.0 / p[4]
        print("    per path-bounce (us): analytic %.2f, closest walk %.2f, shade %.2f, shadow %.2f | %d path-bounces, %.1f node steps and %.1f leaf tests per bounce"
              % (us(p[0]), us(p[1]), us(p[2]), us(p[3]), p[4], p[5] / p[4], p[6] / p[4]))
        if p[7]:
            print("    clock of the rounds: %.0f MHz" % (p[8] / p[7] * 100.0))
        if p[11] and p[10]:
            print("    closest-hit node step: %.0f cycles, of which %.0f waiting for the node record (%d steps)" % (p[10] / p[11], p[9] / p[11], p[11]))
        elif p[11]:
            print("    closest-hit walk: %.1f wave iterations per bounce" % (p[11] / p[4]))
    d.close()
