#!/usr/bin/env python3
"""tools/make_golden.py -- generates tests/golden/*.npz from the REFERENCE itself (dev container only).

Every expected value below is produced by oracle/_ref/libref_<W>x<H>.so, i.e. the reference's own primitive.h /
bvh.h / template/scene.h / camera.h / renderer.cpp / postprocessing.h compiled where they lie by
oracle/ref_build/build_ref.sh.  The inputs are seeded numpy data or the reference's default scene.
The fixtures are data only (inputs + expected outputs); they are what the GPU box checks against, because
/root/reference does not exist there.

Usage:  python tools/make_golden.py            (builds the two reference libraries if needed, then writes fixtures)
One subprocess per resolution: the reference fixes SCRWIDTH/SCRHEIGHT at compile time (camera.h:4-5).
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def rand_dirs(rng, n):
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return d.astype(np.float32)


def room_rays(rng, n):
    """incoherent rays starting inside the default room (x in -3..2.99, y in -1..2, z in -3..3.99)"""
    O = np.stack([rng.uniform(-2.9, 2.9, n), rng.uniform(-0.95, 1.95, n), rng.uniform(-2.9, 3.9, n)], 1).astype(np.float32)
    return O, rand_dirs(rng, n)


def unit_vectors(r):
    """primitive-level vectors: random + adversarial (edge-on, zero direction components, shared origins, t-limits)"""
    rng = np.random.default_rng(20240611)
    out = {}
    n = 12000
    # triangles: random triangles around the origin, rays aimed near them
    v9 = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
    O = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    tgt = (v9.reshape(n, 3, 3) * rng.dirichlet((1, 1, 1), n).astype(np.float32)[:, :, None]).sum(1)
    tgt += rng.normal(scale=0.15, size=(n, 3)).astype(np.float32) * (rng.random((n, 1)) < 0.5)
    D = tgt - O
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    D = D.astype(np.float32)
    # adversarial block: axis-aligned triangles and rays with exact zero components / edge hits
    k = 2000
    v9[:k] = np.round(v9[:k] * 4) / 4
    D[:k] = np.round(D[:k] * 2) / 2
    D[:k][np.all(D[:k] == 0, axis=1)] = (0, 0, 1)
    O[:k] = np.round(O[:k] * 2) / 2
    tin = np.full(n, 1e34, np.float32)
    tin[k:2 * k] = rng.uniform(0.5, 4, k).astype(np.float32)  # finite ray.t
    hit, tout = r.tri_intersect(O, D, tin, v9)
    N, cen = r.tri_derived(v9)
    out.update(tri_O=O, tri_D=D, tri_tin=tin, tri_v9=v9, tri_hit=hit, tri_t=tout, tri_N=N, tri_centroid=cen)
    # spheres
    pr = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0.05, 1.0, (n, 1))], 1).astype(np.float32)
    O = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    O[: n // 4] = pr[: n // 4, :3] + rng.uniform(-0.3, 0.3, (n // 4, 3)).astype(np.float32) * pr[: n // 4, 3:]  # inside the sphere
    D = pr[:, :3] + rng.normal(scale=0.5, size=(n, 3)).astype(np.float32) * pr[:, 3:] - O
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    D = D.astype(np.float32)
    tin = np.where(rng.random(n) < 0.3, rng.uniform(0.2, 3, n), 1e34).astype(np.float32)
    hit, tout = r.sphere_intersect(O, D, tin, pr)
    out.update(sph_O=O, sph_D=D, sph_tin=tin, sph_pr=pr, sph_hit=hit, sph_t=tout)
    # planes
    nd = np.concatenate([rand_dirs(rng, n), rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    nd[:600, :3] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 600)] * rng.choice([-1, 1], (600, 1))
    O = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    D = rand_dirs(rng, n)
    D[:300] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)]  # parallel / perpendicular cases
    tin = np.where(rng.random(n) < 0.3, rng.uniform(0.2, 6, n), 1e34).astype(np.float32)
    hit, tout = r.plane_intersect(O, D, tin, nd)
    out.update(pl_O=O, pl_D=D, pl_tin=tin, pl_nd=nd, pl_hit=hit, pl_t=tout)
    # quad / cube of the default scene
    O, D = room_rays(rng, n)
    D[: n // 2] = (np.array([0, 1.7, 2], np.float32) + rng.normal(scale=1.2, size=(n // 2, 3)).astype(np.float32)) - O[: n // 2]
    D[: n // 2] /= np.linalg.norm(D[: n // 2], axis=1, keepdims=True)
    tin = np.where(rng.random(n) < 0.3, rng.uniform(0.2, 6, n), 1e34).astype(np.float32)
    hit, tout = r.quad_intersect(O, D, tin)
    out.update(quad_O=O, quad_D=D.astype(np.float32), quad_tin=tin, quad_hit=hit, quad_t=tout)
    O, D = room_rays(rng, n)
    D[: n // 2] = (np.array([1.4, 0, 2], np.float32) + rng.normal(scale=0.5, size=(n // 2, 3)).astype(np.float32)) - O[: n // 2]
    D[: n // 2] /= np.linalg.norm(D[: n // 2], axis=1, keepdims=True)
    O[:200] = np.array([1.4, 0, 2], np.float32) + rng.uniform(-0.2, 0.2, (200, 3)).astype(np.float32)  # inside the cube
    hit, tout = r.cube_intersect(O, D.astype(np.float32), tin)
    out.update(cube_O=O, cube_D=D.astype(np.float32), cube_tin=tin, cube_hit=hit, cube_t=tout)
    # slab tests, including zero direction components and planes through the origin coordinate
    bmin = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    bmax = bmin + rng.uniform(0, 2, (n, 3)).astype(np.float32)
    O = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    D = rand_dirs(rng, n)
    k = 3000
    zero = rng.integers(0, 3, k)
    D[np.arange(k), zero] = 0
    D[:k] /= np.linalg.norm(D[:k], axis=1, keepdims=True)
    sel = rng.random(k) < 0.6
    side = rng.random(k) < 0.5
    O[np.arange(k)[sel], zero[sel]] = np.where(side[sel], bmin[np.arange(k)[sel], zero[sel]], bmax[np.arange(k)[sel], zero[sel]])
    bmax[k:k + 500] = bmin[k:k + 500]  # flat boxes
    bmax[k:k + 500, rng.integers(0, 3, 500)] += 1
    tin = np.where(rng.random(n) < 0.5, rng.uniform(0.1, 5, n), 1e34).astype(np.float32)
    a, b, c = r.aabb(O, D.astype(np.float32), tin, bmin, bmax)
    out.update(box_O=O, box_D=D.astype(np.float32), box_tin=tin, box_min=bmin, box_max=bmax, box_scalar=a, box_stack=b, box_simd=c)
    # RNG
    out.update(rng_seeds=r.seeds(4096), rng_floats=r.rng_floats(0x12345678, 4096),
               rng_sample_seed_in=np.array([[1, 0], [0x12345678, 7], [0xFFFFFFFF, 3], [12345, 1023], [0x9E3779B9, 1]], np.uint32))
    out["rng_sample_seed_out"] = np.array([r.sample_seed(int(a), int(b)) for a, b in out["rng_sample_seed_in"]], np.uint32)
    # light / hemisphere sampling with explicit streams
    s0 = rng.integers(1, 2**32 - 1, 512, dtype=np.uint64).astype(np.uint32)
    lp = np.zeros((512, 10), np.float32)
    ls1 = np.zeros(512, np.uint32)
    Ns = rand_dirs(rng, 512)
    dr = np.zeros((512, 3), np.float32)
    ds1 = np.zeros(512, np.uint32)
    for i in range(512):
        lp[i], ls1[i] = r.random_point_on_light(int(s0[i]))
        dr[i], ds1[i] = r.diffuse_reflection(int(s0[i]), Ns[i])
    out.update(smp_s0=s0, smp_light=lp, smp_light_s1=ls1, smp_N=Ns, smp_diffuse=dr, smp_diffuse_s1=ds1)
    # finalize + post-processing on a gradient
    g = np.zeros((90, 160, 4), np.float32)
    yy, xx = np.mgrid[0:90, 0:160]
    g[..., 0] = xx / 100.0
    g[..., 1] = yy / 60.0
    g[..., 2] = (xx + yy) / 150.0
    out["fin_acc"] = g
    out["fin_1"] = r.finalize(g, 0).reshape(90, 160)
    out["fin_7"] = r.finalize(g * 7, 7).reshape(90, 160)
    px = out["fin_1"]
    out["fx_gamma"] = r.postfx(px, 0, 2.2)
    out["fx_chroma"] = r.postfx(px, 1, 5.0)
    out["fx_vignette"] = r.postfx(px, 2, 0.8)
    out["fx_all"] = r.postfx(r.postfx(r.postfx(px, 0, 2.2), 1, 5.0), 2, 0.8)
    return out


def bvh_vectors(r):
    """BuildBVH dumps (bvh.h:251-441) and traversal results over caller meshes"""
    from raytracer_amd import api
    rng = np.random.default_rng(7)
    out = {}
    for name in ("cow", "ball"):
        s = api.HostScene()
        s.load_obj(os.path.join(ROOT, "assets", name + ".obj"), api.Material())
        s.finalize()
        v9 = s.desc.arrays()["tri_verts"][:, :, :3].reshape(-1, 9).copy()
        h, nodes, prim, nq = r.bvh_create(v9)
        ctr = v9.reshape(-1, 3).mean(0)
        ext = np.abs(v9.reshape(-1, 3) - ctr).max()
        n = 20000
        O = (ctr + rand_dirs(rng, n) * ext * rng.uniform(0.2, 3, (n, 1))).astype(np.float32)
        D = (ctr + rng.normal(scale=ext * 0.4, size=(n, 3)) - O)
        D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
        tin = np.full(n, 1e34, np.float32)
        i0, t0 = r.bvh_trace(h, 0, O, D, tin)
        i1, t1 = r.bvh_trace(h, 1, O, D, tin)
        i2, t2 = r.bvh_trace(h, 2, O, D, tin)
        tsh = rng.uniform(0.05, 2.5 * ext, n).astype(np.float32)
        occ = r.bvh_occluded(h, 0, O, D, tsh)
        occq = r.bvh_occluded(h, 2, O, D, tsh)
        out.update({name + "_v9": v9, name + "_nodes": nodes, name + "_prim_idx": prim, name + "_nq": np.int32(nq),
                    name + "_O": O, name + "_D": D, name + "_idx_rec": i0, name + "_t_rec": t0, name + "_idx_stack": i1, name + "_t_stack": t1,
                    name + "_idx_qbvh": i2, name + "_t_qbvh": t2, name + "_tshadow": tsh, name + "_occ": occ, name + "_occ_qbvh": occq})
        r.bvh_destroy(h)
    # mixed spheres + random triangles, incl. duplicate centroids and coincident triangles (t-ties)
    nt = 3000
    base = rng.uniform(-1, 1, (nt, 1, 3))
    v9 = (base + rng.normal(scale=0.06, size=(nt, 3, 3))).astype(np.float32)
    v9[100:200] = v9[0:100]          # coincident copies -> exact t ties
    v9[200:260] = np.round(v9[200:260] * 8) / 8
    v9 = v9.reshape(nt, 9)
    pr = np.concatenate([rng.uniform(-1, 1, (40, 3)), rng.uniform(0.03, 0.2, (40, 1))], 1).astype(np.float32)
    h, nodes, prim, nq = r.bvh_create(v9, pr)
    n = 30000
    O = (rand_dirs(rng, n) * rng.uniform(0.0, 2.5, (n, 1))).astype(np.float32)
    D = rand_dirs(rng, n)
    D[:1500, rng.integers(0, 3, 1500)[0]] = 0  # a block with one zero component
    D[:1500] /= np.maximum(np.linalg.norm(D[:1500], axis=1, keepdims=True), 1e-20)
    D = D.astype(np.float32)
    tin = np.full(n, 1e34, np.float32)
    i0, t0 = r.bvh_trace(h, 0, O, D, tin)
    i2, t2 = r.bvh_trace(h, 2, O, D, tin)
    tsh = rng.uniform(0.05, 3, n).astype(np.float32)
    occ = r.bvh_occluded(h, 0, O, D, tsh)
    out.update(mix_v9=v9, mix_pr=pr, mix_nodes=nodes, mix_prim_idx=prim, mix_O=O, mix_D=D, mix_idx_rec=i0, mix_t_rec=t0, mix_idx_qbvh=i2, mix_t_qbvh=t2,
               mix_tshadow=tsh, mix_occ=occ)
    r.bvh_destroy(h)
    return out


def scene_vectors(r, full):
    """default scene: marshalled arrays, primary hits, incoherent queries, hit attributes, images"""
    rng = np.random.default_rng(99)
    out = {"camera": r.camera(), "quad_T": r.quad_T()}
    W, H = r.W, r.H
    ob, tb = r.primary_hits(False)
    oq, tq = r.primary_hits(True)
    out.update(hit_obj_bvh=ob.astype(np.int16), hit_t_bvh=tb)
    d = np.nonzero((ob != oq) | (tb != tq))[0].astype(np.int32)
    out.update(qbvh_diff_idx=d, qbvh_diff_obj=oq[d].astype(np.int16), qbvh_diff_t=tq[d])
    if not full:
        return out
    for k, v in r.scene_arrays().items():
        out["scene_" + k] = v
    O, D, rD = r.primary_rays()
    sel = rng.choice(W * H, 4000, replace=False)
    out.update(pr_sel=sel.astype(np.int32), pr_O=O[sel], pr_D=D[sel], pr_rD=rD[sel])
    # incoherent rays: closest hit + occlusion, binary and QBVH
    n = 60000
    O, D = room_rays(rng, n)
    tmax = np.full(n, 1e34, np.float32)
    fo, ft = r.find_nearest(O, D, tmax, False)
    fq, ftq = r.find_nearest(O, D, tmax, True)
    tsh = rng.uniform(0.05, 5, n).astype(np.float32)
    oc = r.is_occluded(O, D, tsh, False)
    ocq = r.is_occluded(O, D, tsh, True)
    out.update(inc_O=O, inc_D=D, inc_obj=fo.astype(np.int16), inc_t=ft, inc_obj_qbvh=fq.astype(np.int16), inc_t_qbvh=ftq, inc_tshadow=tsh, inc_occ=oc.astype(np.int8),
               inc_occ_qbvh=ocq.astype(np.int8))
    # hit attributes at those hits
    live = fo >= 0
    I = (O + ft[:, None] * D).astype(np.float32)
    N, alb, mat = r.hit_attribs(fo[live], I[live], D[live])
    out.update(att_obj=fo[live].astype(np.int16), att_I=I[live], att_wo=D[live], att_N=N, att_albedo=alb, att_mat=mat.astype(np.int8))
    # images (binary BVH = the parity target; QBVH = the reference default, differs only on zero-component rays)
    seeds = r.seeds()
    acc = r.sample(seeds, 0, 1)
    out["pt_1spp"] = acc.copy()
    acc = r.sample(seeds, 1, 3, acc)
    out["pt_4spp"] = acc.copy()
    acc = r.sample(seeds, 4, 12, acc)
    out["pt_16spp"] = acc.copy()
    out["pt_16spp_qbvh"] = r.sample(seeds, 0, 16, None, True)
    # sample-split view: samples 1,3,5,.. only (what rank 1 of 2 renders)
    out["pt_odd8"] = r.sample_strided(seeds, 1, 8, 2)
    out["whitted"] = r.trace(3, 5)
    out["whitted_depth2"] = r.trace(3, 2)
    out["whitted_aa"] = r.trace(3, 5, aa=True)
    out["view_normal"] = r.trace(0)
    out["view_distance"] = r.trace(1)
    out["view_albedo"] = r.trace(2)
    out["pt_16spp_rgb8"] = r.finalize(out["pt_16spp"], 16).reshape(H, W)
    out["whitted_rgb8"] = r.finalize(out["whitted"], 0).reshape(H, W)
    px = out["pt_16spp_rgb8"]
    out["pt_16spp_fx"] = r.postfx(r.postfx(r.postfx(px, 0, 2.2), 1, 5.0), 2, 0.8)
    return out


def worker(w, h):
    from oracle.ref import Ref
    r = Ref(w, h)
    os.makedirs(GOLD, exist_ok=True)
    if (w, h) == (320, 180):
        np.savez_compressed(os.path.join(GOLD, "unit_vectors.npz"), **unit_vectors(r))
        np.savez_compressed(os.path.join(GOLD, "bvh_vectors.npz"), **bvh_vectors(r))
        np.savez_compressed(os.path.join(GOLD, "default_320x180.npz"), **scene_vectors(r, True))
    else:
        np.savez_compressed(os.path.join(GOLD, "default_%dx%d_hits.npz" % (w, h)), **scene_vectors(r, False))
    print("wrote fixtures for %dx%d" % (w, h))


if __name__ == "__main__":
    if len(sys.argv) == 3:
        worker(int(sys.argv[1]), int(sys.argv[2]))
    else:
        from raytracer_amd import build
        build.build_host()
        build.build_ref()
        for w, h in ((320, 180), (1280, 720)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(w), str(h)])
        for f in sorted(os.listdir(GOLD)):
            print("%8d KiB  %s" % (os.path.getsize(os.path.join(GOLD, f)) // 1024, f))
