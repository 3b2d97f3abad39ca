"""C-ABI shape and host logic (no GPU): librtmi.so loads and exports every symbol include/rtmi.h declares,
struct layouts agree between the header, ctypes and the C++ host mirror, and error paths that need no device
return status codes instead of crashing."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from raytracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtmi.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = api.device_lib()
    names = declared_functions()
    assert len(names) >= 20 and "rtmi_render" in names and "rtmi_upload_scene" in names
    for n in names:
        assert hasattr(L, n), "librtmi.so does not export " + n
    assert b"gfx950" in L.rtmi_version()


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof as a C compiler sees include/rtmi.h == the ctypes mirrors in raytracer_amd/api.py"""
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rtmi.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(rtmi_bvh_node),sizeof(rtmi_scene_desc),sizeof(rtmi_postfx),sizeof(rtmi_stats),offsetof(rtmi_scene_desc,nodes),'
                    'offsetof(rtmi_scene_desc,light_corners),offsetof(rtmi_scene_desc,whitted_intensity),offsetof(rtmi_stats,ms_extend));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(api.BvhNode), C.sizeof(api.SceneDesc), C.sizeof(api.PostFx), C.sizeof(api.Stats), api.SceneDesc.nodes.offset,
            api.SceneDesc.light_corners.offset, api.SceneDesc.whitted_intensity.offset, api.Stats.ms_extend.offset]
    assert got == want
    assert C.sizeof(api.BvhNode) == 32  # GPUBVHNode, bvh.h:11-16


def test_null_arguments_are_rejected_without_a_device():
    L = api.device_lib()
    assert L.rtmi_create(None, 0, 16, 16) == -1
    h = C.c_void_p()
    assert L.rtmi_create(C.byref(h), 0, 0, 16) == -1 and not h.value
    for f in (L.rtmi_destroy, L.rtmi_synchronize, L.rtmi_reset_accumulator, L.rtmi_reset_stats):
        assert f(None) == -1
    assert L.rtmi_render(None, 4, 0, 1, 1, 5, 0) == -1
    assert L.rtmi_last_error(None) == b"null context"


def test_default_scene_conventions(default_scene):
    """node array conventions of bvh.h:56, 262-274, 380-381 and the object index order of SURVEY A.1"""
    info = default_scene.info()
    assert (info["quads"], info["spheres"], info["cubes"], info["planes"], info["triangles"]) == (1, 2, 1, 6, 2901)
    a = default_scene.desc.arrays()
    nodes = a["nodes"].view(np.int32)
    n = info["nodes_used"]
    left, count = nodes[:n, 6], nodes[:n, 7]
    interior = np.nonzero(count[2:] == 0)[0] + 2
    assert np.all(left[interior] >= 3) and np.all((left[interior] - 3) % 2 == 0)  # children in adjacent pairs from node 3
    children = np.concatenate([left[interior], left[interior] + 1])
    assert len(np.unique(children)) == len(children) == n - 3  # every node but 0,1,2 is the child of exactly one node
    leaves = np.nonzero(count[2:] > 0)[0] + 2
    covered = np.concatenate([np.arange(l, l + c) for l, c in zip(left[leaves], count[leaves])])
    assert np.array_equal(np.sort(covered), np.arange(info["spheres"] + info["triangles"]))
    assert np.array_equal(np.sort(a["prim_idx"]), np.arange(info["spheres"] + info["triangles"]))
    assert list(a["material_type"][:10]) == [api.LIGHT, api.GLASS, api.DIFFUSE, api.DIFFUSE, api.MIRROR] + [api.DIFFUSE] * 5


@pytest.mark.parametrize("builder", [api.REFERENCE_SAH, api.BINNED_SAH])
def test_builders_bound_their_primitives(builder):
    """every leaf box contains its triangles, every interior box its children, depth within the device stack bound"""
    s = api.HostScene().make_room()
    s.add_torus_knot(24, 24, (0, 0, 1.5), api.Material())
    s.finalize(builder)
    a, info = s.desc.arrays(), s.info()
    nodes = a["nodes"][:info["nodes_used"]]
    f, i = nodes.view(np.float32), nodes.view(np.int32)
    v = a["tri_verts"][:, :, :3]
    assert info["max_depth"] <= 64
    for k in range(2, info["nodes_used"]):
        lo, hi, left, cnt = f[k, 0:3], f[k, 3:6], i[k, 6], i[k, 7]
        if cnt > 0:
            tri = v[a["prim_idx"][left:left + cnt]]
            assert np.all(tri >= lo - 0) and np.all(tri <= hi + 0), k
        else:
            for c in (left, left + 1):
                assert np.all(f[c, 0:3] >= lo) and np.all(f[c, 3:6] <= hi), k


def test_obj_round_trip(tmp_path):
    """the OBJ path (Scene::LoadObject, scene.h:222-291) keeps file order and bytes: write a knot, load it, compare"""
    L = api.host_lib()
    n = 12 * 12 * 2
    v9 = np.zeros((n, 9), np.float32)
    assert L.rth_torus_knot(12, 12, 0.42, 0.17, 0.075, 2, 3, 0.0, 0.0, 1.5, v9.ctypes.data) == n
    path = str(tmp_path / "knot.obj")
    assert L.rth_write_obj(path.encode(), v9.ctypes.data, n) == 0
    s = api.HostScene()
    s.load_obj(path, api.Material())
    s.finalize()
    got = s.desc.arrays()["tri_verts"][:, :, :3].reshape(n, 9)
    assert np.array_equal(got.view(np.uint32), v9.view(np.uint32))


def test_binned_builder_is_independent_of_thread_count():
    """the top levels of a large tree are built in parallel chunks (host/bvh.cpp): min / max / counts are merged, so the
    node array must be the same bit for bit with 1 thread and with many (only the order inside a leaf range may differ)"""
    import subprocess
    import sys
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from raytracer_amd import api\n"
        "s = api.mesh_scene(200000, api.BINNED_SAH)\n"
        "a = s.desc.arrays()\n"
        "nodes = np.ascontiguousarray(a['nodes']); pi = np.ascontiguousarray(a['prim_idx'])\n"
        "nv = nodes.view(np.int32).reshape(-1, 8); leaves = nv[3:][nv[3:, 7] > 0]\n"
        "sets = np.add.reduceat(pi.astype(np.int64), np.sort(leaves[:, 6]))\n"
        "print(hashlib.sha1(nodes.tobytes()).hexdigest(), hashlib.sha1(sets.tobytes()).hexdigest(), s.info()['nodes_used'])\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for threads in ("1", "5"):
        env = dict(os.environ, OMP_NUM_THREADS=threads)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] and int(outs[0].split()[2]) > 200000


def test_camera_mirror_matches_oracle_restatement():
    """rtmi::Camera (host/rt_types.h) and Renderer::KeyDown/KeyUp against the oracle's restatement of camera.h:52-159 and
    renderer.cpp:368-441 over a fly-through script: translation keys held and released, rotations, SetFov, SetAspectRatio.
    (Both are restatements by reading; no reference-derived vector covers the camera: parity unpinned for this test.)"""
    from oracle.oracle import Oracle
    L = api.host_lib()
    W, H = 1280, 720
    rng = np.random.default_rng(3)
    keys = [api.KEY_A, api.KEY_D, api.KEY_E, api.KEY_Q, api.KEY_S, api.KEY_W, api.KEY_RIGHT, api.KEY_LEFT, api.KEY_DOWN, api.KEY_UP]
    events, held = [("update",)], set()
    for _ in range(400):
        k = keys[rng.integers(0, len(keys))]
        u = rng.random()
        if u < 0.25 and k not in held:
            events.append(("down", k)), held.add(k)
        elif u < 0.45 and k in held:
            events.append(("up", k)), held.discard(k)
        elif u < 0.5:
            events.append(("fov", int(rng.integers(45, 136))))
        elif u < 0.53:
            events.append(("aspect", float(rng.choice([4 / 3, 16 / 9, 1.0, 21 / 9]))))
        else:
            events.append(("update",))
    cams, moved = Oracle().camera_script(W, H, events)
    kinds = {"down": 0, "up": 1, "update": 2, "fov": 3, "aspect": 4}
    ev = np.array([[kinds[e[0]], e[1] if e[0] in ("down", "up") else 0] for e in events], np.int32)
    val = np.array([e[1] if e[0] in ("fov", "aspect") else 0 for e in events], np.float32)
    got, gm = np.zeros_like(cams), np.zeros_like(moved)
    assert L.rth_camera_script(W, H, len(events), ev.ctypes.data, val.ctypes.data, got.ctypes.data, gm.ctypes.data) == len(cams)
    assert np.array_equal(got.view(np.uint32), cams.view(np.uint32)) and np.array_equal(gm, moved)
    assert moved.any() and not moved.all() and np.isfinite(cams).all() and len(cams) > 100


def test_obj_with_materials_quads_and_index_forms(tmp_path):
    """`usemtl` / `mtllib` -> per-triangle Materials (the reference parses the .mtl and ignores it, scene.h:266), a quad
    split along its shorter diagonal as tinyobj does (tiny_obj_loader.h:1485-1580), v/vt/vn and negative indices, CRLF line ends"""
    (tmp_path / "m.mtl").write_text("newmtl red\nKd 0.9 0.1 0.2\nillum 2\n\nnewmtl mirror\nKd 0.8 0.8 0.8\nillum 3\n"
                                    "newmtl glass\nKd 1 1 1\nNi 1.52\nillum 7\nTf 0.5 0.25 1.0\nnewmtl lamp\nKe 2 2 2\nKd 1 1 1\n")
    obj = ("mtllib m.mtl\r\n"
           "v 0 0 0\r\nv 1 0 0\r\nv 1 1 0\r\nv 0 2 0\r\nv 0 0 1\r\nv 1 0 1\r\nv 0 1 1\r\n"
           "vn 0 0 1\nvt 0 0\n"
           "f 1 2 3\n"                    # before any usemtl: the fallback
           "usemtl red\nf 1/1/1 2/1/1 3/1/1 4/1/1\n"   # quad: diagonals |p3-p1|^2 = 2, |p4-p2|^2 = 5 -> [0,1,2],[0,2,3]
           "usemtl mirror\nf -3 -2 -1\n"
           "usemtl glass\nf 5//1 6//1 7//1\n"
           "usemtl unknown\nf 1 3 5\n"
           "usemtl lamp\nf 2 4 6 7 5\n")   # pentagon: fanned into 3
    (tmp_path / "t.obj").write_bytes(obj.encode())
    fb = api.Material(api.DIFFUSE, (0.5, 0.5, 0.5))
    s = api.HostScene()
    s.load_obj_with_materials(str(tmp_path / "t.obj"), fb, (0.25, 0, 0))
    s.finalize()
    a = s.desc.arrays()
    assert s.info()["triangles"] == 9
    types = a["material_type"].tolist()
    assert types == [api.DIFFUSE, api.DIFFUSE, api.DIFFUSE, api.MIRROR, api.GLASS, api.DIFFUSE, api.LIGHT, api.LIGHT, api.LIGHT]
    assert np.allclose(a["albedo"][0, :3], 0.5) and np.allclose(a["albedo"][1, :3], (0.9, 0.1, 0.2)) and np.allclose(a["albedo"][5, :3], 0.5)
    assert a["ior"][4] == np.float32(1.52) and np.allclose(a["absorption"][4, :3], -np.log(np.float32([0.5, 0.25, 1.0])), rtol=1e-6)
    v = a["tri_verts"][:, :, :3] - np.float32([0.25, 0, 0])
    P = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 2, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1]])
    want = [(0, 1, 2), (0, 1, 2), (0, 2, 3), (4, 5, 6), (4, 5, 6), (0, 2, 4), (1, 3, 5), (1, 5, 6), (1, 6, 4)]
    for k, tri in enumerate(want):
        assert np.array_equal(v[k], P[list(tri)]), k
    # the reference's own entry point: one Material per file whatever the file says
    s2 = api.HostScene()
    s2.load_obj(str(tmp_path / "t.obj"), fb)
    s2.finalize()
    assert set(s2.desc.arrays()["material_type"].tolist()) == {api.DIFFUSE} and s2.info()["triangles"] == 9
    # a face that names a vertex that does not exist is refused, nothing is added
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    s3 = api.HostScene()
    with pytest.raises(api.RtmiError):
        s3.load_obj(str(tmp_path / "bad.obj"), fb)


def test_obj_ingest_at_scale(tmp_path):
    """the 1M-triangle scene of configs[2] through the OBJ path: written as text (rth_write_obj), read back by the chunk-parallel
    reader (Scene::LoadObject): every vertex bit-identical, triangles in file order, with one thread as with many"""
    import subprocess
    import sys
    import time
    L = api.host_lib()
    nu, nv = 5400, 31  # one knot of the 1M scene: 334 800 triangles; three of them in one file
    n1 = nu * nv * 2
    v9 = np.zeros((3 * n1, 9), np.float32)
    for k, x in enumerate((-10.5, 0.0, 10.5)):
        assert L.rth_torus_knot(nu, nv, 0.42 * 7, 0.17 * 7, 0.075 * 7, 2, 3, x, 0.0, 10.5, v9[k * n1:].ctypes.data) == n1
    path = str(tmp_path / "knots1m.obj")
    assert L.rth_write_obj(path.encode(), v9.ctypes.data, len(v9)) == 0
    assert os.path.getsize(path) > 50e6
    t0 = time.time()
    s = api.HostScene()
    s.load_obj(path, api.Material())
    dt = time.time() - t0
    s.finalize(api.BINNED_SAH)
    got = s.desc.arrays()["tri_verts"][:, :, :3].reshape(-1, 9)
    assert len(got) == 3 * n1 > 1000000 and np.array_equal(got.view(np.uint32), v9.view(np.uint32))
    print("OBJ ingest: %d triangles, %.0f MB of text in %.2f s" % (len(got), os.path.getsize(path) / 1e6, dt))
    # one thread reads the same bytes into the same triangles
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\nfrom raytracer_amd import api\n"
            "s = api.HostScene(); s.load_obj(%r, api.Material()); s.finalize(api.BINNED_SAH)\n"
            "print(hashlib.sha1(np.ascontiguousarray(s.desc.arrays()['tri_verts']).tobytes()).hexdigest())\n") % (ROOT, path)
    one = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, check=True).stdout.strip()
    import hashlib
    assert one == hashlib.sha1(np.ascontiguousarray(s.desc.arrays()["tri_verts"]).tobytes()).hexdigest()


def test_host_build_qbvh_matches_the_reference_counts(default_scene):
    """BVH::BuildQBVH of the host mirror (bvh.h:213-249 via host/qbvh.h): 812 nodes for the default scene (measured on the compiled
    reference), as many as the oracle's independent restatement; every leaf position owned by exactly one slot."""
    from oracle.oracle import Oracle
    q = default_scene.build_qbvh()
    o = Oracle(default_scene.desc)
    assert o.set_qbvh(True)
    assert len(q) == 812 == o.qbvh_nodes()
    child, count = q[:, 24:28].view(np.int32), q[:, 28:32].view(np.int32)
    nb = default_scene.info()["spheres"] + default_scene.info()["triangles"]
    owned = np.zeros(nb, np.int32)
    for n in range(1, len(q)):  # (node 0 holds the plane leaf and the primitive root)
        for s in range(4):
            if child[n, s] == -1:
                assert (child[n, s:] == -1).all()  # the traversal leaves a node at its first empty slot (bvh.h:611)
                break
            if count[n, s] > 0:
                owned[child[n, s]:child[n, s] + count[n, s]] += 1
    assert (owned == 1).all()
    assert child[0, 1] == 1 and count[0, 0] == default_scene.info()["planes"]


def test_integration_sections_compile_verbatim_against_compat(tmp_path):
    """INTEGRATION.md 1-3 are code, not pseudo-code: the three blocks stand character for character in tests/compat/integration_tu.cpp,
    which compiles against compat/ (the reference's header names in namespace Tmpl8) with the line a maintainer would use, and
    the library it yields exports what tests/test_gpu_compat.py calls (no compute here: no GPU)"""
    import re
    import subprocess
    from raytracer_amd import build
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    tu = open(build.COMPAT_TU).read()
    heads = [m.start() for m in re.finditer(r"^## [123]\. ", text, re.M)]
    assert len(heads) == 3
    for k, at in enumerate(heads):
        block = re.search(r"```cpp\n(.*?)```", text[at:], re.S).group(1)
        assert len(block.splitlines()) >= 8 and block in tu, "INTEGRATION.md section %d is not in the TU verbatim" % (k + 1)
    assert "scene.FindNearest( ray )" in tu
    for h in ("camera.h", "primitive.h", "material.h", "bvh.h", "scene.h", "renderer.h", "precomp.h"):
        assert "namespace Tmpl8" in open(os.path.join(ROOT, "compat", h)).read(), h
    out = str(tmp_path / "libcompat_tu.so")
    subprocess.check_call(build.compat_tu_command(out, extra=["-Werror"]))
    L = C.CDLL(out)
    for name in ("tu_init", "tu_tick", "tu_seeds", "tu_set_light", "tu_scene_queries", "tu_brute_force"):
        assert hasattr(L, name), name
