"""Build helpers: compile the HIP device library, the C++ host mirror and (test infrastructure) the oracle.

Everything is built in-tree with explicit compiler lines (hipcc cross-compiles gfx950 without a GPU):

  raytracer_amd/librtmi.so        include/rtmi.h: csrc/rt_kernels.hip (hipcc) + csrc/rtmi_host.cpp (g++), -ffp-contract=off
  raytracer_amd/librtmi_host.so   Scene/BVH/Camera/Renderer host mirror    (g++,   -ffp-contract=off, OpenMP)
  tests/compat/libcompat_tu.so    INTEGRATION.md 1-3 compiled against compat/ (the source-level drop-in headers), test only
  oracle/liboracle.so             CPU restatement, the parity checker      (g++,   -ffp-contract=off, OpenMP)
  oracle/_ref/libref_WxH.so       the reference itself, dev container only (oracle/ref_build/build_ref.sh)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracer_amd")
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
ORACLE = os.path.join(ROOT, "oracle")

DEVICE_LIB = os.path.join(PKG, "librtmi.so")
HOST_LIB = os.path.join(PKG, "librtmi_host.so")
ORACLE_LIB = os.path.join(ORACLE, "liboracle.so")
COMPAT = os.path.join(ROOT, "compat")
COMPAT_TU = os.path.join(ROOT, "tests", "compat", "integration_tu.cpp")
COMPAT_TU_LIB = os.path.join(ROOT, "tests", "compat", "libcompat_tu.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_device(force=False):
    """librtmi.so = rt_kernels.hip (kernels + launchers, hipcc for gfx950) + rtmi_host.cpp (the C ABI and all host-side
    orchestration, plain C++ compiled by g++ against the HIP runtime API headers)."""
    kern = [os.path.join(CSRC, f) for f in ("rt_kernels.hip", "rt_device.h", "rt_shared.h")]
    host = [os.path.join(CSRC, f) for f in ("rtmi_host.cpp", "rt_shared.h")] + [os.path.join(ROOT, "include", "rtmi.h")]
    host_obj = os.path.join(CSRC, "rtmi_host.o")
    if not force and not _newer(DEVICE_LIB, kern + host):
        return DEVICE_LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if force or _newer(host_obj, host):
        _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__",
              "-I" + os.path.join(rocm, "include"), "-c", "-o", host_obj, host[0]])
    kern_obj = os.path.join(CSRC, "rt_kernels.o")
    if force or _newer(kern_obj, kern):
        _run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result", "-c", "-o", kern_obj, kern[0]])
    _run([_hipcc(), "-fPIC", "-shared", "-o", DEVICE_LIB, kern_obj, host_obj, "-ldl", "-lpthread"])
    return DEVICE_LIB


def build_host(force=False):
    srcs = [os.path.join(HOST, f) for f in ("bvh.cpp", "scene.cpp", "renderer.cpp", "capi.cpp")]
    deps = srcs + [os.path.join(HOST, f) for f in ("bvh.h", "scene.h", "renderer.h", "rt_types.h", "qbvh.h", "frames.h")] + [os.path.join(ROOT, "include", "rtmi.h")]
    if not force and not _newer(HOST_LIB, deps):
        return HOST_LIB
    build_device(force)
    _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared", "-Wall",
          "-o", HOST_LIB] + srcs + ["-L" + PKG, "-lrtmi", "-Wl,-rpath,$ORIGIN"])
    return HOST_LIB


def compat_tu_command(out=COMPAT_TU_LIB, extra=()):
    """the compiler line a maintainer of the reference would use for renderer.cpp after the switch: the reference's include names
    resolved by compat/, the C ABI by include/, linked against the two libraries"""
    return ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall", "-I" + COMPAT, "-I" + os.path.join(ROOT, "include"),
            "-o", out, COMPAT_TU] + list(extra) + ["-L" + PKG, "-lrtmi_host", "-lrtmi", "-Wl,-rpath," + PKG, "-Wl,-rpath,$ORIGIN/../../raytracer_amd"]


def build_compat(force=False):
    deps = [COMPAT_TU] + [os.path.join(COMPAT, f) for f in sorted(os.listdir(COMPAT))] + [os.path.join(HOST, f) for f in ("bvh.h", "scene.h", "rt_types.h", "qbvh.h")]
    deps.append(os.path.join(ROOT, "include", "rtmi.h"))
    if not force and not _newer(COMPAT_TU_LIB, deps):
        return COMPAT_TU_LIB
    build_host(force)
    _run(compat_tu_command())
    return COMPAT_TU_LIB


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE, "rt_oracle.cpp")]
    deps = srcs + [os.path.join(ROOT, "include", "rtmi.h")]
    if not force and not _newer(ORACLE_LIB, deps):
        return ORACLE_LIB
    _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared", "-Wall",
          "-o", ORACLE_LIB] + srcs)
    return ORACLE_LIB


def build_ref(resolutions=((1280, 720), (320, 180))):
    """Dev container only: compiles the reference where it lies.  No-op when /root/reference is absent."""
    if not os.path.isdir(os.environ.get("RT_REF_ROOT", "/root/reference/Raytracer")):
        return []
    out = []
    for w, h in resolutions:
        lib = os.path.join(ORACLE, "_ref", "libref_%dx%d.so" % (w, h))
        script = os.path.join(ORACLE, "ref_build", "build_ref.sh")
        deps = [script, os.path.join(ORACLE, "ref_build", "ref_driver.cpp"), os.path.join(ORACLE, "ref_build", "shim", "precomp.h")]
        if _newer(lib, deps):
            _run(["bash", script, str(w), str(h)])
        out.append(lib)
    return out


def build_all(force=False):
    build_device(force)
    build_host(force)
    build_oracle(force)
    build_compat(force)
    build_ref()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
