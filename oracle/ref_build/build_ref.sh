#!/bin/bash
# oracle/ref_build/build_ref.sh W H  -- TEST INFRASTRUCTURE, dev-container only.
#
# Compiles the REFERENCE's own hot-path sources, where they lie under $RT_REF_ROOT
# (default /root/reference/Raytracer), into oracle/_ref/libref_<W>x<H>.so following the recipe of
# SURVEY.md section 8(c).  Everything cut or patched out of the reference lives only in a
# temporary directory that is deleted afterwards; the only output is the .so.
#
#   text cut verbatim with sed -n (no edits):
#     template/precomp.h:170-280,344-356,431-473,487-1227   types, Timer, pixel ops, RNG decls, math
#     template/template.cpp:871-878,883-900,954-1035,1761-1803   WangHash, xorshift32, math impl, Surface loader
#   patched copies (one token each, SURVEY 8c steps 5,7,8):
#     camera.h     SCRWIDTH/SCRHEIGHT -> W/H (the reference #defines them unconditionally, camera.h:4-5)
#     renderer.h, renderer.cpp   Trace(Ray&..)/Sample(Ray&) -> by value (MSVC-only temporary->Ray& binding)
#     template/scene.h:348       `float3 lights[]` -> `static float3 lights[]` (dangling stack pointer)
set -euo pipefail
W=${1:-1280}; H=${2:-720}
R=${RT_REF_ROOT:-/root/reference/Raytracer}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
[ -d "$R" ] || { echo "reference tree not present at $R: skipping" >&2; exit 0; }
mkdir -p "$OUT"
B=$(mktemp -d /tmp/rt_ref_build.XXXXXX)
trap 'rm -rf "$B"' EXIT
P=$R/template/precomp.h; T=$R/template/template.cpp
sed -n '170,280p'   "$P" > "$B/_pc_types.inc"
sed -n '344,356p'   "$P" > "$B/_pc_timer.inc"
sed -n '431,473p'   "$P" > "$B/_pc_pixel.inc"
sed -n '487,1227p'  "$P" > "$B/_pc_math.inc"
sed -n '871,878p'   "$T" > "$B/_tc_wang.inc"
sed -n '883,900p'   "$T" > "$B/_tc_rng.inc"
sed -n '954,1035p'  "$T" > "$B/_tc_math.inc"
sed -n '1761,1803p' "$T" > "$B/_tc_surface.inc"
sed -e "s/^#define SCRWIDTH.*/#define SCRWIDTH $W/" -e "s/^#define SCRHEIGHT.*/#define SCRHEIGHT $H/" "$R/camera.h" > "$B/camera.h"
sed -e 's/Trace( Ray& ray/Trace( Ray ray/' -e 's/Sample(Ray& ray)/Sample(Ray ray)/' "$R/renderer.h" > "$B/renderer.h"
sed -e 's/Trace( Ray& ray/Trace( Ray ray/' -e 's/Sample(Ray& ray)/Sample(Ray ray)/' "$R/renderer.cpp" > "$B/renderer.cpp"
sed -e 's/float3 lights\[\] = /static float3 lights[] = /' "$R/template/scene.h" > "$B/scene.h"
grep -q "Trace( Ray ray" "$B/renderer.cpp" && grep -q "Sample(Ray ray)" "$B/renderer.h" && grep -q "static float3 lights" "$B/scene.h"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-std=c++17 -O2 -fopenmp -fms-extensions -fdeclspec -mavx2 -ffp-contract=off -w -fPIC"
INC="-I$B -I$HERE/shim -I$R -I$R/template -I$R/tinyobjloader -I$R/lib"
$CXX $FLAGS $INC -c "$B/renderer.cpp" -o "$B/renderer.o"
$CXX $FLAGS $INC -c "$HERE/ref_driver.cpp" -o "$B/driver.o"
$CXX $FLAGS -I$R/tinyobjloader -c "$R/tinyobjloader/tiny_obj_loader.cc" -o "$B/tinyobj.o"
$CXX -shared -fopenmp -o "$OUT/libref_${W}x${H}.so" "$B/renderer.o" "$B/driver.o" "$B/tinyobj.o"
echo "built $OUT/libref_${W}x${H}.so"
