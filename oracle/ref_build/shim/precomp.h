// oracle/ref_build/shim/precomp.h -- TEST INFRASTRUCTURE, dev-container only.
//
// Include-path front door used ONLY by oracle/ref_build/build_ref.sh to compile the
// reference's own hot-path sources (primitive.h, bvh.h, material.h, camera.h,
// template/scene.h, renderer.h, renderer.cpp, postprocessing.h) where they lie under
// /root/reference/Raytracer, following the recipe recorded in SURVEY.md section 8(c).
//
// The reference's real template/precomp.h cannot be included on Linux (windows.h, io.h,
// WGL, GLFW and OpenCL import libraries, precomp.h:20,76-92).  This file therefore
//   * pulls the reference's OWN text for vector types, pixel ops, RNG declarations and
//     all math (precomp.h:170-280, 344-356, 431-473, 487-1227) from *.inc files that
//     build_ref.sh cuts out of the reference with `sed -n` into a temporary directory
//     at build time (nothing of the reference is stored in this repository), and
//   * declares the window/OpenCL *plumbing* classes (Surface, Buffer, Kernel, TheApp)
//     as inert shells.  None of them does arithmetic on the per-pixel trace path; the
//     only one whose data reaches a pixel is Surface("assets/logo.png"), whose loader
//     is again the reference's own text (template.cpp:1761-1803) over its vendored
//     lib/stb_image.h.
// Nothing here is product code and nothing here is shipped to or needed on the GPU box
// except the resulting binary oracle/_ref/*.so.
#pragma once

#include <chrono>
#include <fstream>
#include <iostream>
#include <vector>
#include <list>
#include <string>
#include <thread>
#include <tuple>
#include <math.h>
#include <cmath>
#include <algorithm>
#include <assert.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <immintrin.h>

using namespace std;

typedef unsigned char uchar;
typedef unsigned int uint;
typedef unsigned short ushort;

namespace Tmpl8
{
// precomp.h:109-134, reduced to the members the trace path touches
class Surface
{
public:
	Surface() = default;
	Surface( int w, int h, uint* buffer );
	Surface( int w, int h );
	Surface( const char* file );
	~Surface();
	void LoadImage( const char* file );
	uint* pixels = 0;
	int width = 0, height = 0;
	bool ownBuffer = false;
};
}
using namespace Tmpl8;

#include "_pc_types.inc"   // precomp.h:170-280  ALIGN/MALLOC64, int2..float4, float3, uchar4
#include "_pc_timer.inc"   // precomp.h:344-356  Timer
#include "_pc_pixel.inc"   // precomp.h:431-473  RGBF32_to_RGB8 x2, RNG declarations
void FatalError( const char* fmt, ... );
#include "_pc_math.inc"    // precomp.h:487-1227 fminf..mat4..TransformPosition decl

// ---- inert plumbing shells (precomp.h:1351-1553 are OpenCL wrappers) ----
#define CL_MEM_READ_ONLY 4
#define CL_MEM_WRITE_ONLY 2
class Buffer
{
public:
	Buffer() {}
	Buffer( unsigned int N, void* ptr = 0, unsigned int t = 0 ) : hostBuffer( (unsigned int*)ptr ), size( N ) {}
	unsigned int* GetHostPtr() { return hostBuffer; }
	void CopyToDevice( bool = true ) {}
	void CopyFromDevice( bool = true ) {}
	unsigned int* hostBuffer = 0;
	unsigned int size = 0;
};
class Kernel
{
public:
	Kernel( const char*, const char* ) {}
	template <typename... A> void SetArguments( A... ) {}
	template <typename T> void S( uint, T ) {}
	void Run( const size_t, const size_t = 0 ) {}
};

#include "common.h"        // reference template/common.h (PI etc.)

// GLFW key codes referenced by renderer.cpp:368-441 (interactive camera, out of scope)
enum { GLFW_KEY_A = 65, GLFW_KEY_D = 68, GLFW_KEY_E = 69, GLFW_KEY_Q = 81, GLFW_KEY_S = 83, GLFW_KEY_W = 87,
	GLFW_KEY_RIGHT = 262, GLFW_KEY_LEFT = 263, GLFW_KEY_DOWN = 264, GLFW_KEY_UP = 265 };

namespace Tmpl8
{
// precomp.h:1675-1688
class TheApp
{
public:
	virtual void Init() = 0;
	virtual void Tick( float deltaTime ) = 0;
	virtual void Shutdown() = 0;
	virtual void MouseUp( int button ) = 0;
	virtual void MouseDown( int button ) = 0;
	virtual void MouseMove( int x, int y ) = 0;
	virtual void MouseWheel( float y ) = 0;
	virtual void KeyUp( int key ) = 0;
	virtual void KeyDown( int key ) = 0;
	Surface* screen = 0;
};
}

#include "tiny_obj_loader.h"

// bvh.h:616,731 read MSVC's __m128::m128_i32.  Give the name a union view for the span of
// the reference headers only; arithmetic is untouched.
struct Ref128
{
	union { __v4sf v; float m128_f32[4]; int m128_i32[4]; };
	Ref128() = default;
	Ref128( __v4sf a ) : v( a ) {}
	operator __v4sf() const { return v; }
};
#define __m128 Ref128
#include "scene.h"         // -> material.h, bvh.h -> primitive.h   (scene.h via patched copy, see build_ref.sh)
#undef __m128
#include "camera.h"        // patched copy: SCRWIDTH/SCRHEIGHT per build
#include "renderer.h"      // patched copy: Trace/Sample take Ray by value
#include "material.h"
#include "postprocessing.h"
