// oracle/ref_build/ref_driver.cpp -- TEST INFRASTRUCTURE, dev-container only.
//
// C entry points around the REFERENCE's own compiled hot path (see shim/precomp.h and
// build_ref.sh).  Used by tests/ and tools/make_golden.py to (1) validate this repo's CPU
// restatement (oracle/rt_oracle.cpp) and (2) generate the golden vectors in tests/golden/.
// Never linked, imported or executed by the product path.
//
// What is reference code here: everything reached through Scene / BVH / Camera / Renderer /
// PostProcessing / primitive classes / the math in _pc_math.inc / the out-of-line math and RNG text
// cut from template.cpp by build_ref.sh (_tc_*.inc).  What is ours: the thread_local RNG state the
// reference RNG text reads (the reference's global RNG is racy under its own OpenMP
// loop), the per-(pixel,sample) seeding, and the marshalling below.
#include "precomp.h"
#include <stdarg.h>
#include <unistd.h>
#include <stdint.h>
#include <omp.h>

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h" // reference lib/stb_image.h, compiled where it lies

void FatalError( const char* fmt, ... )
{
	va_list a; va_start( a, fmt ); vfprintf( stderr, fmt, a ); va_end( a ); fputc( '\n', stderr ); exit( 3 );
}

// ---- reference RNG text (template.cpp:871-878 WangHash, 883-900 xorshift32) over OUR state variable
static thread_local uint seed = 0x12345678;
#include "_tc_wang.inc"
#include "_tc_rng.inc"
// ---- reference out-of-line math (template.cpp:954-1035)
#include "_tc_math.inc"
// ---- reference Surface ctor/LoadImage/dtor (template.cpp:1761-1803)
#include "_tc_surface.inc"

static Renderer* R = 0;
static Surface* g_logo = 0; // assets/logo.png through the reference loader (template.cpp:1775-1798)

static inline uint SampleSeed( uint s0, uint s )
{
	// SURVEY A.5 contract: state(pixel,sample) = WangHash(seeds0[p] + s*0x9E3779B9), 0 -> 1
	uint st = WangHash( s0 + s * 0x9E3779B9u );
	return st ? st : 1u;
}

extern "C" {

int ref_init( const char* root )
{
	if (R) return 0;
	char cwd[4096];
	if (!getcwd( cwd, sizeof( cwd ) )) return -1;
	if (chdir( root ) != 0) return -2;
	seed = 0x12345678;
	R = new Renderer();
	R->screen = new Surface( SCRWIDTH, SCRHEIGHT );
	// force the lazily-loaded logo texture while cwd is the reference root (primitive.h:178)
	R->scene.planes[5].GetAlbedo( float3( 0, 0, 3.99f ) );
	g_logo = new Surface( "assets/logo.png" );
	if (chdir( cwd ) != 0) return -3;
	return 0;
}
int ref_width() { return SCRWIDTH; }
int ref_height() { return SCRHEIGHT; }
void ref_sizes( int* out )
{
	out[0] = (int)sizeof( Ray ), out[1] = (int)sizeof( BVHNode ), out[2] = (int)sizeof( GPUBVHNode );
	out[3] = (int)sizeof( SIMD_BVH_Node ), out[4] = (int)sizeof( Triangle ), out[5] = (int)sizeof( mat4 );
}
void ref_counts( int* out )
{
	Scene& s = R->scene;
	out[0] = s.quads_size, out[1] = s.spheres_size, out[2] = s.cubes_size, out[3] = s.planes_size, out[4] = s.triangles_size;
	out[5] = (int)s.bvh->nodesUsed, out[6] = (int)s.bvh->nodesUsedQBVH;
}
// the marshalled arrays the reference uploads (scene.h:94-217), copied out as-is
void ref_scene_arrays( float* triInfos, float* sphereInfos, float* sphereInvrs, float* quadInvT, float* quadSizes,
	float* cubeM, float* cubeInvM, float* cubeBs, float* primitives, float* albedos, int* mats, float* iors,
	float* absorptions, float* lights, float* lightASE /* A, s, emission xyz */, void* gpuNodes, int* primIdx, uint* texture )
{
	Scene& s = R->scene;
	int total = s.quads_size + s.spheres_size + s.cubes_size + s.planes_size + s.triangles_size;
	memcpy( triInfos, s.triangleInfos, sizeof( float4 ) * 3 * s.triangles_size );
	memcpy( sphereInfos, s.sphereInfos, sizeof( float4 ) * s.spheres_size );
	memcpy( sphereInvrs, s.sphereInvrs, sizeof( float ) * s.spheres_size );
	memcpy( quadInvT, s.quadMatrices, 64 * s.quads_size );
	memcpy( quadSizes, s.quadSizes, 4 * s.quads_size );
	memcpy( cubeM, s.cubeMatrices, 64 * s.cubes_size );
	memcpy( cubeInvM, s.cubeInvMatrices, 64 * s.cubes_size );
	memcpy( cubeBs, s.cubeBs, 32 * s.cubes_size );
	memcpy( primitives, s.primitives, 16 * total );
	memcpy( albedos, s.albedos, 16 * total );
	memcpy( mats, s.primMaterials, 4 * total );
	memcpy( iors, s.refractiveIndices, 4 * total );
	memcpy( absorptions, s.absorptions, 16 * total );
	float4 l[3] = { float4( s.quads[0].c1, 0 ), float4( s.quads[0].c2, 0 ), float4( s.quads[0].c3, 0 ) };
	memcpy( lights, l, 48 );
	lightASE[0] = s.quads[0].A, lightASE[1] = s.quads[0].s;
	lightASE[2] = s.quads[0].material.emission.x, lightASE[3] = s.quads[0].material.emission.y, lightASE[4] = s.quads[0].material.emission.z;
	memcpy( gpuNodes, s.bvh->gpuBvhNode, sizeof( GPUBVHNode ) * ((s.spheres_size + s.triangles_size) * 2 + 1) );
	memcpy( primIdx, s.bvh->primitiveIdx, 4 * (s.spheres_size + s.triangles_size) );
	if (g_logo && g_logo->pixels) memcpy( texture, g_logo->pixels, 4 * 128 * 64 );
}
void ref_quad_T( float* T16 ) { memcpy( T16, R->scene.quads[0].T.cell, 64 ); }
void ref_camera( float* cam16 ) { memcpy( cam16, R->camera->gpuCamProp, 64 ); }

// first n outputs of the reference generator from its initial state (renderer.cpp:51-55)
void ref_seeds( uint* out, int n )
{
	seed = 0x12345678;
	for (int i = 0; i < n; i++) out[i] = RandomUInt();
}
void ref_rng_floats( uint s0, int n, float* out ) { seed = s0; for (int i = 0; i < n; i++) out[i] = RandomFloat(); }
uint ref_sample_seed( uint s0, uint s ) { return SampleSeed( s0, s ); }

void ref_primary_rays( float* O, float* D, float* rD )
{
	for (int y = 0; y < SCRHEIGHT; y++) for (int x = 0; x < SCRWIDTH; x++)
	{
		Ray r = R->camera->GetPrimaryRay( x, y );
		int p = x + y * SCRWIDTH;
		memcpy( O + 3 * p, &r.O, 12 ), memcpy( D + 3 * p, &r.D, 12 ), memcpy( rD + 3 * p, &r.rD, 12 );
	}
}
// Scene::FindNearest on primary rays; mode 0 = binary BVH (scalar slab test), 1 = QBVH (reference default)
void ref_primary_hits( int useQBVH, int* objIdx, float* t )
{
	R->scene.SetQBVH( useQBVH != 0 );
#pragma omp parallel for schedule( dynamic )
	for (int y = 0; y < SCRHEIGHT; y++) for (int x = 0; x < SCRWIDTH; x++)
	{
		Ray r = R->camera->GetPrimaryRay( x, y );
		R->scene.FindNearest( r );
		objIdx[x + y * SCRWIDTH] = r.objIdx, t[x + y * SCRWIDTH] = r.t;
	}
	R->scene.SetQBVH( true );
}
// generic ray batch against the default scene: FindNearest / IsOccluded
void ref_find_nearest( int useQBVH, int n, const float* O, const float* D, const float* tmax, int* objIdx, float* t )
{
	R->scene.SetQBVH( useQBVH != 0 );
#pragma omp parallel for schedule( dynamic, 256 )
	for (int i = 0; i < n; i++)
	{
		Ray r( float3( O[3 * i], O[3 * i + 1], O[3 * i + 2] ), float3( D[3 * i], D[3 * i + 1], D[3 * i + 2] ), tmax[i] );
		R->scene.FindNearest( r );
		objIdx[i] = r.objIdx, t[i] = r.t;
	}
	R->scene.SetQBVH( true );
}
void ref_is_occluded( int useQBVH, int n, const float* O, const float* D, const float* tmax, int* occ )
{
	R->scene.SetQBVH( useQBVH != 0 );
#pragma omp parallel for schedule( dynamic, 256 )
	for (int i = 0; i < n; i++)
	{
		Ray r( float3( O[3 * i], O[3 * i + 1], O[3 * i + 2] ), float3( D[3 * i], D[3 * i + 1], D[3 * i + 2] ), tmax[i] );
		occ[i] = R->scene.IsOccluded( r ) ? 1 : 0;
	}
	R->scene.SetQBVH( true );
}
// hit attributes as Renderer::Sample reads them (renderer.cpp:256-258, 274)
void ref_hit_attribs( int n, const int* objIdx, const float* I, const float* wo, float* N, float* albedo, int* mat )
{
	for (int i = 0; i < n; i++)
	{
		Material m = R->scene.GetMaterial( objIdx[i] );
		float3 Ii( I[3 * i], I[3 * i + 1], I[3 * i + 2] ), w( wo[3 * i], wo[3 * i + 1], wo[3 * i + 2] );
		float3 nn = R->scene.GetNormal( objIdx[i], Ii, w );
		float3 a = R->scene.GetAlbedo( objIdx[i], Ii, m );
		memcpy( N + 3 * i, &nn, 12 ), memcpy( albedo + 3 * i, &a, 12 ), mat[i] = (int)m.type;
	}
}

// Renderer::Sample accumulated exactly like Tick's CPU path (renderer.cpp:559-567), with the
// per-(pixel,sample) stream of SURVEY A.5.  acc is W*H*4 floats, caller-zeroed for s_begin==0.
void ref_sample_strided( int useQBVH, const uint* seeds0, int s_begin, int s_count, int stride, float* acc )
{
	R->scene.SetQBVH( useQBVH != 0 );
	R->visualizationMode = Renderer::PathTracing;
	for (int k = 0; k < s_count; k++)
	{
		const int s = s_begin + k * stride;
#pragma omp parallel for schedule( dynamic )
		for (int y = 0; y < SCRHEIGHT; y++) for (int x = 0; x < SCRWIDTH; x++)
		{
			int p = x + y * SCRWIDTH;
			seed = SampleSeed( seeds0[p], (uint)s );
			float4 c = float4( R->Sample( R->camera->GetPrimaryRay( x, y ) ), 0 );
			float4* a = (float4*)acc + p;
			*a += c;
		}
	}
	R->scene.SetQBVH( true );
}
void ref_sample( int useQBVH, const uint* seeds0, int s_begin, int s_count, float* acc ) { ref_sample_strided( useQBVH, seeds0, s_begin, s_count, 1, acc ); }
// Renderer::Trace for the non-path-tracing modes (renderer.cpp:578-592); vis = enum renderer.h:9
void ref_trace( int useQBVH, int vis, int depth, float* out )
{
	R->scene.SetQBVH( useQBVH != 0 );
	R->visualizationMode = (Renderer::VisualizationMode)vis;
	R->recursionDepth = depth;
#pragma omp parallel for schedule( dynamic )
	for (int y = 0; y < SCRHEIGHT; y++) for (int x = 0; x < SCRWIDTH; x++)
	{
		float4 c = float4( R->Trace( R->camera->GetPrimaryRay( x, y ), 0 ), 0 );
		memcpy( out + 4 * (x + y * SCRWIDTH), &c, 16 );
	}
	R->scene.SetQBVH( true );
	R->recursionDepth = 5;
}
void ref_trace_aa( int useQBVH, int depth, float* out ) // renderer.cpp:594-618
{
	R->scene.SetQBVH( useQBVH != 0 );
	R->visualizationMode = Renderer::RayTracing;
	R->recursionDepth = depth;
#pragma omp parallel for schedule( dynamic )
	for (int y = 0; y < SCRHEIGHT; y++) for (int x = 0; x < SCRWIDTH; x++)
	{
		float4 a = float4( 0 );
		for (int sample = 0; sample < 4; ++sample)
		{
			float sx = x + R->samplePattern[2 * sample], sy = y + R->samplePattern[2 * sample + 1];
			a += float4( R->Trace( R->camera->GetPrimaryRay( sx, sy ), 0 ), 0 );
		}
		a /= 4.0f;
		memcpy( out + 4 * (x + y * SCRWIDTH), &a, 16 );
	}
	R->scene.SetQBVH( true );
	R->recursionDepth = 5;
}

// pixel conversion and post-processing on caller data
void ref_finalize( int n, const float* acc, int f, uint* out )
{
	for (int i = 0; i < n; i++)
	{
		float4 v; memcpy( &v, acc + 4 * i, 16 );
		out[i] = f <= 0 ? RGBF32_to_RGB8( &v ) : RGBF32_to_RGB8( &v, f );
	}
}
void ref_postfx( uint* pixels, int w, int h, int which, float amount )
{
	PostProcessing pp( w, h, pixels );
	if (which == 0) pp.GammaCorrection( amount );
	if (which == 1) pp.ChromaticAberration( amount );
	if (which == 2) pp.Vignette( amount );
}

// ---- unit level: primitive intersection on caller-provided data -------------------------------
static Material dummyMat() { return Material( Material::MaterialType::DIFFUSE, float3( 0.5f ), 0 ); }
static Ray mkray( const float* O, const float* D, const float* t, int i )
{
	return Ray( float3( O[3 * i], O[3 * i + 1], O[3 * i + 2] ), float3( D[3 * i], D[3 * i + 1], D[3 * i + 2] ), t[i] );
}
void ref_tri_intersect( int n, const float* O, const float* D, const float* tin, const float* v9, int* hit, float* tout )
{
	for (int i = 0; i < n; i++)
	{
		const float* v = v9 + 9 * i;
		Triangle tri( 7, float3( v[0], v[1], v[2] ), float3( v[3], v[4], v[5] ), float3( v[6], v[7], v[8] ), dummyMat() );
		Ray r = mkray( O, D, tin, i );
		tri.Intersect( r );
		hit[i] = r.objIdx, tout[i] = r.t;
	}
}
void ref_tri_derived( int n, const float* v9, float* N, float* centroid )
{
	for (int i = 0; i < n; i++)
	{
		const float* v = v9 + 9 * i;
		Triangle tri( 7, float3( v[0], v[1], v[2] ), float3( v[3], v[4], v[5] ), float3( v[6], v[7], v[8] ), dummyMat() );
		memcpy( N + 3 * i, &tri.N, 12 ), memcpy( centroid + 3 * i, &tri.centroid, 12 );
	}
}
void ref_sphere_intersect( int n, const float* O, const float* D, const float* tin, const float* pr4, int* hit, float* tout )
{
	for (int i = 0; i < n; i++)
	{
		Sphere s( 7, float3( pr4[4 * i], pr4[4 * i + 1], pr4[4 * i + 2] ), pr4[4 * i + 3], dummyMat() );
		Ray r = mkray( O, D, tin, i );
		s.Intersect( r );
		hit[i] = r.objIdx, tout[i] = r.t;
	}
}
void ref_plane_intersect( int n, const float* O, const float* D, const float* tin, const float* nd4, int* hit, float* tout )
{
	for (int i = 0; i < n; i++)
	{
		Plane p( 7, float3( nd4[4 * i], nd4[4 * i + 1], nd4[4 * i + 2] ), nd4[4 * i + 3], dummyMat() );
		Ray r = mkray( O, D, tin, i );
		p.Intersect( r );
		hit[i] = r.objIdx, tout[i] = r.t;
	}
}
// quad/cube of the default scene (their transforms are the interesting data)
void ref_quad_intersect( int n, const float* O, const float* D, const float* tin, int* hit, float* tout )
{
	for (int i = 0; i < n; i++) { Ray r = mkray( O, D, tin, i ); R->scene.quads[0].Intersect( r ); hit[i] = r.objIdx, tout[i] = r.t; }
}
void ref_cube_intersect( int n, const float* O, const float* D, const float* tin, int* hit, float* tout )
{
	for (int i = 0; i < n; i++) { Ray r = mkray( O, D, tin, i ); R->scene.cubes[0].Intersect( r ); hit[i] = r.objIdx, tout[i] = r.t; }
}
// the three slab tests (bvh.h:443-494); out3 = {scalar bool, stack float, simd lane0 mask != 0}
void ref_aabb( int n, const float* O, const float* D, const float* tin, const float* bmin, const float* bmax, int* outScalar, float* outStack, int* outSimd )
{
	BVH bvh;
	bvh.qbvhNode = new SIMD_BVH_Node[1];
	for (int i = 0; i < n; i++)
	{
		Ray r = mkray( O, D, tin, i );
		float3 mn( bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2] ), mx( bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2] );
		outScalar[i] = bvh.IntersectAABB( r, mn, mx ) ? 1 : 0;
		outStack[i] = bvh.IntersectAABBStack( r, mn, mx );
		SIMD_BVH_Node& q = bvh.qbvhNode[0];
		for (int k = 0; k < 4; k++) q.bminx[k] = mn.x, q.bminy[k] = mn.y, q.bminz[k] = mn.z, q.bmaxx[k] = mx.x, q.bmaxy[k] = mx.y, q.bmaxz[k] = mx.z;
		Ref128 m = bvh.IntersectAABBSIMD( r, 0, _mm_set1_ps( r.t ), _mm_set1_ps( r.O.x ), _mm_set1_ps( r.rD.x ), _mm_set1_ps( r.O.y ), _mm_set1_ps( r.rD.y ), _mm_set1_ps( r.O.z ), _mm_set1_ps( r.rD.z ) );
		outSimd[i] = m.m128_i32[0] != 0;
	}
	delete[] bvh.qbvhNode;
}

// ---- custom BVH over caller triangles/spheres (bvh.h:66-74) ------------------------------------
struct RefBVH { Triangle* tris; Sphere* sph; int nt, ns; BVH* bvh; };
void* ref_bvh_create( const float* v9, int nt, const float* pr4, int ns )
{
	RefBVH* b = new RefBVH;
	b->nt = nt, b->ns = ns;
	b->tris = nt ? new Triangle[nt] : 0, b->sph = ns ? new Sphere[ns] : 0;
	for (int i = 0; i < ns; i++) b->sph[i] = Sphere( i, float3( pr4[4 * i], pr4[4 * i + 1], pr4[4 * i + 2] ), pr4[4 * i + 3], dummyMat() );
	for (int i = 0; i < nt; i++)
	{
		const float* v = v9 + 9 * i;
		b->tris[i] = Triangle( ns + i, float3( v[0], v[1], v[2] ), float3( v[3], v[4], v[5] ), float3( v[6], v[7], v[8] ), dummyMat() );
	}
	b->bvh = new BVH( b->sph, ns, 0, 0, b->tris, nt );
	return b;
}
void ref_bvh_info( void* h, int* out ) { RefBVH* b = (RefBVH*)h; out[0] = (int)b->bvh->nodesUsed, out[1] = (int)b->bvh->nodesUsedQBVH; }
void ref_bvh_nodes( void* h, void* nodes32, int* primIdx )
{
	RefBVH* b = (RefBVH*)h;
	memcpy( nodes32, b->bvh->gpuBvhNode, sizeof( GPUBVHNode ) * b->bvh->nodesUsed );
	memcpy( primIdx, b->bvh->primitiveIdx, 4 * (b->nt + b->ns) );
}
// mode 0 = IntersectBVH (recursive), 1 = IntersectBVHStack (ordered), 2 = IntersectQBVH; objIdx = sphere i / ns + triangle i
void ref_bvh_trace( void* h, int mode, int n, const float* O, const float* D, const float* tin, int* idx, float* tout )
{
	RefBVH* b = (RefBVH*)h;
#pragma omp parallel for schedule( dynamic, 256 )
	for (int i = 0; i < n; i++)
	{
		Ray r = mkray( O, D, tin, i );
		if (mode == 0) b->bvh->IntersectBVH( r ); else if (mode == 1) b->bvh->IntersectBVHStack( r ); else b->bvh->IntersectQBVH( r );
		idx[i] = r.objIdx, tout[i] = r.t;
	}
}
void ref_bvh_occluded( void* h, int mode, int n, const float* O, const float* D, const float* tin, int* occ )
{
	RefBVH* b = (RefBVH*)h;
#pragma omp parallel for schedule( dynamic, 256 )
	for (int i = 0; i < n; i++)
	{
		Ray r = mkray( O, D, tin, i );
		occ[i] = (mode == 2 ? b->bvh->IntersectQBVHShadowRay( r, false ) : b->bvh->IntersectBVHShadowRay( r, false )) ? 1 : 0;
	}
}
void ref_bvh_destroy( void* h ) { RefBVH* b = (RefBVH*)h; delete[] b->tris; delete[] b->sph; delete b; }

// light sampling and hemisphere sampling with an explicit stream (scene.h:317-341, 517-532)
void ref_random_point_on_light( uint s0, float* out /* A, Nl xyz, P xyz, emission xyz */, uint* s1 )
{
	seed = s0;
	auto r = R->scene.RandomPointOnLight();
	out[0] = std::get<0>( r );
	memcpy( out + 1, &std::get<1>( r ), 12 ), memcpy( out + 4, &std::get<2>( r ), 12 ), memcpy( out + 7, &std::get<3>( r ), 12 );
	*s1 = seed;
}
void ref_diffuse_reflection( uint s0, const float* N, float* out, uint* s1 )
{
	seed = s0;
	float3 d = R->scene.DiffuseReflection( float3( N[0], N[1], N[2] ) );
	memcpy( out, &d, 12 );
	*s1 = seed;
}
void ref_set_threads( int n ) { omp_set_num_threads( n ); }

} // extern "C"
