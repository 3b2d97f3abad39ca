// rt_device.h -- gfx950 device code shared by the kernels in rtmi.hip: strict-IEEE vector helpers, xorshift32,
// primitive intersection, slab tests, ordered BVH traversal with a per-lane LDS stack, hit attributes.
//
// Written for CDNA4 wave64.  Everything here must produce the same binary32 results as the reference CPU path
// (primitive.h, bvh.h, template/scene.h, precomp.h math), so this translation unit is compiled with
// -ffp-contract=off (no FMA fusion) and relies on hipcc's default correctly-rounded f32 divide and sqrt.
// Operation order follows the cited reference lines; do not "simplify" expressions here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_shared.h"


__device__ __forceinline__ f3 mk3( float x, float y, float z ) { f3 r; r.x = x, r.y = y, r.z = z; return r; }
__device__ __forceinline__ f3 mk3( float4 v ) { return mk3( v.x, v.y, v.z ); }
__device__ __forceinline__ f3 operator+( f3 a, f3 b ) { return mk3( a.x + b.x, a.y + b.y, a.z + b.z ); }
__device__ __forceinline__ f3 operator-( f3 a, f3 b ) { return mk3( a.x - b.x, a.y - b.y, a.z - b.z ); }
__device__ __forceinline__ f3 operator-( f3 a ) { return mk3( -a.x, -a.y, -a.z ); }
__device__ __forceinline__ f3 operator*( f3 a, f3 b ) { return mk3( a.x * b.x, a.y * b.y, a.z * b.z ); }
__device__ __forceinline__ f3 operator*( f3 a, float b ) { return mk3( a.x * b, a.y * b, a.z * b ); }
__device__ __forceinline__ f3 operator*( float b, f3 a ) { return mk3( b * a.x, b * a.y, b * a.z ); }
__device__ __forceinline__ f3 operator/( f3 a, float b ) { return mk3( a.x / b, a.y / b, a.z / b ); }
__device__ __forceinline__ float dot3( f3 a, f3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // precomp.h:822
__device__ __forceinline__ f3 cross3( f3 a, f3 b ) { return mk3( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x ); } // :875
__device__ __forceinline__ f3 normalize3( f3 v ) { float inv = 1.0f / sqrtf( dot3( v, v ) ); return v * inv; } // :490, 844
__device__ __forceinline__ float magnitude3( f3 v ) { return sqrtf( v.x * v.x + v.y * v.y + v.z * v.z ); }     // :848
__device__ __forceinline__ float sqrf_( float x ) { return x * x; }
// std::min / std::max as the reference's scalar code resolves them (precomp.h:31): NaN-propagation is asymmetric
__device__ __forceinline__ float smin( float a, float b ) { return ( b < a ) ? b : a; }
__device__ __forceinline__ float smax( float a, float b ) { return ( a < b ) ? b : a; }

// row-major mat4 rows 0..2 as three float4 (template.cpp:1021-1035)
struct m34 { float4 r0, r1, r2; };
__device__ __forceinline__ f3 xform_pos( f3 a, const m34& M )
{
	return mk3( M.r0.x * a.x + M.r0.y * a.y + M.r0.z * a.z + M.r0.w * 1.0f,
		M.r1.x * a.x + M.r1.y * a.y + M.r1.z * a.z + M.r1.w * 1.0f,
		M.r2.x * a.x + M.r2.y * a.y + M.r2.z * a.z + M.r2.w * 1.0f );
}
__device__ __forceinline__ f3 xform_vec( f3 a, const m34& M )
{
	return mk3( M.r0.x * a.x + M.r0.y * a.y + M.r0.z * a.z + M.r0.w * 0.0f,
		M.r1.x * a.x + M.r1.y * a.y + M.r1.z * a.z + M.r1.w * 0.0f,
		M.r2.x * a.x + M.r2.y * a.y + M.r2.z * a.z + M.r2.w * 0.0f );
}

// xorshift32 + RandomFloat (template.cpp:883-890)
__device__ __forceinline__ uint32_t rng_next( uint32_t& s ) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ float rng_float( uint32_t& s ) { return (float)rng_next( s ) * 2.3283064365387e-10f; }
__device__ __forceinline__ uint32_t wang_hash( uint32_t s ) // template.cpp:871-878
{
	s = ( s ^ 61u ) ^ ( s >> 16 );
	s *= 9u, s = s ^ ( s >> 4 );
	s *= 0x27d4eb2du;
	s = s ^ ( s >> 15 );
	return s;
}
__device__ __forceinline__ uint32_t sample_seed( uint32_t s0, uint32_t s )
{
	uint32_t st = wang_hash( s0 + s * 0x9E3779B9u );
	return st ? st : 1u;
}


struct RayHit { float t; int obj; int pos; };

// A division whose result cannot be accepted need not be made.  For num >= 0, den >= 0: true only if fl( num / den ) >= limit, so that
// `t < ray.t` fails whatever the quotient's bits are.  q lies above the real product limit * den whatever its two roundings did
// ((1 - 2^-24)^2 (1 + 2^-22) > 1), so num >= q means num / den > limit in real arithmetic, and rounding is monotonic.  A product that
// underflowed bounds nothing (q > 1e-30f); inf and NaN compare false and leave the decision to the division.  (Checked against the
// division itself on 1.6 x 10^8 float triples, numerators within 8 ulp of limit * den included.)  Only the any-hit tests of the kernel
// that creates a shadow ray use it (CERT): there every lane of a wave skips together -- a shadow ray ends 0.002 before the light, inside
// the room -- and a skipped division is ~10 instructions less; closest-hit rays need the quotient of the plane they do hit.
__device__ __forceinline__ bool quotient_reaches( float num, float den, float limit )
{
	const float q = ( limit * den ) * 1.00000023841857910156f; // 1 + 2^-22
	return q > 1e-30f && num >= q;
}

// ---- primitive intersection (closest-hit forms) ---------------------------------------------------------------
// Quad::Intersect, primitive.h:284-295
template <bool CERT = false>
__device__ __forceinline__ void isect_quad_rows( const m34& M, float size, int obj, f3 O, f3 D, RayHit& h )
{
	if (CERT)
	{
		// row 1 alone gives t = o.y / -d.y (the same two sums as below): no hit unless o.y and -d.y have one sign (else t <= 0, +-0,
		// +-inf or NaN) and the quotient is below ray.t
		const float oy = M.r1.x * O.x + M.r1.y * O.y + M.r1.z * O.z + M.r1.w * 1.0f, dy = M.r1.x * D.x + M.r1.y * D.y + M.r1.z * D.z + M.r1.w * 0.0f;
		if (!( ( oy > 0 && dy < 0 ) || ( oy < 0 && dy > 0 ) )) return;
		if (quotient_reaches( fabsf( oy ), fabsf( dy ), h.t )) return;
	}
	const f3 o = xform_pos( O, M ), d = xform_vec( D, M );
	const float t = o.y / -d.y;
	if (t < h.t && t > 0)
	{
		const f3 I = o + t * d;
		if (I.x > -size && I.x < size && I.z > -size && I.z < size) h.t = t, h.obj = obj, h.pos = -1;
	}
}
template <bool CERT = false>
__device__ __forceinline__ void isect_quad( const DScene& s, int i, f3 O, f3 D, RayHit& h ) // (i == 0: every caller)
{
	m34 M; M.r0 = s.quad0[0], M.r1 = s.quad0[1], M.r2 = s.quad0[2];
	isect_quad_rows<CERT>( M, s.quad0[3].x, i, O, D, h );
}
// Cube::Intersect, primitive.h:210-236
__device__ __forceinline__ void isect_cube_rows( const m34& M, float4 b0, float4 b1, int obj, f3 O, f3 D, RayHit& h );
__device__ __forceinline__ void isect_cube( const DScene& s, int i, f3 O, f3 D, RayHit& h )
{
	m34 M; M.r0 = s.cubes[8 * i], M.r1 = s.cubes[8 * i + 1], M.r2 = s.cubes[8 * i + 2];
	isect_cube_rows( M, s.cubes[8 * i + 3], s.cubes[8 * i + 4], s.n_quads + s.n_spheres + i, O, D, h );
}
__device__ __forceinline__ void isect_cube_rows( const m34& M, float4 b0, float4 b1, int obj, f3 O, f3 D, RayHit& h )
{
	const f3 o = xform_pos( O, M ), d = xform_vec( D, M );
	const float rDx = 1 / d.x, rDy = 1 / d.y, rDz = 1 / d.z;
	const bool sx = d.x < 0, sy = d.y < 0, sz = d.z < 0;
	float tmin = ( ( sx ? b1.x : b0.x ) - o.x ) * rDx;
	float tmax = ( ( sx ? b0.x : b1.x ) - o.x ) * rDx;
	const float tymin = ( ( sy ? b1.y : b0.y ) - o.y ) * rDy;
	const float tymax = ( ( sy ? b0.y : b1.y ) - o.y ) * rDy;
	if (tmin > tymax || tymin > tmax) return;
	tmin = smax( tmin, tymin ), tmax = smin( tmax, tymax );
	const float tzmin = ( ( sz ? b1.z : b0.z ) - o.z ) * rDz;
	const float tzmax = ( ( sz ? b0.z : b1.z ) - o.z ) * rDz;
	if (tmin > tzmax || tzmin > tmax) return;
	tmin = smax( tmin, tzmin ), tmax = smin( tmax, tzmax );
	if (tmin > 0) { if (tmin < h.t) h.t = tmin, h.obj = obj, h.pos = -1; }
	else if (tmax > 0) { if (tmax < h.t) h.t = tmax, h.obj = obj, h.pos = -1; }
}
// Plane::Intersect, primitive.h:154-158
__device__ __forceinline__ void isect_plane_nd( float4 p, int obj, f3 O, f3 D, RayHit& h )
{
	const f3 N = mk3( p );
	const float t = -( dot3( O, N ) + p.w ) / ( dot3( D, N ) );
	if (t < h.t && t > 0) h.t = t, h.obj = obj, h.pos = -1;
}
__device__ __forceinline__ void isect_plane( const DScene& s, int i, f3 O, f3 D, RayHit& h ) { isect_plane_nd( s.planes[i], s.n_quads + s.n_spheres + s.n_cubes + i, O, D, h ); }
// Two opposite axis planes at once: plane k = (+e, d1), plane k + 1 = (-e, d2); o, dd = the ray's origin and direction
// component on that axis.  With N = +-e the reference's dot products are the component itself (x * 1 + y * 0 + z * 0, exact for
// finite rays up to the sign of a zero, which no outcome depends on: a zero numerator gives t = +-0, a zero denominator
// t = +-inf or NaN, none of which passes `t < ray.t && t > 0`), so a = (O.N + d) and b = D.N of primitive.h:156 are o + d1, dd
// and d2 - o, -dd.  An origin strictly between the two planes (a1 > 0 and a2 > 0: every ray that starts inside the room) can
// only hit the one it faces -- t = -a / b > 0 needs b < 0 -- so ONE division serves the pair; the other plane's t is negative
// and the reference rejects it.  Any other origin takes both tests as written.  Same t bits, same accept order (k before k + 1).
template <bool CERT = false>
__device__ __forceinline__ void isect_plane_pair( float o, float dd, float d1, float d2, int obj, RayHit& h )
{
	const float a1 = o + d1, a2 = d2 - o;
	if (a1 > 0 && a2 > 0)
	{
		const bool first = dd < 0;
		if (CERT && quotient_reaches( first ? a1 : a2, fabsf( dd ), h.t )) return; // t = a / |dd| >= ray.t: not closer
		const float t = -( first ? a1 : a2 ) / ( first ? dd : -dd );
		if (( dd < 0 || dd > 0 ) && t < h.t && t > 0) h.t = t, h.obj = obj + ( first ? 0 : 1 ), h.pos = -1;
	}
	else
	{
		const float t1 = -a1 / dd;
		if (t1 < h.t && t1 > 0) h.t = t1, h.obj = obj, h.pos = -1;
		const float t2 = -a2 / -dd;
		if (t2 < h.t && t2 > 0) h.t = t2, h.obj = obj + 1, h.pos = -1;
	}
}
// the plane loop of BVH::IntersectBVH (bvh.h:498-502): every plane in order, each accepted when closer than what came before.
// (Scene::IsOccluded's loop, bvh.h:642-646 with the early out, asks only WHETHER some plane is closer than the ray's end: the
// same function answers that through h.obj.)
template <bool CERT = false>
__device__ __forceinline__ void isect_planes( const DScene& s, f3 O, f3 D, RayHit& h )
{
	if (s.plane_pairs)
	{
		const int base = s.n_quads + s.n_spheres + s.n_cubes;
		isect_plane_pair<CERT>( O.x, D.x, s.pair_d[0], s.pair_d[1], base, h );
		isect_plane_pair<CERT>( O.y, D.y, s.pair_d[2], s.pair_d[3], base + 2, h );
		isect_plane_pair<CERT>( O.z, D.z, s.pair_d[4], s.pair_d[5], base + 4, h );
	}
	else for (int i = 0; i < s.n_planes; i++) isect_plane( s, i, O, D, h );
}
// the non-BVH part of Scene::FindNearest + the plane loop of BVH::IntersectBVH (scene.h:361-362, bvh.h:498-502)
__device__ __forceinline__ void isect_analytic( const DScene& s, f3 O, f3 D, RayHit& h )
{
	if (s.n_quads > 0) isect_quad( s, 0, O, D, h );   // the reference tests quads[0] and cubes[0] only
	if (s.n_cubes > 0) isect_cube( s, 0, O, D, h );
	isect_planes( s, O, D, h );
}

// One leaf-ordered primitive record against the ray.  Accepts a hit when it is closer, or equally far but
// earlier in leaf order than the incumbent BVH hit: the reference's recursive left-first traversal visits leaves
// in increasing primitiveIdx position and keeps the first of equal distances (strict `<`, primitive.h:68,116,122),
// so "smallest (t, position)" is what it returns whatever the visiting order.
template <bool ANY>
__device__ __forceinline__ bool isect_record( float4 a, float4 b, float4 c, int pos, f3 O, f3 D, RayHit& h );
template <bool ANY>
__device__ __forceinline__ bool isect_prim( const DScene& s, int pos, f3 O, f3 D, RayHit& h )
{
	const float4 a = s.prims[3 * pos], b = s.prims[3 * pos + 1];
	// (a sphere's record has no third quarter worth fetching)
	const float4 c = __float_as_int( a.w ) == 0 ? s.prims[3 * pos + 2] : make_float4( 0, 0, 0, 0 );
	return isect_record<ANY>( a, b, c, pos, O, D, h );
}
// one 48-byte primitive record {p1, tag 0} {e1, objIdx} {e2, -} (triangle) or {pos, tag 1} {r2, -, -, objIdx} (sphere) against the ray
template <bool ANY>
__device__ __forceinline__ bool isect_record( float4 a, float4 b, float4 c, int pos, f3 O, f3 D, RayHit& h )
{
	if (__float_as_int( a.w ) == 0)
	{
		// Triangle::Intersect, primitive.h:50-72 (edges precomputed on upload with the same two subtractions)
		const f3 p1 = mk3( a ), edge1 = mk3( b ), edge2 = mk3( c );
		const f3 hh = cross3( D, edge2 );
		const float det = dot3( edge1, hh );
		if (det > -0.0001f && det < 0.0001f) return false;
		const float f = 1 / det;
		const f3 sv = O - p1;
		const float u = f * dot3( sv, hh );
		if (u < 0 || u > 1) return false;
		const f3 q = cross3( sv, edge1 );
		const float v = f * dot3( D, q );
		if (v < 0 || u + v > 1) return false;
		const float t = f * dot3( edge2, q );
		if (t > 0.0001f && ( t < h.t || ( !ANY && t == h.t && pos < h.pos ) ))
		{
			h.t = t, h.obj = __float_as_int( b.w ), h.pos = pos;
			return true;
		}
		return false;
	}
	// Sphere::Intersect, primitive.h:108-127
	const f3 oc = O - mk3( a );
	const float bb = dot3( oc, D );
	const float cc = dot3( oc, oc ) - b.x;
	float d = bb * bb - cc;
	if (d <= 0) return false;
	d = sqrtf( d );
	float t = -bb - d;
	if (t > 0 && ( t < h.t || ( !ANY && t == h.t && pos < h.pos ) )) { h.t = t, h.obj = __float_as_int( b.w ), h.pos = pos; return true; }
	t = d - bb;
	if (t > 0 && ( t < h.t || ( !ANY && t == h.t && pos < h.pos ) )) { h.t = t, h.obj = __float_as_int( b.w ), h.pos = pos; return true; }
	return false;
}

// BVH::IntersectAABB (bvh.h:443-452) for both children of a pair.  EXACT reproduces std::min/std::max NaN behaviour
// (needed only when a reciprocal direction component is infinite: 0*inf = NaN); otherwise v_min/v_max
// give identical results on NaN-free data.
template <bool EXACT>
__device__ __forceinline__ void slab( float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 O, f3 rD, float& tmin, float& tmax )
{
	const float tx1 = ( bminx - O.x ) * rD.x, tx2 = ( bmaxx - O.x ) * rD.x;
	const float ty1 = ( bminy - O.y ) * rD.y, ty2 = ( bmaxy - O.y ) * rD.y;
	const float tz1 = ( bminz - O.z ) * rD.z, tz2 = ( bmaxz - O.z ) * rD.z;
	if (EXACT)
	{
		tmin = smin( tx1, tx2 ), tmax = smax( tx1, tx2 );
		tmin = smax( tmin, smin( ty1, ty2 ) ), tmax = smin( tmax, smax( ty1, ty2 ) );
		tmin = smax( tmin, smin( tz1, tz2 ) ), tmax = smin( tmax, smax( tz1, tz2 ) );
	}
	else
	{
		tmin = fmaxf( fmaxf( fminf( tx1, tx2 ), fminf( ty1, ty2 ) ), fminf( tz1, tz2 ) );
		tmax = fminf( fminf( fmaxf( tx1, tx2 ), fmaxf( ty1, ty2 ) ), fmaxf( tz1, tz2 ) );
	}
}

// Streams that are touched once per bounce (the path state: 3.5 GB per set at full batch size, far beyond L2 and the Infinity Cache).
// RTMI_NT: load / store them with the non-temporal hint so that they do not displace the tree the walks beside k_shade re-read.
#ifndef RTMI_NT_LD
#define RTMI_NT_LD 0
#endif
#ifndef RTMI_NT_ST
#define RTMI_NT_ST 1
#endif
typedef float v4f_t __attribute__( ( ext_vector_type( 4 ) ) );
__device__ __forceinline__ float4 ld_stream( const float4* p )
{
#if RTMI_NT_LD
	const v4f_t v = __builtin_nontemporal_load( (const v4f_t*)p );
	return make_float4( v.x, v.y, v.z, v.w );
#else
	return *p;
#endif
}
__device__ __forceinline__ uint32_t ld_stream( const uint32_t* p )
{
#if RTMI_NT_LD
	return __builtin_nontemporal_load( p );
#else
	return *p;
#endif
}
__device__ __forceinline__ void st_stream( float4* p, float4 a )
{
#if RTMI_NT_ST
	v4f_t v; v.x = a.x, v.y = a.y, v.z = a.z, v.w = a.w;
	__builtin_nontemporal_store( v, (v4f_t*)p );
#else
	*p = a;
#endif
}
__device__ __forceinline__ void st_stream( uint32_t* p, uint32_t a )
{
#if RTMI_NT_ST
	__builtin_nontemporal_store( a, p );
#else
	*p = a;
#endif
}

struct TravCount { uint32_t pairs, prims, wide, confirms; }; // pair records / leaf primitive records / 4-wide records fetched, leaf boxes fetched to confirm a hit

// Ordered (near child first) traversal of the pair array with a per-lane stack in LDS.
//   stack: this lane's column, entries at stack[level * stride]
//   ANY:   stop at the first hit (Scene::IsOccluded / IntersectBVHShadowRay, bvh.h:640-688); the box test uses the
//          fixed ray.t like the reference.  Otherwise closest hit (IntersectBVH, bvh.h:507-530) with the
//          (t, position) rule above; boxes are kept while tmin <= t so that an equally distant, earlier
//          primitive is still reached.
// Returns true iff ANY and a hit was found.
template <bool ANY, bool EXACT, bool COUNT>
__device__ __forceinline__ bool traverse( const DScene& s, f3 O, f3 D, f3 rD, RayHit& h, uint32_t* stack, const int stride, TravCount& cnt )
{
	float tmin, tmax;
	slab<EXACT>( s.root_min[0], s.root_min[1], s.root_min[2], s.root_max[0], s.root_max[1], s.root_max[2], O, rD, tmin, tmax );
	uint32_t cur = ( tmax >= tmin && tmin < h.t && tmax > 0 ) ? s.root_ref : RT_DONE;
	int sp = 0;
	while (true)
	{
		while (!( cur & RT_LEAF ))
		{
			const float4 q0 = s.pairs[4 * cur], q1 = s.pairs[4 * cur + 1], q2 = s.pairs[4 * cur + 2], q3 = s.pairs[4 * cur + 3];
			if (COUNT) cnt.pairs++;
			float tminL, tmaxL, tminR, tmaxR;
			slab<EXACT>( q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, O, rD, tminL, tmaxL );
			slab<EXACT>( q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, O, rD, tminR, tmaxR );
			const bool hitL = tmaxL >= tminL && ( ANY ? tminL < h.t : tminL <= h.t ) && tmaxL > 0;
			const bool hitR = tmaxR >= tminR && ( ANY ? tminR < h.t : tminR <= h.t ) && tmaxR > 0;
			const uint32_t refL = __float_as_uint( q1.z ), refR = __float_as_uint( q3.z );
			if (hitL && hitR)
			{
				const bool swap = tminR < tminL;
				stack[sp * stride] = swap ? refL : refR;
				sp++;
				cur = swap ? refR : refL;
			}
			else if (hitL) cur = refL;
			else if (hitR) cur = refR;
			else if (sp > 0) cur = stack[--sp * stride];
			else cur = RT_DONE;
		}
		if (cur == RT_DONE) break;
		const int first = (int)( cur & RT_LEAF_FIRST_MASK ), count = (int)( ( cur >> RT_LEAF_FIRST_BITS ) & ( RT_LEAF_MAX_COUNT - 1 ) ) + 1;
		for (int i = 0; i < count; i++)
		{
			if (COUNT) cnt.prims++;
			const bool hit = isect_prim<ANY>( s, first + i, O, D, h );
			if (ANY && hit) return true;
		}
		cur = sp > 0 ? stack[--sp * stride] : RT_DONE;
	}
	return false;
}

// ---- the reference's default traversal, for the rays on which it differs from the binary one (option qbvh) --------------------
// IntersectQBVH / IntersectQBVHShadowRay (bvh.h:579-638, 690-758) over the reference's own QBVH nodes, statement for statement:
// the four children of a node are box-tested at once with the ray's t as it was BEFORE the walk (ray_t is never refreshed,
// bvh.h:590, 603) and with _mm_min_ps / _mm_max_ps, which return their SECOND operand when either is NaN (bvh.h:465-486); children
// are visited in slot order, a node is left at its first child == -1; leaf primitives are judged with a plain `t < ray.t`, so the
// first one visited keeps an equal distance; the primitive root's own box is never tested.  Only rays with an infinite reciprocal
// direction component come here (0 * inf = NaN is where the two traversals of the reference part ways): a few in a
// thousand camera rays, so one lane walks one ray with an explicit frame stack (one entry per level:
// node << 8 | prevOccluded << 7 | box mask << 3 | next slot).  Returns true iff ANY and occluded.
__device__ __forceinline__ float sse_min( float a, float b ) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max( float a, float b ) { return a > b ? a : b; }
// `prune`: a child is also left out when the ray provably enters its box beyond `prune` -- the entry distance over the axes whose
// reciprocal is finite is a lower bound of every hit inside the box, NaN-free -- which the caller sets a little beyond the best hit so
// far (closest hit) or to +inf (any hit).  IntersectQBVH never refreshes ray_t and so visits every box on the ray's line to the far
// wall, thousands of nodes for a ray through a dense mesh; a box entered beyond the best hit cannot hold a primitive that passes
// `t < ray.t`, so the outcome is the same, hit for hit and tie for tie.
__device__ __forceinline__ uint32_t qbvh_boxes( const DScene& s, uint32_t node, f3 O, f3 rD, float ray_t, float prune )
{
	const bool fx = fabsf( rD.x ) < INFINITY, fy = fabsf( rD.y ) < INFINITY, fz = fabsf( rD.z ) < INFINITY;
	const float4* q = s.qnodes + 8 * (size_t)node;
	const float4 nx = q[0], xx = q[1], ny = q[2], xy = q[3], nz = q[4], xz = q[5];
	const float bnx[4] = { nx.x, nx.y, nx.z, nx.w }, bxx[4] = { xx.x, xx.y, xx.z, xx.w }, bny[4] = { ny.x, ny.y, ny.z, ny.w };
	const float bxy[4] = { xy.x, xy.y, xy.z, xy.w }, bnz[4] = { nz.x, nz.y, nz.z, nz.w }, bxz[4] = { xz.x, xz.y, xz.z, xz.w };
	uint32_t mask = 0;
#pragma unroll
	for (int l = 0; l < 4; l++)
	{
		const float tx1 = ( bnx[l] - O.x ) * rD.x, tx2 = ( bxx[l] - O.x ) * rD.x;
		float tmin = sse_min( tx1, tx2 ), tmax = sse_max( tx1, tx2 );
		const float ty1 = ( bny[l] - O.y ) * rD.y, ty2 = ( bxy[l] - O.y ) * rD.y;
		tmin = sse_max( tmin, sse_min( ty1, ty2 ) ), tmax = sse_min( tmax, sse_max( ty1, ty2 ) );
		const float tz1 = ( bnz[l] - O.z ) * rD.z, tz2 = ( bxz[l] - O.z ) * rD.z;
		tmin = sse_max( tmin, sse_min( tz1, tz2 ) ), tmax = sse_min( tmax, sse_max( tz1, tz2 ) );
		float enter = -INFINITY;
		if (fx) enter = fmaxf( enter, fminf( tx1, tx2 ) );
		if (fy) enter = fmaxf( enter, fminf( ty1, ty2 ) );
		if (fz) enter = fmaxf( enter, fminf( tz1, tz2 ) );
		if (tmax >= tmin && tmin < ray_t && tmax > 0 && !( enter > prune )) mask |= 1u << l;
	}
	return mask;
}
template <bool ANY>
__device__ __forceinline__ bool traverse_qbvh( const DScene& s, f3 O, f3 D, f3 rD, RayHit& h, uint32_t* stack, const int stride )
{
	const float ray_t = h.t;
	// (0.1 % and 0.001 beyond the best hit: a hit's t and its box's entry distance are each within a few ulp of the same exact number)
	#define QBVH_PRUNE ( ANY ? INFINITY : h.t * 1.001f + 0.001f )
	uint32_t node = s.qroot, slot = 0, mask = qbvh_boxes( s, node, O, rD, ray_t, QBVH_PRUNE );
	bool prev = false;   // ANY: prevOccluded of the current call (bvh.h:724)
	bool anyHit = false; // ANY: ray.objIdx != -1
	int sp = 0, best = -1;
	for (;;)
	{
		bool leave = false, result = false; // leave the current call, with this return value (ANY)
		if (slot < 4)
		{
			const int4 ch = *(const int4*)( s.qnodes + 8 * (size_t)node + 6 ), cn = *(const int4*)( s.qnodes + 8 * (size_t)node + 7 );
			const int c = slot == 0 ? ch.x : slot == 1 ? ch.y : slot == 2 ? ch.z : ch.w, n = slot == 0 ? cn.x : slot == 1 ? cn.y : slot == 2 ? cn.z : cn.w;
			if (c == -1) leave = true, result = false; // bvh.h:611 `return`, :727 `return false`
			else if (!( mask >> slot & 1u )) slot++;
			else if (n > 0)
			{
				for (int j = 0; j < n && !leave; j++)
				{
					h.pos = -1; // plain `t < ray.t` (primitive.h:68, 116, 122): no (t, position) rule here, the visiting order decides
					if (isect_prim<ANY>( s, c + j, O, D, h )) best = c + j, anyHit = true;
					if (ANY && anyHit) leave = true, result = true; // bvh.h:738, 744: `if (ray.objIdx != -1) return true`
				}
				slot++;
			}
			else if (ANY && prev) slot++; // `prevOccluded || ...`: the call is not made
			else
			{
				stack[sp * stride] = node << 8 | ( prev ? 128u : 0u ) | mask << 3 | ( slot + 1 );
				sp++;
				node = (uint32_t)c, slot = 0, prev = false, mask = qbvh_boxes( s, node, O, rD, ray_t, QBVH_PRUNE );
			}
		}
		else leave = true, result = prev; // bvh.h:757
		if (leave)
		{
			if (sp == 0) { h.pos = best; return ANY && result; }
			const uint32_t e = stack[--sp * stride];
			node = e >> 8, slot = e & 7u, mask = ( e >> 3 ) & 15u;
			prev = ( e & 128u ) != 0 || ( ANY && result );
		}
	}
}
#undef QBVH_PRUNE
// a ray with an infinite reciprocal direction component: the reference's default traversal where the scene carries it (option
// qbvh), else its binary traversal with the scalar slab test's NaN propagation.  QBVH is a template parameter of the kernels in
// which one lane or wave runs a whole path or ray (tails, hooks, Whitted): the instantiations a handle without the option launches
// carry none of traverse_qbvh's code or registers.  k_trace has no such parameter: there these rays are left to k_qbvh_fix.
template <bool ANY, bool QBVH>
__device__ __forceinline__ bool walk_inexact( const DScene& s, f3 O, f3 D, f3 rD, RayHit& h, uint32_t* stack, const int stride )
{
	if (QBVH) return traverse_qbvh<ANY>( s, O, D, rD, h, stack, stride );
	TravCount tc = {};
	return traverse<ANY, true, false>( s, O, D, rD, h, stack, stride, tc );
}

// ---- the same traversal cut into steps, for the persistent kernels that refill idle lanes with new rays ----------
template <bool EXACT>
__device__ __forceinline__ uint32_t bvh_enter( const DScene& s, f3 O, f3 rD, float t )
{
	float tmin, tmax;
	slab<EXACT>( s.root_min[0], s.root_min[1], s.root_min[2], s.root_max[0], s.root_max[1], s.root_max[2], O, rD, tmin, tmax );
	return ( tmax >= tmin && tmin < t && tmax > 0 ) ? s.root_ref : RT_DONE;
}
// one interior node: tests both children of pair `cur`, descends into the nearer hit child, pushes the other
template <bool ANY, bool EXACT>
__device__ __forceinline__ uint32_t interior_step( const DScene& s, uint32_t cur, f3 O, f3 rD, float t, uint32_t* stack, const int stride, int& sp )
{
	const float4 q0 = s.pairs[4 * cur], q1 = s.pairs[4 * cur + 1], q2 = s.pairs[4 * cur + 2], q3 = s.pairs[4 * cur + 3];
	float tminL, tmaxL, tminR, tmaxR;
	slab<EXACT>( q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, O, rD, tminL, tmaxL );
	slab<EXACT>( q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, O, rD, tminR, tmaxR );
	const bool hitL = tmaxL >= tminL && ( ANY ? tminL < t : tminL <= t ) && tmaxL > 0;
	const bool hitR = tmaxR >= tminR && ( ANY ? tminR < t : tminR <= t ) && tmaxR > 0;
	const uint32_t refL = __float_as_uint( q1.z ), refR = __float_as_uint( q3.z );
	if (hitL && hitR)
	{
		const bool swap = tminR < tminL;
		stack[sp * stride] = swap ? refL : refR;
		sp++;
		return swap ? refR : refL;
	}
	if (hitL) return refL;
	if (hitR) return refR;
	if (sp > 0) return stack[--sp * stride];
	return RT_DONE;
}

// ---- 4-wide nodes with quantised child boxes (the walk kernels' interior nodes; built on upload from the pair array) ----
// One 64-byte record holds up to four children: {origin.xyz, scale exponents} {4 refs} {lo.x[4] lo.y[4] lo.z[4] hi.x[4]}
// {hi.y[4] hi.z[4] - -}, one byte per child and plane.  A plane is origin + byte * 2^e (one fma, checked on the host
// with the same fma to lie on or outside the child's exact box), and the slab test on it is the reference's formula
// (b - O) * rD, which is monotone in b: whenever the exact box of a node passes the test, so do these boxes of all its
// ancestors.  Passing the exact test of a LEAF's box therefore implies passing the exact tests of the whole chain the
// reference walks to reach it (bvh.h:507-530), and the set of primitives that may be accepted is unchanged as long as
// every accepted hit is confirmed against the exact box of its leaf (s.leafbox, in k_trace).  Halves the number of
// dependent fetches per ray.  Needs finite reciprocal directions (no NaN in the slab test): other rays walk the pairs.
__device__ __forceinline__ float wide_byte( uint32_t w, int c ) { return (float)( ( w >> ( 8 * c ) ) & 255u ); } // v_cvt_f32_ubyteN
template <bool ANY>
__device__ __forceinline__ uint32_t wide_step( const DScene& s, uint32_t cur, f3 O, f3 rD, float t, uint32_t* stack, const int stride, int& sp )
{
	const float4 q0 = s.wide[4 * cur], q1 = s.wide[4 * cur + 1], q2 = s.wide[4 * cur + 2], q3 = s.wide[4 * cur + 3];
	const uint32_t ex = __float_as_uint( q0.w );
	const float sx = __uint_as_float( ( ex & 255u ) << 23 ), sy = __uint_as_float( ( ( ex >> 8 ) & 255u ) << 23 ), sz = __uint_as_float( ( ( ex >> 16 ) & 255u ) << 23 );
	const uint32_t lox = __float_as_uint( q2.x ), loy = __float_as_uint( q2.y ), loz = __float_as_uint( q2.z );
	const uint32_t hix = __float_as_uint( q2.w ), hiy = __float_as_uint( q3.x ), hiz = __float_as_uint( q3.y );
	// entry and exit plane of each axis by the sign of the direction: min(tx1, tx2) of bvh.h:445-450 without the min
	const uint32_t nx = rD.x < 0 ? hix : lox, fx = rD.x < 0 ? lox : hix;
	const uint32_t ny = rD.y < 0 ? hiy : loy, fy = rD.y < 0 ? loy : hiy;
	const uint32_t nz = rD.z < 0 ? hiz : loz, fz = rD.z < 0 ? loz : hiz;
	const uint32_t ref[4] = { __float_as_uint( q1.x ), __float_as_uint( q1.y ), __float_as_uint( q1.z ), __float_as_uint( q1.w ) };
	uint32_t key[4];
	// two children at a time: gfx950 has packed binary32 fma / add / mul (v_pk_*_f32), each lane of a pair rounds as the scalar form
	typedef float v2f __attribute__( ( ext_vector_type( 2 ) ) );
	const v2f Sx = { sx, sx }, Sy = { sy, sy }, Sz = { sz, sz }, Bx = { q0.x, q0.x }, By = { q0.y, q0.y }, Bz = { q0.z, q0.z };
	const v2f Ox = { O.x, O.x }, Oy = { O.y, O.y }, Oz = { O.z, O.z }, Rx = { rD.x, rD.x }, Ry = { rD.y, rD.y }, Rz = { rD.z, rD.z };
#pragma unroll
	for (int c = 0; c < 4; c += 2)
	{
		const v2f qnx = { wide_byte( nx, c ), wide_byte( nx, c + 1 ) }, qfx = { wide_byte( fx, c ), wide_byte( fx, c + 1 ) };
		const v2f qny = { wide_byte( ny, c ), wide_byte( ny, c + 1 ) }, qfy = { wide_byte( fy, c ), wide_byte( fy, c + 1 ) };
		const v2f qnz = { wide_byte( nz, c ), wide_byte( nz, c + 1 ) }, qfz = { wide_byte( fz, c ), wide_byte( fz, c + 1 ) };
		const v2f tnx = ( __builtin_elementwise_fma( qnx, Sx, Bx ) - Ox ) * Rx, tfx = ( __builtin_elementwise_fma( qfx, Sx, Bx ) - Ox ) * Rx;
		const v2f tny = ( __builtin_elementwise_fma( qny, Sy, By ) - Oy ) * Ry, tfy = ( __builtin_elementwise_fma( qfy, Sy, By ) - Oy ) * Ry;
		const v2f tnz = ( __builtin_elementwise_fma( qnz, Sz, Bz ) - Oz ) * Rz, tfz = ( __builtin_elementwise_fma( qfz, Sz, Bz ) - Oz ) * Rz;
#pragma unroll
		for (int j = 0; j < 2; j++)
		{
			// entry clamped at 0, exit clamped at the ray's t: a superset of tmax >= tmin && tmin <= t && tmax > 0 (it also
			// lets tmax == 0 and, for ANY, tmin == t through, which these boxes may).  An empty slot has an inverted box and
			// a harmless reference (the builder), so it needs no test of its own
			const float tmin = fmaxf( fmaxf( fmaxf( tnx[j], tny[j] ), tnz[j] ), 0.0f ), tmax = fminf( fminf( fminf( tfx[j], tfy[j] ), tfz[j] ), t );
			// order key: entry distance with the child number in the two lowest bits
			key[c + j] = tmin <= tmax ? ( ( __float_as_uint( tmin ) & ~3u ) | (uint32_t)( c + j ) ) : 0xFFFFFFFFu;
		}
	}
#define RT_CSWAP( a, b ) { const uint32_t lo_ = key[a] < key[b] ? key[a] : key[b], hi_ = key[a] < key[b] ? key[b] : key[a]; key[a] = lo_, key[b] = hi_; }
	RT_CSWAP( 0, 1 ) RT_CSWAP( 2, 3 ) RT_CSWAP( 0, 2 ) RT_CSWAP( 1, 3 ) RT_CSWAP( 1, 2 )
#undef RT_CSWAP
#define RT_PICK( k ) ( ( ( k ) & 2u ) ? ( ( ( k ) & 1u ) ? ref[3] : ref[2] ) : ( ( ( k ) & 1u ) ? ref[1] : ref[0] ) )
	if (key[0] == 0xFFFFFFFFu) return sp > 0 ? stack[--sp * stride] : RT_DONE;
	// the nearest child next, the others on the stack, farthest deepest
	if (key[3] != 0xFFFFFFFFu) { stack[sp * stride] = RT_PICK( key[3] ); sp++; }
	if (key[2] != 0xFFFFFFFFu) { stack[sp * stride] = RT_PICK( key[2] ); sp++; }
	if (key[1] != 0xFFFFFFFFu) { stack[sp * stride] = RT_PICK( key[1] ); sp++; }
	return RT_PICK( key[0] );
#undef RT_PICK
}

// The whole walk over the 4-wide nodes for one lane (k_tail): the caller has done the root box test and knows that rD is
// finite.  A hit counts once the exact box of its leaf passes the reference's test (see above).
template <bool ANY>
__device__ __forceinline__ bool traverse_wide( const DScene& s, f3 O, f3 D, f3 rD, RayHit& h, uint32_t* stack, const int stride )
{
	uint32_t cur = s.wide_root;
	int sp = 0;
	while (true)
	{
		while (!( cur & RT_LEAF )) cur = wide_step<ANY>( s, cur, O, rD, h.t, stack, stride, sp );
		if (cur == RT_DONE) break;
		const int first = (int)( cur & RT_LEAF_FIRST_MASK ), count = (int)( ( cur >> RT_LEAF_FIRST_BITS ) & ( RT_LEAF_MAX_COUNT - 1 ) ) + 1;
		for (int i = 0; i < count; i++)
		{
			const float t0 = h.t;
			const int obj0 = h.obj, pos0 = h.pos;
			if (!isect_prim<ANY>( s, first + i, O, D, h )) continue;
			const float4 b0 = s.leafbox[2 * first], b1 = s.leafbox[2 * first + 1];
			float tmn, tmx;
			slab<false>( b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, O, rD, tmn, tmx );
			if (tmx >= tmn && ( ANY ? tmn < t0 : tmn <= t0 ) && tmx > 0) { if (ANY) return true; }
			else h.t = t0, h.obj = obj0, h.pos = pos0;
		}
		cur = sp > 0 ? stack[--sp * stride] : RT_DONE;
	}
	return false;
}

// ---- the same walk by a TEAM of four lanes (k_tail_team) ----------------------------------------------------------------
// A path that is alone in its wave (the last few thousand paths of a render) pays for every instruction of its walk at the
// issue rate of one wave, about 4 cycles each.  Four neighbouring lanes (a DPP quad) therefore share one ray: every
// lane holds the same ray and the same traversal state, lane j of the team tests child j of a 4-wide node, the four keys
// are ranked with three quad rotations (1 VALU op each, no LDS), and the team's stack lives in one LDS column (plain LDS
// accesses: a lane reads what another lane of its wave wrote earlier in program order, which the in-order LDS queue of a
// wave guarantees; every such write is a store the compiler sees on the path, so no read is carried over it).  Nodes are
// visited in exactly the order wide_step visits them and leaf primitives are judged in leaf order, so a team returns what a
// single lane returns, bit for bit.
#define RT_QUAD_ROT1( x ) __builtin_amdgcn_update_dpp( 0, (int)( x ), 0x39, 0xF, 0xF, false ) /* lane j reads lane j+1 of its quad */
#define RT_QUAD_ROT2( x ) __builtin_amdgcn_update_dpp( 0, (int)( x ), 0x4E, 0xF, 0xF, false )
#define RT_QUAD_ROT3( x ) __builtin_amdgcn_update_dpp( 0, (int)( x ), 0x93, 0xF, 0xF, false )
#define RT_QUAD_BCAST( x, l ) __builtin_amdgcn_update_dpp( 0, (int)( x ), ( l ) * 0x55, 0xF, 0xF, false ) /* every lane reads lane l of its quad */
#define RT_MISS_KEY 0xFFFFFFFFu
template <bool ANY>
__device__ __forceinline__ uint32_t team_wide_step( const DScene& s, uint32_t cur, int j, f3 O, f3 rD, float t, uint32_t* stack, const int stride, int& sp )
{
	const float4 q0 = s.wide[4 * cur], q2 = s.wide[4 * cur + 2], q3 = s.wide[4 * cur + 3];
	const uint32_t ref = ( (const uint32_t*)s.wide )[16 * cur + 4 + j];
	uint32_t ref_pinned = ref;
	asm volatile( "" : "+v"( ref_pinned ) ); // keep the child reference's load with the node's other loads: one round trip per step, not two
	const uint32_t ex = __float_as_uint( q0.w );
	const float sx = __uint_as_float( ( ex & 255u ) << 23 ), sy = __uint_as_float( ( ( ex >> 8 ) & 255u ) << 23 ), sz = __uint_as_float( ( ( ex >> 16 ) & 255u ) << 23 );
	const uint32_t sh = 8u * (uint32_t)j;
	const uint32_t lox = ( __float_as_uint( q2.x ) >> sh ) & 255u, loy = ( __float_as_uint( q2.y ) >> sh ) & 255u, loz = ( __float_as_uint( q2.z ) >> sh ) & 255u;
	const uint32_t hix = ( __float_as_uint( q2.w ) >> sh ) & 255u, hiy = ( __float_as_uint( q3.x ) >> sh ) & 255u, hiz = ( __float_as_uint( q3.y ) >> sh ) & 255u;
	// the same operations as wide_step, child j only: plane = fma(byte, 2^e, origin), then (plane - O) * rD
	const float tnx = ( __builtin_fmaf( (float)( rD.x < 0 ? hix : lox ), sx, q0.x ) - O.x ) * rD.x, tfx = ( __builtin_fmaf( (float)( rD.x < 0 ? lox : hix ), sx, q0.x ) - O.x ) * rD.x;
	const float tny = ( __builtin_fmaf( (float)( rD.y < 0 ? hiy : loy ), sy, q0.y ) - O.y ) * rD.y, tfy = ( __builtin_fmaf( (float)( rD.y < 0 ? loy : hiy ), sy, q0.y ) - O.y ) * rD.y;
	const float tnz = ( __builtin_fmaf( (float)( rD.z < 0 ? hiz : loz ), sz, q0.z ) - O.z ) * rD.z, tfz = ( __builtin_fmaf( (float)( rD.z < 0 ? loz : hiz ), sz, q0.z ) - O.z ) * rD.z;
	const float tmin = fmaxf( fmaxf( fmaxf( tnx, tny ), tnz ), 0.0f ), tmax = fminf( fminf( fminf( tfx, tfy ), tfz ), t );
	const uint32_t key = tmin <= tmax ? ( ( __float_as_uint( tmin ) & ~3u ) | (uint32_t)j ) : RT_MISS_KEY;
	const uint32_t k1 = (uint32_t)RT_QUAD_ROT1( key ), k2 = (uint32_t)RT_QUAD_ROT2( key ), k3 = (uint32_t)RT_QUAD_ROT3( key );
	const bool hit = key != RT_MISS_KEY;
	const int rank = (int)( k1 < key ) + (int)( k2 < key ) + (int)( k3 < key ); // valid keys differ in their two lowest bits
	const int nhit = (int)hit + (int)( k1 != RT_MISS_KEY ) + (int)( k2 != RT_MISS_KEY ) + (int)( k3 != RT_MISS_KEY );
	if (nhit == 0) return sp > 0 ? stack[--sp * stride] : RT_DONE; // (the same for the four lanes)
	// the nearest child next, the others on the stack, farthest deepest (as wide_step pushes them)
	if (hit && rank >= 1) stack[( sp + nhit - 1 - rank ) * stride] = ref_pinned;
	sp += nhit - 1;
	uint32_t next = hit && rank == 0 ? ref_pinned : 0u;
	next |= (uint32_t)RT_QUAD_ROT1( next ), next |= (uint32_t)RT_QUAD_ROT2( next ); // after ROT1 lane j holds j | j+1, after ROT2 all four
	return next;
}
// One leaf primitive without the comparison against the incumbent hit: the distances isect_prim would consider, in its
// order (a sphere has two), and whether each passes its own test (Triangle::Intersect primitive.h:50-72 up to `t > 0.0001f`,
// Sphere::Intersect :108-127 up to `t > 0`).
struct PrimCand { float t1, t2; int obj; bool v1, v2; };
__device__ __forceinline__ void prim_candidates( const DScene& s, int pos, f3 O, f3 D, PrimCand& c )
{
	c.v1 = false, c.v2 = false, c.t1 = 0, c.t2 = 0;
	const float4 a = s.prims[3 * pos], b = s.prims[3 * pos + 1];
	c.obj = __float_as_int( b.w );
	float4 cc4 = s.prims[3 * pos + 2];
	asm volatile( "" : "+v"( cc4.x ) ); // the third quarter of the record travels with the first two: one round trip per primitive, not two
	if (__float_as_int( a.w ) == 0)
	{
		const float4 cc = cc4;
		const f3 p1 = mk3( a ), edge1 = mk3( b ), edge2 = mk3( cc );
		const f3 hh = cross3( D, edge2 );
		const float det = dot3( edge1, hh );
		if (det > -0.0001f && det < 0.0001f) return;
		const float f = 1 / det;
		const f3 sv = O - p1;
		const float u = f * dot3( sv, hh );
		if (u < 0 || u > 1) return;
		const f3 q = cross3( sv, edge1 );
		const float v = f * dot3( D, q );
		if (v < 0 || u + v > 1) return;
		c.t1 = f * dot3( edge2, q );
		c.v1 = c.t1 > 0.0001f;
		return;
	}
	const f3 oc = O - mk3( a );
	const float bb = dot3( oc, D );
	const float cc = dot3( oc, oc ) - b.x;
	float d = bb * bb - cc;
	if (d <= 0) return;
	d = sqrtf( d );
	c.t1 = -bb - d, c.v1 = c.t1 > 0;
	c.t2 = d - bb, c.v2 = c.t2 > 0;
}
// The walk over the 4-wide nodes by a team (see traverse_wide for the single-lane form it reproduces).  Every lane of the
// team passes the same ray, hit and stack column; `j` is the lane's number in its team.  Returns true iff ANY and occluded.
template <bool ANY>
__device__ __forceinline__ bool team_traverse_wide( const DScene& s, int j, f3 O, f3 D, f3 rD, RayHit& h, uint32_t* stack, const int stride, uint32_t& nSteps, uint32_t& nPrims )
{
	uint32_t cur = s.wide_root;
	int sp = 0;
	while (true)
	{
		while (!( cur & RT_LEAF )) { cur = team_wide_step<ANY>( s, cur, j, O, rD, h.t, stack, stride, sp ); nSteps++; }
		if (cur == RT_DONE) break;
		const int first = (int)( cur & RT_LEAF_FIRST_MASK ), count = (int)( ( cur >> RT_LEAF_FIRST_BITS ) & ( RT_LEAF_MAX_COUNT - 1 ) ) + 1;
		// the exact box of the leaf: a hit found under quantised boxes counts only if this passes the reference's test (traverse_wide)
		const float4 b0 = s.leafbox[2 * first], b1 = s.leafbox[2 * first + 1];
		float tmn, tmx;
		slab<false>( b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, O, rD, tmn, tmx );
		const bool boxAlive = tmx >= tmn && tmx > 0;
		for (int base = 0; base < count; base += 4)
		{
			PrimCand c;
			c.v1 = false, c.v2 = false, c.t1 = 0, c.t2 = 0, c.obj = -1;
			if (base + j < count) prim_candidates( s, first + base + j, O, D, c ); // lane j: primitive base + j of the leaf
			nPrims += (uint32_t)( count - base < 4 ? count - base : 4 );
			if (ANY)
			{
				// IntersectBVHShadowRay: any primitive closer than the ray's (fixed) t occludes
				bool occ = boxAlive && tmn < h.t && ( ( c.v1 && c.t1 < h.t ) || ( c.v2 && c.t2 < h.t ) );
				occ = occ | ( RT_QUAD_ROT1( occ ) != 0 ), occ = occ | ( RT_QUAD_ROT2( occ ) != 0 );
				if (occ) return true;
			}
			else
			{
				// judge the up to four results in leaf order, each against the hit as the earlier ones left it: what isect_prim
				// does primitive by primitive (strict <, or equal distance and earlier leaf position)
#define RT_JUDGE( l )                                                                                                                              \
				if (base + ( l ) < count) /* (the same for the four lanes) */                                                                          \
				{                                                                                                                                      \
					const bool v1 = RT_QUAD_BCAST( c.v1, l ) != 0, v2 = RT_QUAD_BCAST( c.v2, l ) != 0;                                                 \
					const float t1 = __int_as_float( RT_QUAD_BCAST( __float_as_int( c.t1 ), l ) ), t2 = __int_as_float( RT_QUAD_BCAST( __float_as_int( c.t2 ), l ) ); \
					const int obj = RT_QUAD_BCAST( c.obj, l ), pos = first + base + ( l );                                                              \
					const bool a1 = v1 && ( t1 < h.t || ( t1 == h.t && pos < h.pos ) );                                                                \
					const bool a2 = !a1 && v2 && ( t2 < h.t || ( t2 == h.t && pos < h.pos ) );                                                         \
					if (( a1 || a2 ) && boxAlive && tmn <= h.t) h.t = a1 ? t1 : t2, h.obj = obj, h.pos = pos;                                           \
				}
				RT_JUDGE( 0 ) RT_JUDGE( 1 ) RT_JUDGE( 2 ) RT_JUDGE( 3 )
#undef RT_JUDGE
			}
		}
		cur = sp > 0 ? stack[--sp * stride] : RT_DONE;
	}
	return false;
}

// ---- one ray per WAVE (k_tail_wave: the last few thousand paths of a render; the parity hook with hook_kernel = 2) --------
// A lone ray's walk is a chain of dependent steps, each a fetch (~400 cycles) plus ~100 instructions at the 4-5 cycles a
// wave alone needs per instruction: about 27 node steps and 5 leaf visits per bounce for a path inside the mirror object,
// 17 us.  Here the 64 lanes of a wave walk ONE ray breadth-first: every iteration takes up to 16 nodes from a stack in LDS
// (lane 4q + j tests child j of node q with exactly wide_step's arithmetic) and up to 64 leaf primitives from a queue (one
// per lane, each with the exact box of its leaf), so the number of dependent steps falls from the number of nodes
// visited to about the depth of the tree.  The closest hit is the smallest (t, leaf position) over the primitives whose
// leaf box passes the reference's test -- the rule the other walkers apply primitive by primitive -- found with wave
// reductions; it does not depend on the order of the visit except in one corner (a hit that lies numerically in front of
// its own leaf box while a competing hit lies between the two: there the result depends on the order in every walker,
// the reference's recursion included).
struct WaveLds
{
	enum { NODE_CAP = 512, LEAF_CAP = 256 };
	uint32_t nref[NODE_CAP]; // node stack: 4-wide node index ...
	uint32_t nkey[NODE_CAP]; // ... and the entry distance it was pushed with (float bits, rounded down)
	uint32_t leaf[LEAF_CAP]; // leaf primitive queue: position | offset within its leaf << 25
};
__device__ __forceinline__ float wave_min_f32( float v ) // the minimum over the 64 lanes, in every lane
{
#define RT_WMIN( ctrl, rm ) v = fminf( v, __int_as_float( __builtin_amdgcn_update_dpp( __float_as_int( v ), __float_as_int( v ), ctrl, rm, 0xF, false ) ) );
	RT_WMIN( 0xB1, 0xF ) RT_WMIN( 0x4E, 0xF ) RT_WMIN( 0x124, 0xF ) RT_WMIN( 0x128, 0xF ) RT_WMIN( 0x142, 0xA ) RT_WMIN( 0x143, 0xC )
#undef RT_WMIN
	return __int_as_float( __builtin_amdgcn_readlane( __float_as_int( v ), 63 ) );
}
__device__ __forceinline__ uint32_t wave_min_u32( uint32_t v )
{
#define RT_WMIN( ctrl, rm ) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp( (int)v, (int)v, ctrl, rm, 0xF, false ); v = o < v ? o : v; }
	RT_WMIN( 0xB1, 0xF ) RT_WMIN( 0x4E, 0xF ) RT_WMIN( 0x124, 0xF ) RT_WMIN( 0x128, 0xF ) RT_WMIN( 0x142, 0xA ) RT_WMIN( 0x143, 0xC )
#undef RT_WMIN
	return (uint32_t)__builtin_amdgcn_readlane( (int)v, 63 );
}
__device__ __forceinline__ uint32_t lane_rank( unsigned long long m ) { return __builtin_amdgcn_mbcnt_hi( (uint32_t)( m >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)m, 0 ) ); }

// The analytic part of Scene::FindNearest (isect_analytic: quads[0], cubes[0], the planes in order, each accepted when
// closer than what came before) with one candidate per lane: lane 0 the quad, lane 1 the cube, lanes 2.. the planes.
// "Closer than what came before" in that order is the smallest distance, the earliest candidate among equals.
__device__ __forceinline__ void wave_analytic( const DScene& s, int lane, f3 O, f3 D, float t0, RayHit& h )
{
	float best = INFINITY;
	int bestObj = -1;
	for (int base = 0; base < 2 + s.n_planes; base += 64) // (one round unless a scene has more than 62 planes)
	{
		RayHit c; c.t = t0, c.obj = -1, c.pos = -1;
		const int k = base + lane;
		if (k == 0) { if (s.n_quads > 0) isect_quad( s, 0, O, D, c ); }
		else if (k == 1) { if (s.n_cubes > 0) isect_cube( s, 0, O, D, c ); }
		else if (k - 2 < s.n_planes) isect_plane( s, k - 2, O, D, c );
		const float tc = c.obj != -1 ? c.t : INFINITY;
		const float tm = wave_min_f32( tc );
		if (tm < best) // (the same in every lane) strictly closer than the earlier rounds' best
		{
			const unsigned long long m = __ballot( tc == tm );
			best = tm, bestObj = __builtin_amdgcn_readlane( c.obj, __ffsll( (long long)m ) - 1 );
		}
	}
	h.t = bestObj != -1 ? best : t0, h.obj = bestObj, h.pos = -1;
}
// Scene::IsOccluded before the BVH: any of them closer than the ray's end occludes (begin_any without the order)
__device__ __forceinline__ bool wave_analytic_any( const DScene& s, int lane, f3 O, f3 D, float tmaxRay )
{
	bool occ = false;
	for (int base = 0; base < 2 + s.n_planes && !occ; base += 64)
	{
		RayHit c; c.t = tmaxRay, c.obj = -1, c.pos = -1;
		const int k = base + lane;
		if (k == 0) { if (s.n_quads > 0) isect_quad( s, 0, O, D, c ); }
		else if (k == 1) { if (s.n_cubes > 0) isect_cube( s, 0, O, D, c ); }
		else if (k - 2 < s.n_planes) isect_plane( s, k - 2, O, D, c );
		occ = __ballot( c.obj != -1 ) != 0;
	}
	return occ;
}

// The walk.  Every lane passes the same ray and hit; returns true iff ANY and occluded.  `overflow`: the node stack or the
// leaf queue would not hold what this ray pushes -- the caller walks the ray another way (result untouched: h restored).
template <bool ANY>
__device__ __forceinline__ bool wave_traverse_wide( const DScene& s, int lane, f3 O, f3 D, f3 rD, RayHit& h, WaveLds& L, bool& overflow, uint32_t& nIter )
{
	const RayHit h0 = h;
	overflow = false;
	const int j = lane & 3, q = lane >> 2;
	if (lane == 0) L.nref[0] = s.wide_root, L.nkey[0] = 0u;
	int nN = 1, nL = 0; // entries on the node stack / in the leaf queue (the same in every lane)
	while (nN > 0 || nL > 0)
	{
		nIter++;
		// ---- issue this iteration's fetches together: up to 16 nodes from the top of the stack, up to 64 leaf primitives
		const int takeN = nN < 16 ? nN : 16, takeL = nL < 64 ? nL : 64;
		uint32_t node = 0, nodeKey = 0;
		float4 q0 = make_float4( 0, 0, 0, 0 ), q2 = q0, q3 = q0;
		uint32_t ref = RT_LEAF;
		const bool haveNode = q < takeN;
		if (haveNode)
		{
			node = L.nref[nN - 1 - q], nodeKey = L.nkey[nN - 1 - q];
			q0 = s.wide[4 * node], q2 = s.wide[4 * node + 2], q3 = s.wide[4 * node + 3];
			ref = ( (const uint32_t*)s.wide )[16 * node + 4 + j];
		}
		const bool haveLeaf = lane < takeL;
		PrimCand c;
		c.v1 = false, c.v2 = false, c.t1 = 0, c.t2 = 0, c.obj = -1;
		int pos = 0;
		float tmn = 0, tmx = -1;
		if (haveLeaf)
		{
			const uint32_t e = L.leaf[nL - 1 - lane];
			pos = (int)( e & RT_LEAF_FIRST_MASK );
			const int first = pos - (int)( e >> RT_LEAF_FIRST_BITS );
			const float4 b0 = s.leafbox[2 * first], b1 = s.leafbox[2 * first + 1];
			prim_candidates( s, pos, O, D, c );
			slab<false>( b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, O, rD, tmn, tmx );
		}
		nN -= takeN, nL -= takeL;
		// ---- leaf primitives first: a closer hit prunes the children tested below
		if (takeL > 0) // (the same in every lane)
		{
			const bool boxAlive = tmx >= tmn && tmx > 0;
			if (ANY)
			{
				const bool occ = haveLeaf && boxAlive && tmn < h.t && ( ( c.v1 && c.t1 < h.t ) || ( c.v2 && c.t2 < h.t ) );
				if (__ballot( occ ) != 0) return true;
			}
			else
			{
				// this lane's candidate as isect_prim would judge it against the hit so far, confirmed by its leaf's exact box
				const bool a1 = c.v1 && ( c.t1 < h.t || ( c.t1 == h.t && pos < h.pos ) );
				const bool a2 = !a1 && c.v2 && ( c.t2 < h.t || ( c.t2 == h.t && pos < h.pos ) );
				const bool ok = haveLeaf && ( a1 || a2 ) && boxAlive && tmn <= h.t;
				const float tc = ok ? ( a1 ? c.t1 : c.t2 ) : INFINITY;
				const float tm = wave_min_f32( tc );
				if (tm < INFINITY) // the smallest (t, position) of the batch
				{
					const uint32_t pm = wave_min_u32( tc == tm ? (uint32_t)pos : 0xFFFFFFFFu );
					const unsigned long long m = __ballot( tc == tm && (uint32_t)pos == pm );
					h.t = tm, h.pos = (int)pm, h.obj = __builtin_amdgcn_readlane( c.obj, __ffsll( (long long)m ) - 1 );
				}
			}
		}
		// ---- children of the nodes (wide_step's arithmetic, child j of node q in lane 4q + j)
		if (takeN > 0)
		{
			uint32_t key = RT_MISS_KEY;
			// a node pushed before the hit got closer than its entry distance needs no test (its key was rounded down: never wrong)
			if (haveNode && ( nodeKey & ~3u ) <= __float_as_uint( h.t ))
			{
				const uint32_t ex = __float_as_uint( q0.w );
				const float sx = __uint_as_float( ( ex & 255u ) << 23 ), sy = __uint_as_float( ( ( ex >> 8 ) & 255u ) << 23 ), sz = __uint_as_float( ( ( ex >> 16 ) & 255u ) << 23 );
				const uint32_t sh = 8u * (uint32_t)j;
				const uint32_t lox = ( __float_as_uint( q2.x ) >> sh ) & 255u, loy = ( __float_as_uint( q2.y ) >> sh ) & 255u, loz = ( __float_as_uint( q2.z ) >> sh ) & 255u;
				const uint32_t hix = ( __float_as_uint( q2.w ) >> sh ) & 255u, hiy = ( __float_as_uint( q3.x ) >> sh ) & 255u, hiz = ( __float_as_uint( q3.y ) >> sh ) & 255u;
				const float tnx = ( __builtin_fmaf( (float)( rD.x < 0 ? hix : lox ), sx, q0.x ) - O.x ) * rD.x, tfx = ( __builtin_fmaf( (float)( rD.x < 0 ? lox : hix ), sx, q0.x ) - O.x ) * rD.x;
				const float tny = ( __builtin_fmaf( (float)( rD.y < 0 ? hiy : loy ), sy, q0.y ) - O.y ) * rD.y, tfy = ( __builtin_fmaf( (float)( rD.y < 0 ? loy : hiy ), sy, q0.y ) - O.y ) * rD.y;
				const float tnz = ( __builtin_fmaf( (float)( rD.z < 0 ? hiz : loz ), sz, q0.z ) - O.z ) * rD.z, tfz = ( __builtin_fmaf( (float)( rD.z < 0 ? loz : hiz ), sz, q0.z ) - O.z ) * rD.z;
				const float tmin = fmaxf( fmaxf( fmaxf( tnx, tny ), tnz ), 0.0f ), tmax = fminf( fminf( fminf( tfx, tfy ), tfz ), h.t );
				if (tmin <= tmax) key = ( __float_as_uint( tmin ) & ~3u ) | (uint32_t)j;
			}
			const bool hit = key != RT_MISS_KEY;
			const uint32_t k1 = (uint32_t)RT_QUAD_ROT1( key ), k2 = (uint32_t)RT_QUAD_ROT2( key ), k3 = (uint32_t)RT_QUAD_ROT3( key );
			const int rank = (int)( k1 < key ) + (int)( k2 < key ) + (int)( k3 < key ); // among the children of its own node
			const bool toLeaf = hit && ( ref & RT_LEAF ) != 0, toNode = hit && !toLeaf;
			// interior children: onto the stack, every node's farthest first and nearest last, so that the next iteration
			// takes the nearest child of each of these nodes from the top
			const unsigned long long mAll = __ballot( toNode );
			if (nN + __popcll( mAll ) > WaveLds::NODE_CAP) { overflow = true; h = h0; return false; }
			if (nN + __popcll( mAll ) <= 16) // the next iteration takes them all: their order on the stack does not matter
			{
				if (toNode) { const int at = nN + (int)lane_rank( mAll ); L.nref[at] = ref, L.nkey[at] = key; }
				nN += __popcll( mAll );
			}
			else
			{
#pragma unroll
				for (int r = 3; r >= 0; r--)
				{
					const bool mine = toNode && rank == r;
					const unsigned long long m = __ballot( mine );
					if (mine) { const int at = nN + (int)lane_rank( m ); L.nref[at] = ref, L.nkey[at] = key; }
					nN += __popcll( m );
				}
			}
			// leaf children: every primitive of the leaf becomes one entry of the queue
			const int cnt = toLeaf ? (int)( ( ref >> RT_LEAF_FIRST_BITS ) & ( RT_LEAF_MAX_COUNT - 1 ) ) + 1 : 0;
			const int first = (int)( ref & RT_LEAF_FIRST_MASK );
			for (int r = 0; __ballot( r < cnt ) != 0; r++)
			{
				const bool mine = r < cnt;
				const unsigned long long m = __ballot( mine );
				if (nL + __popcll( m ) > WaveLds::LEAF_CAP) { overflow = true; h = h0; return false; }
				if (mine) L.leaf[nL + (int)lane_rank( m )] = (uint32_t)( first + r ) | ( (uint32_t)r << RT_LEAF_FIRST_BITS );
				nL += __popcll( m );
			}
		}
	}
	return false;
}

// Scene::FindNearest (scene.h:359-368): quads[0], cubes[0], planes, BVH.  h.t carries the ray's tmax in.
template <bool COUNT, bool QBVH>
__device__ __forceinline__ void find_nearest( const DScene& s, f3 O, f3 D, RayHit& h, uint32_t* stack, int stride, TravCount& cnt )
{
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z ); // Ray ctor, primitive.h:11
	isect_analytic( s, O, D, h );
	if (s.n_bvh_prims == 0) return;
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (QBVH) { if (inexact) traverse_qbvh<false>( s, O, D, rD, h, stack, stride ); else traverse<false, false, COUNT>( s, O, D, rD, h, stack, stride, cnt ); }
	else if (__any( inexact )) traverse<false, true, COUNT>( s, O, D, rD, h, stack, stride, cnt );
	else traverse<false, false, COUNT>( s, O, D, rD, h, stack, stride, cnt );
}
// Scene::IsOccluded (scene.h:369-379): quad, cube with the shrinking ray.t, then planes and BVH with early out.
template <bool COUNT, bool QBVH>
__device__ __forceinline__ bool is_occluded( const DScene& s, f3 O, f3 D, float tmaxRay, uint32_t* stack, int stride, TravCount& cnt )
{
	RayHit h; h.t = tmaxRay, h.obj = -1, h.pos = -1;
	if (s.n_quads > 0) isect_quad( s, 0, O, D, h );
	if (s.n_cubes > 0) isect_cube( s, 0, O, D, h );
	if (h.obj != -1) return true;
	isect_planes( s, O, D, h );
	if (h.obj != -1) return true;
	if (s.n_bvh_prims == 0) return false;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (QBVH) return inexact ? traverse_qbvh<true>( s, O, D, rD, h, stack, stride ) : traverse<true, false, COUNT>( s, O, D, rD, h, stack, stride, cnt );
	if (__any( inexact )) return traverse<true, true, COUNT>( s, O, D, rD, h, stack, stride, cnt );
	return traverse<true, false, COUNT>( s, O, D, rD, h, stack, stride, cnt );
}

// ---- hit attributes (Scene::GetNormal / GetAlbedo / GetMaterial, scene.h:380-478) ---------------------------
// Cube::GetNormal, primitive.h:237-254
__device__ __forceinline__ f3 cube_normal( const DScene& s, int i, f3 I )
{
	m34 invM; invM.r0 = s.cubes[8 * i], invM.r1 = s.cubes[8 * i + 1], invM.r2 = s.cubes[8 * i + 2];
	m34 M; M.r0 = s.cubes[8 * i + 5], M.r1 = s.cubes[8 * i + 6], M.r2 = s.cubes[8 * i + 7];
	const float4 b0 = s.cubes[8 * i + 3], b1 = s.cubes[8 * i + 4];
	const f3 o = xform_pos( I, invM );
	f3 N = mk3( -1, 0, 0 );
	const float d0 = fabsf( o.x - b0.x ), d1 = fabsf( o.x - b1.x );
	const float d2 = fabsf( o.y - b0.y ), d3 = fabsf( o.y - b1.y );
	const float d4 = fabsf( o.z - b0.z ), d5 = fabsf( o.z - b1.z );
	float minDist = d0;
	if (d1 < minDist) minDist = d1, N.x = 1;
	if (d2 < minDist) minDist = d2, N = mk3( 0, -1, 0 );
	if (d3 < minDist) minDist = d3, N = mk3( 0, 1, 0 );
	if (d4 < minDist) minDist = d4, N = mk3( 0, 0, -1 );
	if (d5 < minDist) minDist = d5, N = mk3( 0, 0, 1 );
	return xform_vec( N, M );
}
// Plane::GetAlbedo, primitive.h:163-186
__device__ __forceinline__ f3 plane_albedo( const DScene& s, f3 N, f3 I )
{
	if (N.y == 1)
	{
		int ix = (int)( I.x * 2 + 96.01f );
		int iz = (int)( I.z * 2 + 96.01f );
		if (ix == 98 && iz == 98) ix = (int)( I.x * 32.01f ), iz = (int)( I.z * 32.01f );
		if (ix == 94 && iz == 98) ix = (int)( I.x * 64.01f ), iz = (int)( I.z * 64.01f );
		const float c = ( ( ix + iz ) & 1 ) ? 1.0f : 0.3f;
		return mk3( c, c, c );
	}
	else if (N.z == -1)
	{
		const int ix = (int)( ( I.x + 4 ) * ( 128.0f / 8 ) );
		const int iy = (int)( ( 2 - I.y ) * ( 64.0f / 3 ) );
		const uint32_t p = s.texture ? s.texture[( ix & 127 ) + ( iy & 63 ) * 128] : 0u;
		return mk3( (float)( ( p >> 16 ) & 255 ), (float)( ( p >> 8 ) & 255 ), (float)( p & 255 ) ) * ( 1.0f / 255.0f );
	}
	return mk3( 0.93f, 0.93f, 0.93f );
}
struct HitAttr { f3 N; f3 albedo; const DMaterial* mat; };
// wo = ray.D; WANT_ALBEDO mirrors where the reference calls GetAlbedo (renderer.cpp:145, 274, 324)
__device__ __forceinline__ void hit_attributes( const DScene& s, int obj, f3 I, f3 wo, HitAttr& a )
{
	const float4 pr = s.primitives[obj];
	a.mat = &s.materials[__float_as_uint( pr.w )];
	const f3 color = mk3( a.mat->color );
	int k = obj;
	f3 N;
	if (k < s.n_quads) { N = mk3( pr ); a.albedo = mk3( 10, 10, 10 ); }                    // primitive.h:296-304
	else if (( k -= s.n_quads ) < s.n_spheres) { N = ( I - mk3( pr ) ) * s.sphere_invr[k]; a.albedo = color; } // :128-131
	else if (( k -= s.n_spheres ) < s.n_cubes) { N = cube_normal( s, k, I ); a.albedo = color; }
	else if (( k -= s.n_cubes ) < s.n_planes) { N = mk3( pr ); a.albedo = plane_albedo( s, N, I ); }
	else { N = mk3( pr ); a.albedo = color; }
	if (dot3( N, wo ) > 0) N = -N; // scene.h:412
	a.N = N;
}
