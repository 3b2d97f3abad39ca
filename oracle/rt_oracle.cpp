// oracle/rt_oracle.cpp -- TEST INFRASTRUCTURE: CPU restatement of the reference's per-pixel trace loop.
//
// This is the parity checker, not a product path.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load liboracle.so; the product (raytracer_amd/) never does and has no CPU fallback.
//
// PARITY UNPINNED in the sense of this project's evidence rule: the reference holds no tests, fixtures or golden vectors of
// its own, and its sources do not compile here as they are (Win32/OpenCL/GLFW precomp.h).  What this file IS
// checked against: the reference's own primitive.h / bvh.h / template/scene.h / camera.h / renderer.cpp /
// postprocessing.h compiled in round 1 behind a stand-in precomp.h (oracle/ref_build/, dev container only,
// tests/test_oracle_vs_ref.py) and the golden vectors generated from that build (tests/golden/*.npz, tools/make_golden.py,
// tests/test_oracle_golden.py) -- bit for bit, but a build behind stand-ins does not count as the reference.  The camera
// script at the end of this file (round 2) has no such cross-check at all: it restates camera.h:95-159 by reading.
// Like the reference (Renderer::Sample has no bounce limit, renderer.cpp:251) orc_sample does not return from a path that never
// ends: sample 220 of the default scene at 1280x720 holds one.  The product ends such a path (rtmi option last_round_bounces).
//
// Plain C++ with scalar binary32 arithmetic in the reference's operation order; build with
// g++ -O2 -ffp-contract=off -fno-fast-math (no FMA, no reassociation).  Input is the same flat scene description
// the HIP library takes (include/rtmi.h), so both sides see identical bytes.
// Each function cites the reference lines it restates (paths relative to /root/reference/Raytracer).
#include "../include/rtmi.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

typedef unsigned int uint;
struct V3 { float x, y, z; };
inline V3 v3( float x, float y, float z ) { return V3{ x, y, z }; }
inline V3 v3( const float* p ) { return V3{ p[0], p[1], p[2] }; }
inline V3 operator+( V3 a, V3 b ) { return v3( a.x + b.x, a.y + b.y, a.z + b.z ); }
inline V3 operator-( V3 a, V3 b ) { return v3( a.x - b.x, a.y - b.y, a.z - b.z ); }
inline V3 operator-( V3 a ) { return v3( -a.x, -a.y, -a.z ); }
inline V3 operator*( V3 a, V3 b ) { return v3( a.x * b.x, a.y * b.y, a.z * b.z ); }
inline V3 operator*( V3 a, float b ) { return v3( a.x * b, a.y * b, a.z * b ); }
inline V3 operator*( float b, V3 a ) { return v3( b * a.x, b * a.y, b * a.z ); }
inline V3 operator/( V3 a, float b ) { return v3( a.x / b, a.y / b, a.z / b ); }
inline float dot( V3 a, V3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }                               // precomp.h:822
inline V3 cross( V3 a, V3 b ) { return v3( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x ); } // precomp.h:875
inline V3 normalize( V3 v ) { float inv = 1.0f / sqrtf( dot( v, v ) ); return v * inv; }                   // precomp.h:490, 844
inline float magnitude( V3 v ) { return sqrtf( v.x * v.x + v.y * v.y + v.z * v.z ); }                       // precomp.h:848
inline float sqrf( float x ) { return x * x; }
inline float smin( float a, float b ) { return ( b < a ) ? b : a; } // std::min
inline float smax( float a, float b ) { return ( a < b ) ? b : a; } // std::max
const float PI = 3.14159265358979323846264f;                        // common.h:8

// (a, w) * M, rows 0..2 (template.cpp:1021-1035)
inline V3 xpos( V3 a, const float* M )
{
	return v3( M[0] * a.x + M[1] * a.y + M[2] * a.z + M[3] * 1.0f, M[4] * a.x + M[5] * a.y + M[6] * a.z + M[7] * 1.0f, M[8] * a.x + M[9] * a.y + M[10] * a.z + M[11] * 1.0f );
}
inline V3 xvec( V3 a, const float* M )
{
	return v3( M[0] * a.x + M[1] * a.y + M[2] * a.z + M[3] * 0.0f, M[4] * a.x + M[5] * a.y + M[6] * a.z + M[7] * 0.0f, M[8] * a.x + M[9] * a.y + M[10] * a.z + M[11] * 0.0f );
}

// template.cpp:869-891
inline uint rng_u( uint& s ) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
inline float rng_f( uint& s ) { return rng_u( s ) * 2.3283064365387e-10f; }
inline uint wang( uint s ) { s = ( s ^ 61 ) ^ ( s >> 16 ); s *= 9, s = s ^ ( s >> 4 ); s *= 0x27d4eb2d; s = s ^ ( s >> 15 ); return s; } // :871-878
inline uint sample_seed( uint s0, uint s ) { uint st = wang( s0 + s * 0x9E3779B9u ); return st ? st : 1u; }

struct Ray // primitive.h:3-28
{
	V3 O, D, rD;
	float t = 1e34f;
	int objIdx = -1;
	bool inside = false;
	Ray() {}
	Ray( V3 o, V3 d, float dist = 1e34f ) : O( o ), D( d ), t( dist ) { rD = v3( 1 / D.x, 1 / D.y, 1 / D.z ); }
};

struct Counters { unsigned long long extend = 0, shadow = 0, nodes = 0, prims = 0, draws = 0, snodes = 0, sprims = 0; };

struct Oracle
{
	rtmi_scene_desc d;
	std::vector<float> quad_inv_T, quad_size, sphere_info, sphere_invr, cube_M, cube_inv_M, cube_b, tri_verts, primitives, albedo, ior, absorption;
	std::vector<int> material_type, prim_idx;
	std::vector<rtmi_bvh_node> nodes;
	std::vector<uint32_t> texture;
	int total = 0, sphereBase = 0, cubeBase = 0, planeBase = 0, triBase = 0;

	// ---- primitive.h ----
	void quad( int i, Ray& r ) const // Quad::Intersect, primitive.h:284-295
	{
		const float* M = &quad_inv_T[16 * i];
		const float size = quad_size[i];
		const V3 O = xpos( r.O, M ), D = xvec( r.D, M );
		const float t = O.y / -D.y;
		if (t < r.t && t > 0)
		{
			V3 I = O + t * D;
			if (I.x > -size && I.x < size && I.z > -size && I.z < size) r.t = t, r.objIdx = i;
		}
	}
	void cube( int i, Ray& r ) const // Cube::Intersect, primitive.h:210-236
	{
		const float* M = &cube_inv_M[16 * i];
		const float* b0 = &cube_b[8 * i], *b1 = b0 + 4;
		const V3 O = xpos( r.O, M ), D = xvec( r.D, M );
		float rDx = 1 / D.x, rDy = 1 / D.y, rDz = 1 / D.z;
		int signx = D.x < 0, signy = D.y < 0, signz = D.z < 0;
		const float* b[2] = { b0, b1 };
		float tmin = ( b[signx][0] - O.x ) * rDx;
		float tmax = ( b[1 - signx][0] - O.x ) * rDx;
		float tymin = ( b[signy][1] - O.y ) * rDy;
		float tymax = ( b[1 - signy][1] - O.y ) * rDy;
		if (tmin > tymax || tymin > tmax) return;
		tmin = smax( tmin, tymin ), tmax = smin( tmax, tymax );
		float tzmin = ( b[signz][2] - O.z ) * rDz;
		float tzmax = ( b[1 - signz][2] - O.z ) * rDz;
		if (tmin > tzmax || tzmin > tmax) return;
		tmin = smax( tmin, tzmin ), tmax = smin( tmax, tzmax );
		if (tmin > 0) { if (tmin < r.t) r.t = tmin, r.objIdx = cubeBase + i; }
		else if (tmax > 0) { if (tmax < r.t) r.t = tmax, r.objIdx = cubeBase + i; }
	}
	void plane( int i, Ray& r ) const // Plane::Intersect, primitive.h:154-158
	{
		const float* p = &primitives[4 * ( planeBase + i )];
		const V3 N = v3( p );
		float t = -( dot( r.O, N ) + p[3] ) / ( dot( r.D, N ) );
		if (t < r.t && t > 0) r.t = t, r.objIdx = planeBase + i;
	}
	void sphere( int i, Ray& r ) const // Sphere::Intersect, primitive.h:108-127
	{
		const float* s = &sphere_info[4 * i];
		V3 oc = r.O - v3( s );
		float b = dot( oc, r.D );
		float c = dot( oc, oc ) - s[3];
		float t, d = b * b - c;
		if (d <= 0) return;
		d = sqrtf( d ), t = -b - d;
		if (t < r.t && t > 0) { r.t = t, r.objIdx = sphereBase + i; return; }
		t = d - b;
		if (t < r.t && t > 0) { r.t = t, r.objIdx = sphereBase + i; return; }
	}
	void triangle( int i, Ray& r ) const // Triangle::Intersect, primitive.h:50-72
	{
		const float* v = &tri_verts[12 * (size_t)i];
		const V3 pos1 = v3( v ), pos2 = v3( v + 4 ), pos3 = v3( v + 8 );
		const V3 edge1 = pos2 - pos1;
		const V3 edge2 = pos3 - pos1;
		const V3 h = cross( r.D, edge2 );
		const float a = dot( edge1, h );
		if (a > -0.0001f && a < 0.0001f) return;
		const float f = 1 / a;
		const V3 s = r.O - pos1;
		const float u = f * dot( s, h );
		if (u < 0 || u > 1) return;
		const V3 q = cross( s, edge1 );
		const float vv = f * dot( r.D, q );
		if (vv < 0 || u + vv > 1) return;
		const float t = f * dot( edge2, q );
		if (t > 0.0001f && t < r.t) r.t = t, r.objIdx = triBase + i;
	}
	inline void bvh_prim( int k, Ray& r ) const // bvh.h:515-522
	{
		const int p = prim_idx[k];
		if (p < d.n_spheres) sphere( p, r ); else triangle( p - d.n_spheres, r );
	}

	// ---- bvh.h ----
	static bool aabb( const Ray& ray, const rtmi_bvh_node& n ) // BVH::IntersectAABB, bvh.h:443-452
	{
		float tx1 = ( n.aabbMinx - ray.O.x ) * ray.rD.x, tx2 = ( n.aabbMaxx - ray.O.x ) * ray.rD.x;
		float tmin = smin( tx1, tx2 ), tmax = smax( tx1, tx2 );
		float ty1 = ( n.aabbMiny - ray.O.y ) * ray.rD.y, ty2 = ( n.aabbMaxy - ray.O.y ) * ray.rD.y;
		tmin = smax( tmin, smin( ty1, ty2 ) ), tmax = smin( tmax, smax( ty1, ty2 ) );
		float tz1 = ( n.aabbMinz - ray.O.z ) * ray.rD.z, tz2 = ( n.aabbMaxz - ray.O.z ) * ray.rD.z;
		tmin = smax( tmin, smin( tz1, tz2 ) ), tmax = smin( tmax, smax( tz1, tz2 ) );
		return tmax >= tmin && tmin < ray.t && tmax > 0;
	}
	static float aabb_stack( const Ray& ray, const rtmi_bvh_node& n ) // BVH::IntersectAABBStack, bvh.h:454-463
	{
		float tx1 = ( n.aabbMinx - ray.O.x ) * ray.rD.x, tx2 = ( n.aabbMaxx - ray.O.x ) * ray.rD.x;
		float tmin = smin( tx1, tx2 ), tmax = smax( tx1, tx2 );
		float ty1 = ( n.aabbMiny - ray.O.y ) * ray.rD.y, ty2 = ( n.aabbMaxy - ray.O.y ) * ray.rD.y;
		tmin = smax( tmin, smin( ty1, ty2 ) ), tmax = smin( tmax, smax( ty1, ty2 ) );
		float tz1 = ( n.aabbMinz - ray.O.z ) * ray.rD.z, tz2 = ( n.aabbMaxz - ray.O.z ) * ray.rD.z;
		tmin = smax( tmin, smin( tz1, tz2 ) ), tmax = smin( tmax, smax( tz1, tz2 ) );
		if (tmax >= tmin && tmin < ray.t && tmax > 0) return tmin; else return 1e30f;
	}
	void bvh_rec( Ray& ray, uint nodeIdx, Counters* c ) const // BVH::IntersectBVH(ray, nodeIdx), bvh.h:507-530
	{
		const rtmi_bvh_node& node = nodes[nodeIdx];
		if (c) c->nodes++;
		if (!aabb( ray, node )) return;
		if (node.primitiveCount > 0)
		{
			for (int i = 0; i < node.primitiveCount; i++) { if (c) c->prims++; bvh_prim( node.leftFirst + i, ray ); }
		}
		else bvh_rec( ray, node.leftFirst, c ), bvh_rec( ray, node.leftFirst + 1, c );
	}
	void intersect_bvh( Ray& ray, Counters* c ) const // BVH::IntersectBVH(ray), bvh.h:496-505
	{
		for (int i = 0; i < d.n_planes; i++) plane( i, ray );
		if (d.n_spheres + d.n_triangles > 0) bvh_rec( ray, 2, c );
	}
	void intersect_bvh_stack( Ray& ray ) const // BVH::IntersectBVHStack, bvh.h:532-578
	{
		for (int i = 0; i < d.n_planes; i++) plane( i, ray );
		if (d.n_spheres + d.n_triangles == 0) return;
		const rtmi_bvh_node* node = &nodes[2], *stack[64];
		uint sp = 0;
		while (1)
		{
			if (node->primitiveCount > 0)
			{
				for (int i = 0; i < node->primitiveCount; i++) bvh_prim( node->leftFirst + i, ray );
				if (sp == 0) break; else node = stack[--sp];
				continue;
			}
			const rtmi_bvh_node* c1 = &nodes[node->leftFirst], *c2 = &nodes[node->leftFirst + 1];
			float d1 = aabb_stack( ray, *c1 ), d2 = aabb_stack( ray, *c2 );
			if (d1 > d2) std::swap( d1, d2 ), std::swap( c1, c2 );
			if (d1 == 1e30f) { if (sp == 0) break; else node = stack[--sp]; }
			else { node = c1; if (d2 != 1e30f) stack[sp++] = c2; }
		}
	}
	bool shadow_rec( Ray& ray, bool occluded, uint nodeIdx, Counters* c ) const // bvh.h:656-688
	{
		if (occluded) return occluded;
		const rtmi_bvh_node& node = nodes[nodeIdx];
		if (c) c->snodes++;
		if (!aabb( ray, node )) return occluded;
		if (node.primitiveCount > 0)
		{
			for (int i = 0; i < node.primitiveCount; i++)
			{
				if (c) c->sprims++;
				bvh_prim( node.leftFirst + i, ray );
				if (ray.objIdx != -1) return true;
			}
			return false;
		}
		bool l = shadow_rec( ray, occluded, node.leftFirst, c );
		return shadow_rec( ray, l, node.leftFirst + 1, c );
	}
	bool shadow_bvh( Ray& ray, bool occluded, Counters* c ) const // BVH::IntersectBVHShadowRay, bvh.h:640-654
	{
		if (occluded) return occluded;
		for (int i = 0; i < d.n_planes; i++) { plane( i, ray ); if (ray.objIdx != -1) return true; }
		if (d.n_spheres + d.n_triangles == 0) return false;
		return shadow_rec( ray, occluded, 2, c );
	}

	// ---- the 4-wide QBVH, the reference's default traversal (Scene::useQBVH = true, scene.h:585) -----------------------------------
	// SIMD_BVH_Node (bvh.h:18-27) with scalars in place of the __m128 unions.  Slots that the reference leaves uninitialised (boxes
	// and counts of unused children: `new SIMD_BVH_Node[...]`, bvh.h:215) are zero here; the traversal never reads them (it leaves
	// the node at the first child == -1, bvh.h:611, 727).
	struct QNode { float bminx[4], bmaxx[4], bminy[4], bmaxy[4], bminz[4], bmaxz[4]; int child[4], count[4]; };
	std::vector<QNode> qn;
	bool use_qbvh = false; // Scene::SetQBVH (scene.h:553-556); the contract of this repository is the binary traversal, so off by default
	void q_set( QNode& q, int slot, const rtmi_bvh_node& n )
	{
		q.bminx[slot] = n.aabbMinx, q.bminy[slot] = n.aabbMiny, q.bminz[slot] = n.aabbMinz;
		q.bmaxx[slot] = n.aabbMaxx, q.bmaxy[slot] = n.aabbMaxy, q.bmaxz[slot] = n.aabbMaxz;
		q.count[slot] = n.primitiveCount, q.child[slot] = n.leftFirst;
	}
	void q_assign( int parentIdx, int childrenIdx ) // BVH::AssignNodeQBVH, bvh.h:177-211
	{
		QNode& p = qn[parentIdx];
		q_set( p, 0, nodes[childrenIdx] ), q_set( p, 1, nodes[childrenIdx + 1] );
		p.child[2] = -1, p.child[3] = -1;
	}
	void q_collapse( int nodeIdx ) // BVH::CollapseBVH, bvh.h:75-174
	{
		for (int i = 0; i < 4; i++)
		{
			QNode& p = qn[nodeIdx];
			if (p.child[i] != -1) continue;
			int greatest_child = -1;
			float greatest_area = -1.0f;
			for (int j = 0; j < i; j++) if (p.count[j] == 0)
			{
				const float areaX = fabsf( p.bmaxx[j] - p.bminx[j] ), areaY = fabsf( p.bmaxy[j] - p.bminy[j] ), areaZ = fabsf( p.bmaxz[j] - p.bminz[j] );
				const float area = areaX * areaY + areaY * areaZ + areaZ * areaX;
				if (area > greatest_area) greatest_area = area, greatest_child = j;
			}
			if (greatest_child == -1) break;
			const int left_child = p.child[greatest_child], right_child = left_child + 1;
			q_set( p, greatest_child, nodes[left_child] ); // the left child takes its parent's slot,
			q_set( p, i, nodes[right_child] );             // the right child the first free one
		}
		for (int i = 0; i < 4; i++)
		{
			if (qn[nodeIdx].child[i] != -1 && qn[nodeIdx].count[i] == 0)
			{
				const int newNode = (int)qn.size();
				qn.push_back( QNode{} ); // (the reference indexes a preallocated array: same numbering)
				const int childrenIdx = qn[nodeIdx].child[i];
				qn[nodeIdx].child[i] = newNode;
				q_assign( newNode, childrenIdx );
				q_collapse( newNode );
			}
		}
	}
	// BVH::BuildQBVH, bvh.h:213-249.  Only for an interior primitive root: with a leaf root the reference collapses node 0 itself,
	// whose plane slot it then takes for an interior node when the scene has no planes: such scenes keep the binary walk.
	bool build_qbvh()
	{
		qn.clear();
		if (d.n_spheres + d.n_triangles == 0 || nodes[2].primitiveCount > 0) return false;
		qn.push_back( QNode{} );
		q_set( qn[0], 0, nodes[1] ), q_set( qn[0], 1, nodes[2] );
		qn[0].child[2] = qn[0].child[3] = -1;
		qn.push_back( QNode{} );
		qn[0].child[1] = 1;
		q_assign( 1, nodes[2].leftFirst );
		q_collapse( 1 );
		return true;
	}
	// _mm_min_ps / _mm_max_ps: the SECOND operand whenever the comparison is false, i.e. also when either is NaN
	static float mn( float a, float b ) { return a < b ? a : b; }
	static float mx( float a, float b ) { return a > b ? a : b; }
	unsigned q_aabb( const Ray& ray, const QNode& q, float ray_t ) const // BVH::IntersectAABBSIMD, bvh.h:465-486: one bit per lane
	{
		unsigned mask = 0;
		for (int l = 0; l < 4; l++)
		{
			const float tx1 = ( q.bminx[l] - ray.O.x ) * ray.rD.x, tx2 = ( q.bmaxx[l] - ray.O.x ) * ray.rD.x;
			float tmin = mn( tx1, tx2 ), tmax = mx( tx1, tx2 );
			const float ty1 = ( q.bminy[l] - ray.O.y ) * ray.rD.y, ty2 = ( q.bmaxy[l] - ray.O.y ) * ray.rD.y;
			tmin = mx( tmin, mn( ty1, ty2 ) ), tmax = mn( tmax, mx( ty1, ty2 ) );
			const float tz1 = ( q.bminz[l] - ray.O.z ) * ray.rD.z, tz2 = ( q.bmaxz[l] - ray.O.z ) * ray.rD.z;
			tmin = mx( tmin, mn( tz1, tz2 ) ), tmax = mn( tmax, mx( tz1, tz2 ) );
			if (tmax >= tmin && tmin < ray_t && tmax > 0) mask |= 1u << l;
		}
		return mask;
	}
	void qbvh_rec( Ray& ray, int nodeIdx, float ray_t ) const // BVH::IntersectQBVH(ray, nodeIdx, ...), bvh.h:607-638: ray_t is never refreshed
	{
		const QNode& p = qn[nodeIdx];
		const unsigned intersect = q_aabb( ray, p, ray_t );
		for (int i = 0; i < 4; i++)
		{
			if (p.child[i] == -1) return;
			if (!( intersect >> i & 1 )) continue;
			if (p.count[i] > 0) { for (int j = 0; j < p.count[i]; j++) bvh_prim( p.child[i] + j, ray ); }
			else qbvh_rec( ray, p.child[i], ray_t );
		}
	}
	void intersect_qbvh( Ray& ray ) const // BVH::IntersectQBVH(ray), bvh.h:579-605 (no test of the primitive root's own box)
	{
		for (int i = 0; i < d.n_planes; i++) plane( i, ray );
		qbvh_rec( ray, qn[0].child[1], ray.t );
	}
	bool qshadow_rec( Ray& ray, bool isOccluded, int nodeIdx, float ray_t ) const // bvh.h:716-758, statement for statement
	{
		if (isOccluded) return isOccluded;
		const QNode& p = qn[nodeIdx];
		const unsigned intersect = q_aabb( ray, p, ray_t );
		bool prevOccluded = false;
		for (int j = 0; j < 4; j++)
		{
			if (p.child[j] == -1) return false;
			if (!( intersect >> j & 1 )) continue;
			if (p.count[j] > 0)
			{
				for (int i = 0; i < p.count[j]; i++) { bvh_prim( p.child[j] + i, ray ); if (ray.objIdx != -1) return true; }
			}
			else prevOccluded = prevOccluded || qshadow_rec( ray, prevOccluded, p.child[j], ray_t );
		}
		return prevOccluded;
	}
	bool shadow_qbvh( Ray& ray, bool isOccluded ) const // BVH::IntersectQBVHShadowRay, bvh.h:690-714
	{
		if (isOccluded) return isOccluded;
		for (int i = 0; i < d.n_planes; i++) { plane( i, ray ); if (ray.objIdx != -1) return true; }
		return qshadow_rec( ray, isOccluded, qn[0].child[1], ray.t );
	}

	// ---- template/scene.h ----
	void find_nearest( Ray& ray, Counters* c = 0, int mode = 0 ) const // Scene::FindNearest, scene.h:359-368
	{
		if (c) c->extend++;
		if (d.n_quads > 0) quad( 0, ray );
		if (d.n_cubes > 0) cube( 0, ray );
		if (( mode == 3 || ( mode == 0 && use_qbvh ) ) && !qn.empty()) intersect_qbvh( ray );
		else if (mode == 1) intersect_bvh_stack( ray ); else intersect_bvh( ray, c );
	}
	bool is_occluded( Ray& ray, Counters* c = 0 ) const // Scene::IsOccluded, scene.h:369-379
	{
		if (c) c->shadow++;
		if (d.n_quads > 0) quad( 0, ray );
		if (d.n_cubes > 0) cube( 0, ray );
		if (use_qbvh && !qn.empty()) return shadow_qbvh( ray, ray.objIdx != -1 );
		return shadow_bvh( ray, ray.objIdx != -1, c );
	}
	void brute_force( Ray& ray ) const // every primitive, no BVH: the topology-independent answer
	{
		if (d.n_quads > 0) quad( 0, ray );
		if (d.n_cubes > 0) cube( 0, ray );
		for (int i = 0; i < d.n_planes; i++) plane( i, ray );
		for (int k = 0; k < d.n_spheres + d.n_triangles; k++) bvh_prim( k, ray ); // leaf order = the reference's tie order
	}
	V3 cube_normal( int i, V3 I ) const // Cube::GetNormal, primitive.h:237-254
	{
		const float* invM = &cube_inv_M[16 * i], *M = &cube_M[16 * i], *b0 = &cube_b[8 * i], *b1 = b0 + 4;
		V3 objI = xpos( I, invM );
		V3 N = v3( -1, 0, 0 );
		float d0 = fabsf( objI.x - b0[0] ), d1 = fabsf( objI.x - b1[0] );
		float d2 = fabsf( objI.y - b0[1] ), d3 = fabsf( objI.y - b1[1] );
		float d4 = fabsf( objI.z - b0[2] ), d5 = fabsf( objI.z - b1[2] );
		float minDist = d0;
		if (d1 < minDist) minDist = d1, N.x = 1;
		if (d2 < minDist) minDist = d2, N = v3( 0, -1, 0 );
		if (d3 < minDist) minDist = d3, N = v3( 0, 1, 0 );
		if (d4 < minDist) minDist = d4, N = v3( 0, 0, -1 );
		if (d5 < minDist) minDist = d5, N = v3( 0, 0, 1 );
		return xvec( N, M );
	}
	V3 plane_albedo( V3 N, V3 I ) const // Plane::GetAlbedo, primitive.h:163-186
	{
		if (N.y == 1)
		{
			int ix = (int)( I.x * 2 + 96.01f ), iz = (int)( I.z * 2 + 96.01f );
			if (ix == 98 && iz == 98) ix = (int)( I.x * 32.01f ), iz = (int)( I.z * 32.01f );
			if (ix == 94 && iz == 98) ix = (int)( I.x * 64.01f ), iz = (int)( I.z * 64.01f );
			float c = ( ( ix + iz ) & 1 ) ? 1 : 0.3f;
			return v3( c, c, c );
		}
		else if (N.z == -1)
		{
			int ix = (int)( ( I.x + 4 ) * ( 128.0f / 8 ) ), iy = (int)( ( 2 - I.y ) * ( 64.0f / 3 ) );
			uint p = texture.empty() ? 0u : texture[( ix & 127 ) + ( iy & 63 ) * 128];
			return v3( (float)( ( p >> 16 ) & 255 ), (float)( ( p >> 8 ) & 255 ), (float)( p & 255 ) ) * ( 1.0f / 255.0f );
		}
		return v3( 0.93f, 0.93f, 0.93f );
	}
	V3 get_normal( int obj, V3 I, V3 wo ) const // Scene::GetNormal, scene.h:380-414
	{
		V3 N;
		if (obj < sphereBase) N = v3( &primitives[4 * obj] );                                                     // quad N (primitive.h:296-300)
		else if (obj < cubeBase) N = ( I - v3( &sphere_info[4 * ( obj - sphereBase )] ) ) * sphere_invr[obj - sphereBase]; // primitive.h:128-131
		else if (obj < planeBase) N = cube_normal( obj - cubeBase, I );
		else if (obj < triBase) N = v3( &primitives[4 * obj] );
		else N = v3( &primitives[4 * (size_t)obj] );                                                              // Triangle::N, primitive.h:43
		if (dot( N, wo ) > 0) N = -N;
		return N;
	}
	V3 get_albedo( int obj, V3 I ) const // Scene::GetAlbedo, scene.h:415-446
	{
		if (obj < sphereBase) return v3( 10, 10, 10 ); // Quad::GetAlbedo, primitive.h:301-304
		if (obj >= planeBase && obj < triBase) return plane_albedo( v3( &primitives[4 * obj] ), I );
		return v3( &albedo[4 * (size_t)obj] );
	}
	// Scene::RandomPointOnLight, scene.h:317-341 (corners are the marshalled quads[0].c1..c3, same expression)
	void light_point( uint& seed, V3& lp, Counters* c ) const
	{
		const V3 c1 = v3( d.light_corners ), c2 = v3( d.light_corners + 4 ), c3 = v3( d.light_corners + 8 );
		V3 c1c2 = normalize( c1 - c2 );
		float len = rng_f( seed ) * d.light_s;
		V3 u = c1c2 * len;
		V3 c2c3 = normalize( c3 - c2 );
		len = rng_f( seed ) * d.light_s;
		V3 v = c2c3 * len;
		lp = c2 + u + v - v3( 0, 0.01f, 0 );
		if (c) c->draws += 2;
	}
	V3 diffuse_reflection( uint& seed, V3 N, Counters* c ) const // Scene::DiffuseReflection, scene.h:517-532; draws in x, y, z order
	{
		V3 R;
		do
		{
			float x = rng_f( seed ) * 2.0f - 1.0f;
			float y = rng_f( seed ) * 2.0f - 1.0f;
			float z = rng_f( seed ) * 2.0f - 1.0f;
			R = v3( x, y, z );
			if (c) c->draws += 3;
		} while (magnitude( R ) > 1.0f);
		R = normalize( R );
		if (dot( N, R ) < 0) R = -R;
		return R;
	}
	V3 direct_illumination( V3 I, V3 N ) const // Scene::DirectIllumination, scene.h:490-515 (GetLightPos by value)
	{
		V3 acc = v3( 0, 0, 0 );
		for (int i = 0; i < 2; i++)
		{
			V3 dir = v3( d.whitted_light_pos[i] ) - I;
			V3 dirN = normalize( dir );
			float eps = 0.001f;
			Ray sr( I + dirN * eps, dirN );
			float mag = magnitude( dir );
			sr.t = mag - eps * 2;
			if (!is_occluded( sr ))
			{
				float distanceEnergy = 1 / sqrf( sr.t );
				float angularEnergy = smax( dot( N, dirN ), 0.0f );
				acc = acc + v3( d.whitted_light_color[i] ) * d.whitted_intensity * distanceEnergy * angularEnergy;
			}
		}
		return acc;
	}

	// ---- renderer.cpp ----
	V3 sample( Ray ray, uint& seed, Counters* c ) const // Renderer::Sample, renderer.cpp:246-365
	{
		V3 T = v3( 1, 1, 1 ), E = v3( 0, 0, 0 );
		bool lastSpecular = true;
		const V3 emission = v3( d.light_emission );
		const V3 Nl = d.n_quads ? v3( &primitives[0] ) : v3( 0, -1, 0 );
		while (1)
		{
			find_nearest( ray, c );
			if (ray.objIdx == -1) break;
			const int obj = ray.objIdx, type = material_type[obj];
			V3 I = ray.O + ray.t * ray.D;
			V3 N = get_normal( obj, I, ray.D );
			if (type == RTMI_LIGHT)
			{
				if (lastSpecular) return E + T * emission;
				break;
			}
			else if (type == RTMI_MIRROR)
			{
				V3 refl = ray.D - ( 2 * dot( ray.D, N ) ) * N;
				ray = Ray( I + refl * 0.001f, refl );
				lastSpecular = true;
			}
			else if (type == RTMI_GLASS)
			{
				V3 alb = get_albedo( obj, I );
				float air = 1.0003f, n1, n2, mior = ior[obj];
				V3 absorption = v3( 1, 1, 1 );
				if (ray.inside)
				{
					const float* ab = &this->absorption[4 * (size_t)obj];
					absorption = v3( expf( -ab[0] * ray.t ), expf( -ab[1] * ray.t ), expf( -ab[2] * ray.t ) );
					n1 = mior, n2 = air;
				}
				else n1 = air, n2 = mior;
				float ratio = n1 / n2;
				float inc = dot( N, -ray.D );
				float k = 1.0f - sqrf( ratio ) * ( 1.0f - sqrf( inc ) );
				V3 refr = ratio * ray.D + N * ( ratio * inc - sqrtf( k ) );
				float outc = dot( -N, refr );
				double left = sqrf( ( n1 * inc - mior * outc ) / ( n1 * inc + n2 * outc ) );
				double right = sqrf( ( n1 * outc - mior * inc ) / ( n1 * outc + n2 * inc ) );
				float Fr = 0.5f * ( left + right );
				lastSpecular = true;
				T = T * ( alb * absorption );
				bool reflect = k < 0;
				if (!reflect) { reflect = rng_f( seed ) <= Fr; if (c) c->draws++; }
				if (reflect)
				{
					V3 refl = ray.D - ( 2.0f * dot( ray.D, N ) ) * N;
					ray = Ray( I + refl * 0.001f, refl );
				}
				else
				{
					ray = Ray( I + refr * 0.001f, refr );
					ray.inside = !ray.inside; // the fresh ray starts with inside=false, so this is always true
				}
			}
			else
			{
				V3 alb = get_albedo( obj, I );
				V3 BRDF = alb / PI;
				V3 lp;
				light_point( seed, lp, c );
				V3 L = lp - I;
				float dist = magnitude( L );
				L = normalize( L );
				Ray lr( I + L * 0.001f, L, dist - 2.0f * 0.001f );
				if (dot( N, L ) > 0 && dot( Nl, -L ) > 0 && !is_occluded( lr, c ))
				{
					float solidAngle = ( dot( Nl, -L ) * d.light_A ) / sqrf( dist );
					float lightPDF = 1.0f / solidAngle;
					E = E + T * ( dot( N, L ) / lightPDF ) * BRDF * emission;
				}
				float p = smax( alb.z, smax( alb.x, alb.y ) );
				p = p < 1.0f ? p : 1.0f;   // clamp = fmaxf(a, fminf(f, b)) with the template's ternaries, precomp.h:488-489, 799
				p = 0.0f > p ? 0.0f : p;
				if (c) c->draws++;
				if (p < rng_f( seed )) break;
				T = T * ( 1.0f / p );
				V3 R = diffuse_reflection( seed, N, c );
				float hemiPDF = 1.0f / ( PI * 2.0f );
				ray = Ray( I + R * 0.001f, R );
				T = T * ( ( dot( N, R ) / hemiPDF ) * BRDF );
				lastSpecular = false;
			}
		}
		return E;
	}
	V3 trace( Ray ray, int depth, int maxDepth, int mode ) const // Renderer::Trace + ComputePixelColorGlass, renderer.cpp:80-244
	{
		if (depth >= maxDepth) return v3( 0, 0, 0 );
		find_nearest( ray );
		if (ray.objIdx == -1) return v3( 0, 0, 0 );
		const int obj = ray.objIdx, type = material_type[obj];
		V3 I = ray.O + ray.t * ray.D;
		V3 N = get_normal( obj, I, ray.D );
		V3 alb = get_albedo( obj, I );
		if (mode == RTMI_ALBEDO) return alb;
		if (mode == RTMI_NORMAL) return ( N + v3( 1, 1, 1 ) ) * 0.5f;
		if (mode == RTMI_DISTANCE) return 0.1f * v3( ray.t, ray.t, ray.t );
		if (type == RTMI_DIFFUSE || type == RTMI_LIGHT) return alb / PI * direct_illumination( I, N );
		if (type == RTMI_MIRROR)
		{
			V3 refl = ray.D - ( 2 * dot( ray.D, N ) ) * N;
			return v3( &albedo[4 * (size_t)obj] ) * trace( Ray( I + refl * 0.001f, refl ), depth + 1, maxDepth, mode );
		}
		if (type == RTMI_GLASS)
		{
			V3 refl = ray.D - 2.0f * ( dot( ray.D, N ) ) * N;
			Ray reflectRay( I + refl * 0.001f, refl );
			float air = 1.0003f, n1, n2, mior = ior[obj];
			V3 absorption = v3( 1, 1, 1 );
			if (ray.inside)
			{
				const float* ab = &this->absorption[4 * (size_t)obj];
				absorption = v3( expf( -ab[0] * ray.t ), expf( -ab[1] * ray.t ), expf( -ab[2] * ray.t ) );
				n1 = mior, n2 = air;
				reflectRay.inside = true;
			}
			else n1 = air, n2 = mior;
			float ratio = n1 / n2;
			V3 reflectionColor = alb * absorption * trace( reflectRay, depth + 1, maxDepth, mode );
			float inc = dot( N, -ray.D );
			float k = 1.0f - sqrf( ratio ) * ( 1.0f - sqrf( inc ) );
			if (k < 0) return reflectionColor;
			V3 refr = ratio * ray.D + N * ( ratio * inc - sqrtf( k ) );
			Ray refractRay( I + refr * 0.001f, refr );
			refractRay.inside = !ray.inside;
			V3 refractionColor = alb * absorption * trace( refractRay, depth + 1, maxDepth, mode );
			float outc = dot( -N, refr );
			double left = sqrf( ( n1 * inc - mior * outc ) / ( n1 * inc + n2 * outc ) );
			double right = sqrf( ( n1 * outc - mior * inc ) / ( n1 * outc + n2 * inc ) );
			float Fr = 0.5f * ( left + right );
			return Fr * reflectionColor + ( 1 - Fr ) * refractionColor;
		}
		return v3( 0, 0, 0 ); // SUBSTANCE: the reference falls off the switch without a return (UB)
	}
};

template <typename T> void cp( std::vector<T>& dst, const T* src, size_t n ) { if (src && n) dst.assign( src, src + n ); else dst.clear(); }

// Camera::GetPrimaryRay, camera.h:28-44
inline Ray primary_ray( const float* cam, float x, float y, int W, int H )
{
	const V3 camPos = v3( cam ), TL = v3( cam + 4 ), TR = v3( cam + 8 ), BL = v3( cam + 12 );
	const float u = x * ( 1.0f / W );
	const float v = y * ( 1.0f / H );
	const V3 P = TL + u * ( TR - TL ) + v * ( BL - TL );
	return Ray( camPos, normalize( P - camPos ) );
}

} // namespace

extern "C" {

void* orc_create( const rtmi_scene_desc* d )
{
	Oracle* o = new Oracle;
	o->d = *d;
	o->total = d->n_quads + d->n_spheres + d->n_cubes + d->n_planes + d->n_triangles;
	o->sphereBase = d->n_quads, o->cubeBase = o->sphereBase + d->n_spheres, o->planeBase = o->cubeBase + d->n_cubes, o->triBase = o->planeBase + d->n_planes;
	const size_t t = o->total, nb = (size_t)d->n_spheres + d->n_triangles;
	cp( o->quad_inv_T, d->quad_inv_T, 16 * (size_t)d->n_quads ), cp( o->quad_size, d->quad_size, d->n_quads );
	cp( o->sphere_info, d->sphere_info, 4 * (size_t)d->n_spheres ), cp( o->sphere_invr, d->sphere_invr, d->n_spheres );
	cp( o->cube_M, d->cube_M, 16 * (size_t)d->n_cubes ), cp( o->cube_inv_M, d->cube_inv_M, 16 * (size_t)d->n_cubes ), cp( o->cube_b, d->cube_b, 8 * (size_t)d->n_cubes );
	cp( o->tri_verts, d->tri_verts, 12 * (size_t)d->n_triangles );
	cp( o->primitives, d->primitives, 4 * t ), cp( o->albedo, d->albedo, 4 * t ), cp( o->material_type, d->material_type, t ), cp( o->ior, d->ior, t ), cp( o->absorption, d->absorption, 4 * t );
	cp( o->nodes, d->nodes, (size_t)d->n_nodes ), cp( o->prim_idx, d->prim_idx, nb );
	cp( o->texture, d->texture, (size_t)d->tex_w * d->tex_h );
	return o;
}
void orc_destroy( void* h ) { delete (Oracle*)h; }
// Scene::SetQBVH (scene.h:553-556): FindNearest / IsOccluded -- and with them Sample and Trace -- walk the reference's 4-wide QBVH
// (built on first use).  Returns 1 when the QBVH is in use afterwards (0: off, or a scene whose primitive root is a leaf).
int orc_set_qbvh( void* h, int on )
{
	Oracle* o = (Oracle*)h;
	if (on && o->qn.empty()) o->build_qbvh();
	o->use_qbvh = on != 0 && !o->qn.empty();
	return o->use_qbvh ? 1 : 0;
}
int orc_qbvh_nodes( void* h ) { return (int)( (Oracle*)h )->qn.size(); }
void orc_set_threads( int n ) { omp_set_num_threads( n ); }
int orc_max_threads() { return omp_get_max_threads(); }

void orc_primary_rays( int W, int H, const float* cam, float* O, float* D, float* rD )
{
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
	{
		Ray r = primary_ray( cam, (float)x, (float)y, W, H );
		size_t p = (size_t)x + (size_t)y * W;
		memcpy( O + 3 * p, &r.O, 12 ), memcpy( D + 3 * p, &r.D, 12 ), memcpy( rD + 3 * p, &r.rD, 12 );
	}
}
// mode 0 = recursive IntersectBVH (IntersectQBVH after orc_set_qbvh), 1 = IntersectBVHStack, 2 = brute force over every primitive
void orc_find_nearest( void* h, int mode, int n, const float* O, const float* D, const float* tmax, int* obj, float* t )
{
	const Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule( dynamic, 64 )
	for (int i = 0; i < n; i++)
	{
		Ray r( v3( O + 3 * (size_t)i ), v3( D + 3 * (size_t)i ), tmax[i] );
		if (mode == 2) o->brute_force( r ); else o->find_nearest( r, 0, mode );
		obj[i] = r.objIdx, t[i] = r.t;
	}
}
void orc_primary_hits( void* h, int mode, int W, int H, const float* cam, int* obj, float* t )
{
	const Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule( dynamic )
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
	{
		Ray r = primary_ray( cam, (float)x, (float)y, W, H );
		if (mode == 2) o->brute_force( r ); else o->find_nearest( r, 0, mode );
		obj[x + y * W] = r.objIdx, t[x + y * W] = r.t;
	}
}
void orc_is_occluded( void* h, int n, const float* O, const float* D, const float* tmax, int* occ )
{
	const Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule( dynamic, 64 )
	for (int i = 0; i < n; i++)
	{
		Ray r( v3( O + 3 * (size_t)i ), v3( D + 3 * (size_t)i ), tmax[i] );
		occ[i] = o->is_occluded( r ) ? 1 : 0;
	}
}
void orc_hit_attribs( void* h, int n, const int* obj, const float* I, const float* wo, float* N, float* alb, int* mat )
{
	const Oracle* o = (Oracle*)h;
	for (int i = 0; i < n; i++)
	{
		V3 nn = o->get_normal( obj[i], v3( I + 3 * i ), v3( wo + 3 * i ) ), a = o->get_albedo( obj[i], v3( I + 3 * i ) );
		memcpy( N + 3 * i, &nn, 12 ), memcpy( alb + 3 * i, &a, 12 ), mat[i] = o->material_type[obj[i]];
	}
}
// accumulator[p] += Sample(primary ray p) for samples s_begin, s_begin+stride, ... (renderer.cpp:559-567); rows in
// parallel like the reference's `#pragma omp parallel for schedule(dynamic)` (renderer.cpp:561).
// counters (may be null): extend rays, shadow rays, closest-hit nodes, closest-hit prims, rng draws, any-hit nodes, any-hit prims
void orc_sample_rows( void* h, int W, int H, const float* cam, const uint32_t* seeds0, int s_begin, int s_count, int stride, int row_begin, int row_step,
	float* acc, unsigned long long* counters )
{
	const Oracle* o = (Oracle*)h;
	Counters total;
	for (int k = 0; k < s_count; k++)
	{
		const uint s = (uint)( s_begin + k * stride );
#pragma omp parallel
		{
			Counters local;
#pragma omp for schedule( dynamic )
			for (int y = row_begin; y < H; y += row_step) for (int x = 0; x < W; x++)
			{
				const size_t p = (size_t)x + (size_t)y * W;
				uint seed = sample_seed( seeds0[p], s );
				V3 c = o->sample( primary_ray( cam, (float)x, (float)y, W, H ), seed, counters ? &local : 0 );
				acc[4 * p] += c.x, acc[4 * p + 1] += c.y, acc[4 * p + 2] += c.z, acc[4 * p + 3] += 0.0f;
			}
#pragma omp critical
			{
				total.extend += local.extend, total.shadow += local.shadow, total.nodes += local.nodes, total.prims += local.prims;
				total.draws += local.draws, total.snodes += local.snodes, total.sprims += local.sprims;
			}
		}
	}
	if (counters)
		counters[0] += total.extend, counters[1] += total.shadow, counters[2] += total.nodes, counters[3] += total.prims, counters[4] += total.draws,
		counters[5] += total.snodes, counters[6] += total.sprims;
}
void orc_sample( void* h, int W, int H, const float* cam, const uint32_t* seeds0, int s_begin, int s_count, int stride, float* acc, unsigned long long* counters )
{
	orc_sample_rows( h, W, H, cam, seeds0, s_begin, s_count, stride, 0, 1, acc, counters );
}
// accumulator[p] = Trace(primary ray p, 0) (renderer.cpp:578-618)
void orc_trace( void* h, int W, int H, const float* cam, int mode, int maxDepth, int aa, float* out )
{
	const Oracle* o = (Oracle*)h;
	const float pat[8] = { -1.0f / 8.0f, 3.0f / 8.0f, 3.0f / 8.0f, 1.0f / 8.0f, -3.0f / 8.0f, -1.0f / 8.0f, 1.0f / 8.0f, -3.0f / 8.0f }; // renderer.h:31-36
#pragma omp parallel for schedule( dynamic )
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
	{
		float* a = out + 4 * ( (size_t)x + (size_t)y * W );
		if (!aa)
		{
			V3 c = o->trace( primary_ray( cam, (float)x, (float)y, W, H ), 0, maxDepth, mode );
			a[0] = c.x, a[1] = c.y, a[2] = c.z, a[3] = 0;
		}
		else
		{
			a[0] = a[1] = a[2] = a[3] = 0;
			for (int s = 0; s < 4; ++s)
			{
				float sx = x + pat[2 * s], sy = y + pat[2 * s + 1];
				V3 c = o->trace( primary_ray( cam, sx, sy, W, H ), 0, maxDepth, mode );
				a[0] += c.x, a[1] += c.y, a[2] += c.z, a[3] += 0.0f;
			}
			a[0] /= 4.0f, a[1] /= 4.0f, a[2] /= 4.0f, a[3] /= 4.0f;
		}
	}
}
// RGBF32_to_RGB8, precomp.h:431-465 (the scalar branch; the SSE one is dead code)
void orc_finalize( int n, const float* acc, int f, uint32_t* out )
{
	for (int i = 0; i < n; i++)
	{
		const float* v = acc + 4 * (size_t)i;
		float x = v[0], y = v[1], z = v[2];
		if (f > 0) x = x / f, y = y / f, z = z / f;
		uint r = (uint)( 255.0f * smin( 1.0f, x ) ), g = (uint)( 255.0f * smin( 1.0f, y ) ), b = (uint)( 255.0f * smin( 1.0f, z ) );
		out[i] = ( r << 16 ) + ( g << 8 ) + b;
	}
}
// PostProcessing, postprocessing.h:16-99; which: 0 gamma, 1 chromatic aberration, 2 vignette
void orc_postfx( uint32_t* px, int W, int H, int which, float amount )
{
	const float cx = (float)W / 2.0f, cy = (float)H / 2.0f;
	const float max_distance = sqrtf( cx * cx + cy * cy );
	if (which == 0)
	{
		float gc = 1.0f / amount;
		for (int i = 0; i < W * H; i++)
		{
			uint c = px[i], r = ( c & 0xFF0000 ) >> 16, g = ( c & 0x00FF00 ) >> 8, b = c & 0x0000FF;
			r = 255.0f * powf( ( (float)r / 255.0f ), gc ), g = 255.0f * powf( ( (float)g / 255.0f ), gc ), b = 255.0f * powf( ( (float)b / 255.0f ), gc );
			px[i] = ( r << 16 ) + ( g << 8 ) + b;
		}
	}
	else if (which == 1)
	{
		std::vector<uint32_t> buf( (size_t)W * H );
		const float goff = amount, boff = amount * 2;
		for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
		{
			float fx = (float)x, fy = (float)y;
			float ddx = cx - fx, ddy = cy - fy;
			float distance = sqrtf( ddx * ddx + ddy * ddy ), ratio = distance / max_distance;
			uint r = ( px[x + y * W] & 0xFF0000 ) >> 16;
			float vx = fx - cx, vy = fy - cy;
			float inv = 1.0f / sqrtf( vx * vx + vy * vy );
			float dx = vx * inv, dy = vy * inv;
			if (dx != dx || dy != dy) dx = 0, dy = 0; // centre pixel: 0 * inf in the reference, int cast undefined -> offset 0
			float c2x = fx + (int)( dx * goff * ratio ), c2y = fy + (int)( dy * goff * ratio );
			int nx = std::max( 0, std::min( (int)c2x, W - 1 ) ), ny = std::max( 0, std::min( (int)c2y, H - 1 ) );
			uint g = ( px[nx + ny * W] & 0x00FF00 ) >> 8;
			c2x = fx + (int)( dx * boff * ratio ), c2y = fy + (int)( dy * boff * ratio );
			nx = std::max( 0, std::min( (int)c2x, W - 1 ) ), ny = std::max( 0, std::min( (int)c2y, H - 1 ) );
			uint b = px[nx + ny * W] & 0x0000FF;
			buf[x + y * W] = ( r << 16 ) + ( g << 8 ) + b;
		}
		memcpy( px, buf.data(), buf.size() * 4 );
	}
	else
	{
		for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
		{
			float ddx = cx - (float)x, ddy = cy - (float)y;
			float distance = sqrtf( ddx * ddx + ddy * ddy );
			float dimming = smax( 1 - ( distance * amount / max_distance ), 0.0f );
			uint c = px[x + y * W];
			uint r = ( ( c & 0xFF0000 ) >> 16 ) * dimming, g = ( ( c & 0x00FF00 ) >> 8 ) * dimming, b = ( c & 0x0000FF ) * dimming;
			px[x + y * W] = ( r << 16 ) + ( g << 8 ) + b;
		}
	}
}

// ---- unit level on caller data -----------------------------------------------------------------------------------
static Oracle unit_oracle( int nt, const float* v12, int ns, const float* sph4, int np, const float* nd4 )
{
	Oracle o;
	o.d = rtmi_scene_desc{};
	o.d.n_triangles = nt, o.d.n_spheres = ns, o.d.n_planes = np;
	o.sphereBase = 0, o.cubeBase = ns, o.planeBase = ns, o.triBase = ns + np;
	if (nt) o.tri_verts.assign( v12, v12 + 12 * (size_t)nt );
	if (ns) o.sphere_info.assign( sph4, sph4 + 4 * (size_t)ns );
	o.primitives.assign( 4 * (size_t)( ns + np + nt ), 0.0f );
	if (np) memcpy( &o.primitives[4 * (size_t)ns], nd4, 16 * (size_t)np );
	return o;
}
void orc_tri_intersect( int n, const float* O, const float* D, const float* tin, const float* v9, int* hit, float* tout )
{
	for (int i = 0; i < n; i++)
	{
		float v12[12] = { v9[9 * i], v9[9 * i + 1], v9[9 * i + 2], 0, v9[9 * i + 3], v9[9 * i + 4], v9[9 * i + 5], 0, v9[9 * i + 6], v9[9 * i + 7], v9[9 * i + 8], 0 };
		Oracle o = unit_oracle( 1, v12, 0, 0, 0, 0 );
		Ray r( v3( O + 3 * i ), v3( D + 3 * i ), tin[i] );
		o.triangle( 0, r );
		hit[i] = r.objIdx == -1 ? -1 : 7, tout[i] = r.t;
	}
}
void orc_sphere_intersect( int n, const float* O, const float* D, const float* tin, const float* pr4, int* hit, float* tout )
{
	for (int i = 0; i < n; i++)
	{
		float s[4] = { pr4[4 * i], pr4[4 * i + 1], pr4[4 * i + 2], pr4[4 * i + 3] * pr4[4 * i + 3] }; // r2(r * r), primitive.h:106
		Oracle o = unit_oracle( 0, 0, 1, s, 0, 0 );
		Ray r( v3( O + 3 * i ), v3( D + 3 * i ), tin[i] );
		o.sphere( 0, r );
		hit[i] = r.objIdx == -1 ? -1 : 7, tout[i] = r.t;
	}
}
void orc_plane_intersect( int n, const float* O, const float* D, const float* tin, const float* nd4, int* hit, float* tout )
{
	for (int i = 0; i < n; i++)
	{
		Oracle o = unit_oracle( 0, 0, 0, 0, 1, nd4 + 4 * i );
		Ray r( v3( O + 3 * i ), v3( D + 3 * i ), tin[i] );
		o.plane( 0, r );
		hit[i] = r.objIdx == -1 ? -1 : 7, tout[i] = r.t;
	}
}
void orc_quad_intersect( void* h, int n, const float* O, const float* D, const float* tin, int* hit, float* tout )
{
	const Oracle* o = (Oracle*)h;
	for (int i = 0; i < n; i++) { Ray r( v3( O + 3 * i ), v3( D + 3 * i ), tin[i] ); o->quad( 0, r ); hit[i] = r.objIdx, tout[i] = r.t; }
}
void orc_cube_intersect( void* h, int n, const float* O, const float* D, const float* tin, int* hit, float* tout )
{
	const Oracle* o = (Oracle*)h;
	for (int i = 0; i < n; i++) { Ray r( v3( O + 3 * i ), v3( D + 3 * i ), tin[i] ); o->cube( 0, r ); hit[i] = r.objIdx, tout[i] = r.t; }
}
void orc_aabb( int n, const float* O, const float* D, const float* tin, const float* bmin, const float* bmax, int* outScalar, float* outStack )
{
	for (int i = 0; i < n; i++)
	{
		Ray r( v3( O + 3 * i ), v3( D + 3 * i ), tin[i] );
		rtmi_bvh_node b = { bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2], bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2], 0, 0 };
		outScalar[i] = Oracle::aabb( r, b ) ? 1 : 0, outStack[i] = Oracle::aabb_stack( r, b );
	}
}
void orc_tri_derived( int n, const float* v9, float* N, float* centroid ) // Triangle ctor, primitive.h:38-48, 74-79
{
	for (int i = 0; i < n; i++)
	{
		V3 p1 = v3( v9 + 9 * i ), p2 = v3( v9 + 9 * i + 3 ), p3 = v3( v9 + 9 * i + 6 );
		V3 nn = normalize( cross( p2 - p1, p3 - p1 ) ), c = ( p1 + p2 + p3 ) * 0.33333f;
		memcpy( N + 3 * i, &nn, 12 ), memcpy( centroid + 3 * i, &c, 12 );
	}
}
void orc_seeds( uint32_t* out, int n ) { uint s = 0x12345678; for (int i = 0; i < n; i++) out[i] = rng_u( s ); } // renderer.cpp:51-55
void orc_rng_floats( uint32_t s0, int n, float* out ) { uint s = s0; for (int i = 0; i < n; i++) out[i] = rng_f( s ); }
uint32_t orc_sample_seed( uint32_t s0, uint32_t s ) { return sample_seed( s0, s ); }
void orc_light_point( void* h, uint32_t s0, float* out3, uint32_t* s1 )
{
	uint s = s0; V3 lp; ( (Oracle*)h )->light_point( s, lp, 0 ); memcpy( out3, &lp, 12 ); *s1 = s;
}
void orc_diffuse_reflection( void* h, uint32_t s0, const float* N, float* out3, uint32_t* s1 )
{
	uint s = s0; V3 r = ( (Oracle*)h )->diffuse_reflection( s, v3( N ), 0 ); memcpy( out3, &r, 12 ); *s1 = s;
}

// ---- the reference's SAH builder, literally (bvh.h:251-441): O(n^2) per node, for small inputs only ---------------
struct RefBuild
{
	int ns, nt;
	const float* pr4; // spheres: pos, r
	const float* v9;
	std::vector<V3> centroid;
	std::vector<rtmi_bvh_node> nodes;
	std::vector<int> primIdx;
	uint nodesUsed = 3;
	struct Box
	{
		V3 bmin = v3( 1e30f, 1e30f, 1e30f ), bmax = v3( -1e30f, -1e30f, -1e30f );
		static float mn( float a, float b ) { return a < b ? a : b; } // the template's fminf/fmaxf, precomp.h:488-489
		static float mx( float a, float b ) { return a > b ? a : b; }
		void grow( V3 p ) { bmin = v3( mn( bmin.x, p.x ), mn( bmin.y, p.y ), mn( bmin.z, p.z ) ), bmax = v3( mx( bmax.x, p.x ), mx( bmax.y, p.y ), mx( bmax.z, p.z ) ); }
		float area() const { V3 e = bmax - bmin; return e.x * e.y + e.y * e.z + e.z * e.x; }
	};
	float cen( int prim, int axis ) const { return prim < ns ? pr4[4 * prim + axis] : ( &centroid[prim - ns].x )[axis]; }
	void growPrim( int prim, Box& b ) const
	{
		if (prim < ns) { V3 p = v3( pr4 + 4 * prim ); float r = pr4[4 * prim + 3]; b.grow( v3( p.x - r, p.y - r, p.z - r ) ), b.grow( v3( p.x + r, p.y + r, p.z + r ) ); }
		else { const float* v = v9 + 9 * (size_t)( prim - ns ); b.grow( v3( v ) ), b.grow( v3( v + 3 ) ), b.grow( v3( v + 6 ) ); }
	}
	void bounds( uint idx ) // UpdateNodeBounds, bvh.h:290-316
	{
		rtmi_bvh_node& n = nodes[idx];
		Box b;
		for (int i = 0; i < n.primitiveCount; i++)
		{
			const int prim = primIdx[n.leftFirst + i];
			if (prim < ns) { V3 p = v3( pr4 + 4 * prim ); float r = pr4[4 * prim + 3]; V3 lo = v3( p.x - r, p.y - r, p.z - r ), hi = v3( p.x + r, p.y + r, p.z + r );
				b.bmin = v3( Box::mn( b.bmin.x, lo.x ), Box::mn( b.bmin.y, lo.y ), Box::mn( b.bmin.z, lo.z ) ), b.bmax = v3( Box::mx( b.bmax.x, hi.x ), Box::mx( b.bmax.y, hi.y ), Box::mx( b.bmax.z, hi.z ) ); }
			else growPrim( prim, b );
		}
		n.aabbMinx = b.bmin.x, n.aabbMiny = b.bmin.y, n.aabbMinz = b.bmin.z, n.aabbMaxx = b.bmax.x, n.aabbMaxy = b.bmax.y, n.aabbMaxz = b.bmax.z;
	}
	float sah( const rtmi_bvh_node& n, int axis, float pos ) const // EvaluateSAH, bvh.h:395-441
	{
		Box l, r;
		int lc = 0, rc = 0;
		for (int i = 0; i < n.primitiveCount; i++)
		{
			const int prim = primIdx[n.leftFirst + i];
			if (cen( prim, axis ) < pos) lc++, growPrim( prim, l ); else rc++, growPrim( prim, r );
		}
		float cost = lc * l.area() + rc * r.area();
		return cost > 0 ? cost : 1e30f;
	}
	void subdivide( uint idx ) // Subdivide, bvh.h:318-393
	{
		rtmi_bvh_node& node = nodes[idx];
		float ex = node.aabbMaxx - node.aabbMinx, ey = node.aabbMaxy - node.aabbMiny, ez = node.aabbMaxz - node.aabbMinz;
		float parentArea = ex * ey + ey * ez + ez * ex;
		float parentCost = (uint)node.primitiveCount * parentArea;
		if (node.primitiveCount <= 2) return;
		int bestAxis = -1;
		float bestPos = 0, bestCost = 1e30f;
		for (int axis = 0; axis < 3; axis++) for (int i = 0; i < node.primitiveCount; i++)
		{
			float pos = cen( primIdx[node.leftFirst + i], axis );
			float cost = sah( node, axis, pos );
			if (cost < bestCost) bestPos = pos, bestAxis = axis, bestCost = cost;
		}
		if (bestCost >= parentCost) return;
		int i = node.leftFirst, j = i + node.primitiveCount - 1;
		while (i <= j)
		{
			if (cen( primIdx[i], bestAxis ) < bestPos) i++; else std::swap( primIdx[i], primIdx[j--] );
		}
		int leftCount = i - node.leftFirst;
		if (leftCount == 0 || leftCount == node.primitiveCount) return;
		int l = nodesUsed++, r = nodesUsed++;
		nodes[l].leftFirst = node.leftFirst, nodes[l].primitiveCount = leftCount;
		nodes[r].leftFirst = i, nodes[r].primitiveCount = node.primitiveCount - leftCount;
		node.leftFirst = l, node.primitiveCount = 0;
		bounds( l ), bounds( r );
		subdivide( l ), subdivide( r );
	}
};
int orc_bvh_build_reference( const float* v9, int nt, const float* pr4, int ns, rtmi_bvh_node* nodes_out, int* prim_idx_out )
{
	RefBuild b;
	b.ns = ns, b.nt = nt, b.pr4 = pr4, b.v9 = v9;
	b.centroid.resize( nt );
	for (int i = 0; i < nt; i++) b.centroid[i] = ( v3( v9 + 9 * (size_t)i ) + v3( v9 + 9 * (size_t)i + 3 ) + v3( v9 + 9 * (size_t)i + 6 ) ) * 0.33333f;
	b.nodes.assign( (size_t)( ns + nt ) * 2 + 1, rtmi_bvh_node{} );
	b.primIdx.resize( ns + nt );
	for (int i = 0; i < ns + nt; i++) b.primIdx[i] = i;
	b.nodes[0] = { -1e30f, -1e30f, -1e30f, 1e30f, 1e30f, 1e30f, 0, 0 };
	b.nodes[1] = { -1e30f, -1e30f, -1e30f, 1e30f, 1e30f, 1e30f, 0, 0 };
	b.nodes[2].leftFirst = 0, b.nodes[2].primitiveCount = ns + nt;
	b.bounds( 2 );
	b.subdivide( 2 );
	memcpy( nodes_out, b.nodes.data(), sizeof( rtmi_bvh_node ) * b.nodesUsed );
	memcpy( prim_idx_out, b.primIdx.data(), sizeof( int ) * ( ns + nt ) );
	return (int)b.nodesUsed;
}


// ---- Camera (camera.h:11-168) driven by a script of events, for the fly-through tests -------------------------------
// ev[2k] = kind, ev[2k+1] = GLFW key (renderer.cpp:368-441): kind 0 KeyDown, 1 KeyUp, 2 Camera::Update (camera.h:95-140;
// writes gpuCamProp = camPos, topLeft, topRight, bottomLeft to cam_out[16 * u] and the return value to moved_out[u] for
// the u-th update), 3 SetFov( (int)val[k] ) (camera.h:142-147), 4 SetAspectRatio( val[k] ) (camera.h:149-159).
// Returns the number of updates.  Restated by reading: no reference-derived vector covers it (parity unpinned).
int orc_camera_script( int W, int H, int n, const int* ev, const float* val, float* cam_out, int* moved_out )
{
	const float aspect = (float)W / (float)H; // camera.h:46
	V3 camPos = v3( 0, 0, -2 ), topLeft = v3( -aspect, 1, 0 ), topRight = v3( aspect, 1, 0 ), bottomLeft = v3( -aspect, -1, 0 ); // :20-23
	V3 translation = v3( 0, 0, 0 ), sub[6] = { v3( 0, 0, 0 ), v3( 0, 0, 0 ), v3( 0, 0, 0 ), v3( 0, 0, 0 ), v3( 0, 0, 0 ), v3( 0, 0, 0 ) };
	bool rot[4] = { false, false, false, false };
	const float angular_velocity = PI / 180; // :15
	enum { Left, Right, Up, Down, In, Out };
	auto center = [&]() { // :58-63
		const float u = W / 2.0f * ( 1.0f / W ), v = H / 2.0f * ( 1.0f / H );
		return topLeft + u * ( topRight - topLeft ) + v * ( bottomLeft - topLeft ); };
	auto direction = [&]() { return normalize( center() - camPos ); }; // :65-68
	auto translate = [&]( V3 t, int dir ) { translation = translation + t, sub[dir] = t; }; // :52-56
	auto undo = [&]( int dir ) { translation = translation - sub[dir], sub[dir] = v3( 0, 0, 0 ); }; // :71-75
	int updates = 0;
	for (int k = 0; k < n; k++)
	{
		const int kind = ev[2 * k], key = ev[2 * k + 1];
		const float velocity = .05f; // renderer.cpp:369
		if (kind == 0) switch (key) // Renderer::KeyDown, renderer.cpp:368-404
		{
		case 65: translate( normalize( topRight - topLeft ) * -velocity, Left ); break;    // A: MoveHorizontal, camera.h:77-81
		case 68: translate( normalize( topRight - topLeft ) * velocity, Right ); break;    // D
		case 87: translate( normalize( topLeft - bottomLeft ) * velocity, Up ); break;     // W: MoveVertical, :83-87
		case 83: translate( normalize( topLeft - bottomLeft ) * -velocity, Down ); break;  // S
		case 69: translate( direction() * velocity, In ); break;                           // E: MoveDistal, :89-92
		case 81: translate( direction() * -velocity, Out ); break;                         // Q
		case 265: rot[2] = true; break; case 264: rot[3] = true; break; case 263: rot[0] = true; break; case 262: rot[1] = true; break;
		default: break;
		}
		else if (kind == 1) switch (key) // Renderer::KeyUp, renderer.cpp:406-441
		{
		case 65: undo( Left ); break; case 68: undo( Right ); break; case 87: undo( Up ); break; case 83: undo( Down ); break;
		case 69: undo( In ); break; case 81: undo( Out ); break;
		case 265: rot[2] = false; break; case 264: rot[3] = false; break; case 263: rot[0] = false; break; case 262: rot[1] = false; break;
		default: break;
		}
		else if (kind == 2) // Camera::Update, camera.h:95-140
		{
			V3 rotation = v3( 0, 0, 0 );
			const V3 horizontalDirection = normalize( topLeft - bottomLeft ), verticalDirection = normalize( topRight - topLeft );
			if (rot[0]) rotation = rotation + horizontalDirection * angular_velocity;
			else if (rot[1]) rotation = rotation - horizontalDirection * angular_velocity;
			else if (rot[2]) rotation = rotation + verticalDirection * angular_velocity;
			else if (rot[3]) rotation = rotation - verticalDirection * angular_velocity;
			if (rot[0] || rot[1] || rot[2] || rot[3])
			{
				// mat4::Rotate( normalize( rotation ), magnitude( rotation ) ), precomp.h:1012-1022, row major
				const V3 u = normalize( rotation );
				const float a = magnitude( rotation ), x = u.x, y = u.y, z = u.z;
				const float c = cosf( a ), l_c = 1 - c, sn = sinf( a );
				float m[16];
				m[0] = x * x + ( 1 - x * x ) * c, m[1] = x * y * l_c + z * sn, m[2] = x * z * l_c - y * sn, m[3] = 0;
				m[4] = x * y * l_c - z * sn, m[5] = y * y + ( 1 - y * y ) * c, m[6] = y * z * l_c + x * sn, m[7] = 0;
				m[8] = x * z * l_c + y * sn, m[9] = y * z * l_c - x * sn, m[10] = z * z + ( 1 - z * z ) * c, m[11] = 0;
				m[12] = m[13] = m[14] = 0, m[15] = 1;
				// (float3)(m * float4( p - camPos )) with w = 0 (template.cpp:960-964, 1014-1020) + camPos
				topLeft = xvec( topLeft - camPos, m ) + camPos;
				topRight = xvec( topRight - camPos, m ) + camPos;
				bottomLeft = xvec( bottomLeft - camPos, m ) + camPos;
			}
			if (translation.x + translation.y + translation.z != 0)
				camPos = camPos + translation, topLeft = topLeft + translation, topRight = topRight + translation, bottomLeft = bottomLeft + translation;
			float* o = cam_out + 16 * updates;
			const V3 p[4] = { camPos, topLeft, topRight, bottomLeft };
			for (int q = 0; q < 4; q++) o[4 * q] = p[q].x, o[4 * q + 1] = p[q].y, o[4 * q + 2] = p[q].z, o[4 * q + 3] = 0;
			if (magnitude( translation ) < 0.01f) translation = v3( 0, 0, 0 );
			if (magnitude( rotation ) < 0.01f) rotation = v3( 0, 0, 0 );
			moved_out[updates++] = ( magnitude( translation ) != 0 || magnitude( rotation ) != 0 ) ? 1 : 0;
		}
		else if (kind == 3) // SetFov, camera.h:142-147
		{
			const int fov = (int)val[k];
			const float size = magnitude( topRight - topLeft );
			const float distance = size / ( 2.0f * tanf( (float)fov * PI / ( 2.0f * 180.0f ) ) );
			camPos = center() - direction() * distance;
		}
		else if (kind == 4) // SetAspectRatio, camera.h:149-159
		{
			const float asp = val[k];
			const V3 mid = ( topLeft + topRight ) / 2;
			const V3 dw = normalize( topRight - topLeft ), dh = normalize( topLeft - bottomLeft );
			topLeft = mid - dw * asp, topRight = mid + dw * asp, bottomLeft = topLeft - dh * 2;
		}
	}
	return updates;
}

} // extern "C"
