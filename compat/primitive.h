// compat/primitive.h -- primitive.h of the reference: Ray, Triangle, Sphere, Plane, Cube, Quad with the reference's fields and
// constructors (the rtmi:: classes) plus its method surface.  Intersect( ray ) runs the library's own device routine for that one
// primitive (rtmi_intersect_primitive: the same code the BVH walks call), one ray per call as in the reference's CPU loop: it is
// there so that existing call sites compile and agree bit for bit, not to be fast.  GetNormal / GetAlbedo are the accessors of
// primitive.h:74-92, 129-141, 160-190, 238-262, 297-314.
#pragma once
#include "material.h"
namespace Tmpl8 {

namespace detail {
inline void IntersectOne( int kind, const float* data, Ray& ray, int objIdx )
{
	int32_t hit = 0;
	float t = ray.t;
	if (rtmi_intersect_primitive( CompatDevice(), kind, data, 1, &ray.O.x, &ray.D.x, &ray.t, &hit, &t ) != RTMI_OK)
		FatalError( "rtmi_intersect_primitive: %s", rtmi_last_error( CompatDevice() ) );
	if (hit) ray.t = t, ray.objIdx = objIdx;
}
}

class Triangle : public rtmi::Triangle // primitive.h:34-94
{
public:
	using rtmi::Triangle::Triangle;
	void Intersect( Ray& ray ) const // primitive.h:50-72
	{
		const float v9[9] = { pos1.x, pos1.y, pos1.z, pos2.x, pos2.y, pos2.z, pos3.x, pos3.y, pos3.z };
		detail::IntersectOne( 0, v9, ray, objIdx );
	}
	float3 GetNormal( const float3 ) const { return normalize( cross( pos2 - pos1, pos3 - pos1 ) ); } // == N, recomputed as primitive.h:74-79 does
	float3 GetAlbedo( const float3 ) const { return float3( 1 ); }
};

class Sphere : public rtmi::Sphere // primitive.h:101-142
{
public:
	using rtmi::Sphere::Sphere;
	void Intersect( Ray& ray ) const // primitive.h:108-127
	{
		const float d[4] = { pos.x, pos.y, pos.z, r2 };
		detail::IntersectOne( 1, d, ray, objIdx );
	}
	float3 GetNormal( const float3 I ) const { return ( I - pos ) * invr; }
	float3 GetAlbedo( const float3 ) const { return float3( 0.93f ); }
};

class Plane : public rtmi::Plane // primitive.h:149-191
{
public:
	using rtmi::Plane::Plane;
	void Intersect( Ray& ray ) const // primitive.h:154-158
	{
		const float nd[4] = { N.x, N.y, N.z, d };
		detail::IntersectOne( 2, nd, ray, objIdx );
	}
	float3 GetNormal( const float3 ) const { return N; }
	// Plane::GetAlbedo (checkerboard floor, logo wall: primitive.h:164-186) reads the texture and exists on the device only; its one
	// caller in the reference is Scene::GetAlbedo, which compat/scene.h serves through rtmi_hit_attributes
};

class Cube : public rtmi::Cube // primitive.h:199-263
{
public:
	using rtmi::Cube::Cube;
	void Intersect( Ray& ray ) const // primitive.h:210-236
	{
		float d[18];
		memcpy( d, invM.cell, 48 );
		d[12] = b[0].x, d[13] = b[0].y, d[14] = b[0].z, d[15] = b[1].x, d[16] = b[1].y, d[17] = b[1].z;
		detail::IntersectOne( 4, d, ray, objIdx );
	}
	float3 GetNormal( const float3 I ) const // primitive.h:238-257: the face whose slab plane is nearest in object space, first on ties
	{
		const float3 o = TransformPosition( I, invM );
		int face = 0; // 2 * axis + (1 = the b[1] side)
		float nearest = fabsf( o.x - b[0].x );
		for (int k = 1; k < 6; k++)
		{
			const float dk = fabsf( o[k >> 1] - b[k & 1][k >> 1] );
			if (dk < nearest) nearest = dk, face = k;
		}
		float3 n( 0 );
		n[face >> 1] = ( face & 1 ) ? 1.0f : -1.0f;
		return TransformVector( n, M );
	}
	float3 GetAlbedo( const float3 ) const { return float3( 1, 1, 1 ); }
};

class Quad : public rtmi::Quad // primitive.h:269-315
{
public:
	using rtmi::Quad::Quad;
	void Intersect( Ray& ray ) const // primitive.h:284-295
	{
		float d[13];
		memcpy( d, invT.cell, 48 );
		d[12] = size;
		detail::IntersectOne( 3, d, ray, objIdx );
	}
	float3 GetNormal( const float3 ) const { return float3( -T.cell[1], -T.cell[5], -T.cell[9] ); }
	float3 GetAlbedo( const float3 ) const { return float3( 10 ); }
};

// the scene container stores rtmi:: objects; the classes above add methods only, so its arrays are read through them in place
static_assert( sizeof( Triangle ) == sizeof( rtmi::Triangle ) && sizeof( Sphere ) == sizeof( rtmi::Sphere ) && sizeof( Plane ) == sizeof( rtmi::Plane )
	&& sizeof( Cube ) == sizeof( rtmi::Cube ) && sizeof( Quad ) == sizeof( rtmi::Quad ), "compat primitives add no data" );

} // namespace Tmpl8
