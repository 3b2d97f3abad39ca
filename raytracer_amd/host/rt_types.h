// rt_types.h -- host-side scene types of the drop-in API (C++17, header only).
//
// Same names, fields and derived values as the reference's host types so that code written against
// camera.h / primitive.h / material.h keeps compiling against this tree:
//   float3/float4/mat4 + free math  <- template/precomp.h:252-271, 488-491, 822-875, 976-1100; template.cpp:975-1035
//   Material                        <- material.h:3-24
//   Ray, Triangle, Sphere, Plane, Cube, Quad (data + constructors)  <- primitive.h:3-315
//   Camera                          <- camera.h:11-168
// Intersection itself is NOT here: on this platform it runs in the HIP kernels behind include/rtmi.h.
// All arithmetic is strict IEEE binary32 in the operation order of the cited lines (build with
// -ffp-contract=off), because derived values (normals, centroids, inverse transforms, light corners)
// are uploaded and must be bit-identical to what the reference host computes.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rtmi {

typedef unsigned int uint;

struct float3
{
	float x, y, z;
	float3() = default;
	float3( float a, float b, float c ) : x( a ), y( b ), z( c ) {}
	float3( float a ) : x( a ), y( a ), z( a ) {}
	float& operator[]( int n ) { return ( &x )[n]; }
	float operator[]( int n ) const { return ( &x )[n]; }
};
struct alignas( 16 ) float4
{
	float x, y, z, w;
	float4() = default;
	float4( float a, float b, float c, float d ) : x( a ), y( b ), z( c ), w( d ) {}
	float4( float a ) : x( a ), y( a ), z( a ), w( a ) {}
	float4( const float3& a, float d ) : x( a.x ), y( a.y ), z( a.z ), w( d ) {}
};

inline float3 operator-( const float3& a ) { return float3( -a.x, -a.y, -a.z ); }
inline float3 operator+( const float3& a, const float3& b ) { return float3( a.x + b.x, a.y + b.y, a.z + b.z ); }
inline float3 operator-( const float3& a, const float3& b ) { return float3( a.x - b.x, a.y - b.y, a.z - b.z ); }
inline float3 operator+( const float3& a, float b ) { return float3( a.x + b, a.y + b, a.z + b ); }
inline float3 operator-( const float3& a, float b ) { return float3( a.x - b, a.y - b, a.z - b ); }
inline float3 operator*( const float3& a, const float3& b ) { return float3( a.x * b.x, a.y * b.y, a.z * b.z ); }
inline float3 operator*( const float3& a, float b ) { return float3( a.x * b, a.y * b, a.z * b ); }
inline float3 operator*( float b, const float3& a ) { return float3( b * a.x, b * a.y, b * a.z ); }
inline float3 operator/( const float3& a, float b ) { return float3( a.x / b, a.y / b, a.z / b ); }
inline void operator+=( float3& a, const float3& b ) { a.x += b.x, a.y += b.y, a.z += b.z; }
inline void operator-=( float3& a, const float3& b ) { a.x -= b.x, a.y -= b.y, a.z -= b.z; }
// the template's own ternary min/max (precomp.h:488-489), not libm's
inline float tmin( float a, float b ) { return a < b ? a : b; }
inline float tmax( float a, float b ) { return a > b ? a : b; }
inline float3 fminf3( const float3& a, const float3& b ) { return float3( tmin( a.x, b.x ), tmin( a.y, b.y ), tmin( a.z, b.z ) ); }
inline float3 fmaxf3( const float3& a, const float3& b ) { return float3( tmax( a.x, b.x ), tmax( a.y, b.y ), tmax( a.z, b.z ) ); }
inline float sqrf( float x ) { return x * x; }
inline float dot( const float3& a, const float3& b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross( const float3& a, const float3& b ) { return float3( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x ); }
inline float3 normalize( const float3& v ) { float inv = 1.0f / sqrtf( dot( v, v ) ); return v * inv; }
inline float magnitude( const float3& v ) { return sqrtf( v.x * v.x + v.y * v.y + v.z * v.z ); }

constexpr float PI = 3.14159265358979323846264f; // template/common.h:8

// row-major 4x4, translation in cells 3/7/11 (precomp.h:976-1067)
struct alignas( 16 ) mat4
{
	float cell[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
	float& operator[]( int i ) { return cell[i]; }
	static mat4 Identity() { return mat4{}; }
	static mat4 RotateX( float a ) { mat4 r; r.cell[5] = cosf( a ), r.cell[6] = -sinf( a ), r.cell[9] = sinf( a ), r.cell[10] = cosf( a ); return r; }
	static mat4 RotateY( float a ) { mat4 r; r.cell[0] = cosf( a ), r.cell[2] = sinf( a ), r.cell[8] = -sinf( a ), r.cell[10] = cosf( a ); return r; }
	static mat4 RotateZ( float a ) { mat4 r; r.cell[0] = cosf( a ), r.cell[1] = -sinf( a ), r.cell[4] = sinf( a ), r.cell[5] = cosf( a ); return r; }
	static mat4 Translate( const float3& P ) { mat4 r; r.cell[3] = P.x, r.cell[7] = P.y, r.cell[11] = P.z; return r; }
	// axis-angle, row major (precomp.h:1012-1022); used by Camera::Update
	static mat4 Rotate( const float3& u, float a )
	{
		const float x = u.x, y = u.y, z = u.z, c = cosf( a ), l_c = 1 - c, s = sinf( a );
		mat4 m;
		m[0] = x * x + ( 1 - x * x ) * c, m[1] = x * y * l_c + z * s, m[2] = x * z * l_c - y * s, m[3] = 0;
		m[4] = x * y * l_c - z * s, m[5] = y * y + ( 1 - y * y ) * c, m[6] = y * z * l_c + x * s, m[7] = 0;
		m[8] = x * z * l_c + y * s, m[9] = y * z * l_c - x * s, m[10] = z * z + ( 1 - z * z ) * c, m[11] = 0;
		m[12] = m[13] = m[14] = 0, m[15] = 1;
		return m;
	}
	// transpose the 3x3, then -R^T t (precomp.h:1077-1100)
	mat4 FastInvertedTransformNoScale() const
	{
		mat4 r;
		r[0] = cell[0], r[1] = cell[4], r[2] = cell[8];
		r[4] = cell[1], r[5] = cell[5], r[6] = cell[9];
		r[8] = cell[2], r[9] = cell[6], r[10] = cell[10];
		r[3] = -( cell[3] * r[0] + cell[7] * r[1] + cell[11] * r[2] );
		r[7] = -( cell[3] * r[4] + cell[7] * r[5] + cell[11] * r[6] );
		r[11] = -( cell[3] * r[8] + cell[7] * r[9] + cell[11] * r[10] );
		return r;
	}
};
inline mat4 operator*( const mat4& a, const mat4& b ) // template.cpp:975-988
{
	mat4 r;
	for (int i = 0; i < 16; i += 4) for (int j = 0; j < 4; ++j)
		r.cell[i + j] = ( a.cell[i + 0] * b.cell[j + 0] ) + ( a.cell[i + 1] * b.cell[j + 4] ) + ( a.cell[i + 2] * b.cell[j + 8] ) + ( a.cell[i + 3] * b.cell[j + 12] );
	return r;
}
// (a, w) * M with rows 0..2 of M, four products summed left to right (template.cpp:1021-1035)
inline float3 TransformPosition( const float3& a, const mat4& M )
{
	return float3( M.cell[0] * a.x + M.cell[1] * a.y + M.cell[2] * a.z + M.cell[3] * 1.0f,
		M.cell[4] * a.x + M.cell[5] * a.y + M.cell[6] * a.z + M.cell[7] * 1.0f,
		M.cell[8] * a.x + M.cell[9] * a.y + M.cell[10] * a.z + M.cell[11] * 1.0f );
}
inline float3 TransformVector( const float3& a, const mat4& M )
{
	return float3( M.cell[0] * a.x + M.cell[1] * a.y + M.cell[2] * a.z + M.cell[3] * 0.0f,
		M.cell[4] * a.x + M.cell[5] * a.y + M.cell[6] * a.z + M.cell[7] * 0.0f,
		M.cell[8] * a.x + M.cell[9] * a.y + M.cell[10] * a.z + M.cell[11] * 0.0f );
}
inline float3 MulPoint( const mat4& M, const float3& v ) // (float3)(M * float4(v,0)), camera.h:112-114
{
	return TransformVector( v, M );
}

class Material // material.h:3-24
{
public:
	enum class MaterialType { DIFFUSE, MIRROR, GLASS, SUBSTANCE, LIGHT };
	Material() = default;
	Material( MaterialType type, float3 color, float specularity, float refractive_index = 1, float3 absorption = float3( 0 ), float3 emission = float3( 0 ) )
		: type( type ), color( color ), specularity( specularity ), refractive_index( refractive_index ), absorption( absorption ), emission( emission ) {}
	MaterialType type = MaterialType::DIFFUSE;
	float3 color = float3( 0 );
	float specularity = 0;
	float refractive_index = 1;
	float3 absorption = float3( 0 );
	float3 emission = float3( 0 );
};

struct alignas( 64 ) Ray // primitive.h:3-28
{
	Ray() = default;
	Ray( float3 origin, float3 direction, float distance = 1e34f )
	{
		O = origin, D = direction, t = distance;
		rD = float3( 1 / D.x, 1 / D.y, 1 / D.z );
	}
	float3 IntersectionPoint() const { return O + t * D; }
	float3 O, D, rD;
	float t = 1e34f;
	int objIdx = -1;
	bool inside = false;
};

class Triangle // primitive.h:34-94
{
public:
	Triangle() = default;
	Triangle( int idx, float3 p1, float3 p2, float3 p3, Material mat ) : pos1( p1 ), pos2( p2 ), pos3( p3 ), objIdx( idx ), material( mat )
	{
		N = normalize( cross( pos2 - pos1, pos3 - pos1 ) );
		D = -dot( N, pos1 );
		centroid = ( pos1 + pos2 + pos3 ) * 0.33333f; // sic, primitive.h:47
	}
	float3 pos1 = 0, pos2 = 0, pos3 = 0;
	int objIdx = -1;
	float3 N;
	float D;
	Material material;
	float3 centroid = float3( 0 );
};

class Sphere // primitive.h:101-142
{
public:
	Sphere() = default;
	Sphere( int idx, float3 p, float r, Material mat ) : pos( p ), r( r ), r2( r * r ), invr( 1 / r ), objIdx( idx ), material( mat ) {}
	float3 pos = 0;
	float r = 0, r2 = 0, invr = 0;
	int objIdx = -1;
	Material material;
};

class Plane // primitive.h:149-191
{
public:
	Plane() = default;
	Plane( int idx, float3 normal, float dist, Material mat ) : N( normal ), d( dist ), objIdx( idx ), material( mat ) {}
	float3 N;
	float d;
	int objIdx = -1;
	Material material;
};

class Cube // primitive.h:199-263
{
public:
	Cube() = default;
	Cube( int idx, float3 pos, float3 size, Material mat, mat4 transform = mat4::Identity() ) : material( mat )
	{
		objIdx = idx;
		b[0] = pos - 0.5f * size, b[1] = pos + 0.5f * size;
		M = transform, invM = transform.FastInvertedTransformNoScale();
	}
	float3 b[2];
	mat4 M, invM;
	int objIdx = -1;
	Material material;
};

class Quad // primitive.h:269-315
{
public:
	Quad() = default;
	Quad( int idx, float s, Material mat, mat4 transform = mat4::Identity() ) : s( s ), material( mat )
	{
		objIdx = idx;
		size = s * 0.5f;
		T = transform, invT = transform.FastInvertedTransformNoScale();
		N = float3( -T.cell[1], -T.cell[5], -T.cell[9] );
		c1 = TransformPosition( float3( -size, 0, -size ), T );
		c2 = TransformPosition( float3( size, 0, -size ), T );
		c3 = TransformPosition( float3( size, 0, size ), T );
		A = sqrf( s );
	}
	float size, s;
	mat4 T, invT;
	int objIdx = -1;
	Material material;
	float3 N;
	float3 c1, c2, c3;
	float A;
};

// camera.h:11-168; the resolution is a runtime parameter here (compile-time SCRWIDTH/SCRHEIGHT there, camera.h:4-5)
class Camera
{
public:
	enum Direction { Left, Right, Up, Down, In, Out };
	float angular_velocity = PI / 180;
	Camera( int width = 1280, int height = 720 ) : scrwidth( width ), scrheight( height )
	{
		aspect = (float)width / (float)height;
		camPos = float3( 0, 0, -2 );
		topLeft = float3( -aspect, 1, 0 );
		topRight = float3( aspect, 1, 0 );
		bottomLeft = float3( -aspect, -1, 0 );
		SyncGpuProps();
	}
	// host restatement of camera.h:28-35 (the device kernel computes the same expression per pixel)
	Ray GetPrimaryRay( const int x, const int y ) const { return GetPrimaryRay( (float)x, (float)y ); }
	Ray GetPrimaryRay( const float x, const float y ) const
	{
		const float u = x * ( 1.0f / scrwidth );
		const float v = y * ( 1.0f / scrheight );
		const float3 P = topLeft + u * ( topRight - topLeft ) + v * ( bottomLeft - topLeft );
		return Ray( camPos, normalize( P - camPos ) );
	}
	void Translate( float3 translate, int direction ) { translation += translate; subTranslation[direction] = translate; }
	void Translate( int direction ) { translation -= subTranslation[direction]; subTranslation[direction] = float3( 0, 0, 0 ); }
	float3 Center() const
	{
		float u = scrwidth / 2.0f * ( 1.0f / scrwidth ), v = scrheight / 2.0f * ( 1.0f / scrheight );
		return topLeft + u * ( topRight - topLeft ) + v * ( bottomLeft - topLeft );
	}
	float3 Direction() const { return normalize( Center() - camPos ); }
	void MoveHorizontal( float speed, int direction ) { Translate( normalize( topRight - topLeft ) * speed, direction ); }
	void MoveVertical( float speed, int direction ) { Translate( normalize( topLeft - bottomLeft ) * speed, direction ); }
	void MoveDistal( float speed, int direction ) { Translate( Direction() * speed, direction ); }
	bool Update() // camera.h:95-140
	{
		float3 rotation = float3( 0 );
		float3 horizontalDirection = normalize( topLeft - bottomLeft );
		float3 verticalDirection = normalize( topRight - topLeft );
		if (rotateDirections[0]) rotation += horizontalDirection * angular_velocity;
		else if (rotateDirections[1]) rotation -= horizontalDirection * angular_velocity;
		else if (rotateDirections[2]) rotation += verticalDirection * angular_velocity;
		else if (rotateDirections[3]) rotation -= verticalDirection * angular_velocity;
		if (rotateDirections[0] || rotateDirections[1] || rotateDirections[2] || rotateDirections[3])
		{
			mat4 m = mat4::Rotate( normalize( rotation ), magnitude( rotation ) );
			topLeft = MulPoint( m, topLeft - camPos ) + camPos;
			topRight = MulPoint( m, topRight - camPos ) + camPos;
			bottomLeft = MulPoint( m, bottomLeft - camPos ) + camPos;
		}
		if (translation.x + translation.y + translation.z != 0)
			camPos += translation, topLeft += translation, topRight += translation, bottomLeft += translation;
		SyncGpuProps();
		if (magnitude( translation ) < 0.01f) translation = float3( 0 );
		if (magnitude( rotation ) < 0.01f) rotation = float3( 0 );
		return magnitude( translation ) != 0 || magnitude( rotation ) != 0;
	}
	void SetFov( int fov )
	{
		float size = magnitude( topRight - topLeft );
		float distance = size / ( 2.0f * tanf( (float)fov * PI / ( 2.0f * 180.0f ) ) );
		camPos = Center() - Direction() * distance;
		SyncGpuProps();
	}
	void SetAspectRatio( float a )
	{
		float3 mid = ( topLeft + topRight ) / 2;
		float3 dw = normalize( topRight - topLeft ), dh = normalize( topLeft - bottomLeft );
		topLeft = mid - dw * a, topRight = mid + dw * a, bottomLeft = topLeft - dh * 2;
		SyncGpuProps();
	}
	void SyncGpuProps()
	{
		gpuCamProp[0] = float4( camPos, 0 ), gpuCamProp[1] = float4( topLeft, 0 );
		gpuCamProp[2] = float4( topRight, 0 ), gpuCamProp[3] = float4( bottomLeft, 0 );
	}
	int scrwidth, scrheight;
	float aspect;
	float3 camPos, topLeft, topRight, bottomLeft;
	float4 gpuCamProp[4]; // camera.h:25 -- what rtmi_set_camera takes
	bool rotateDirections[4] = { false, false, false, false };
private:
	float3 translation = float3( 0, 0, 0 );
	float3 subTranslation[6] = { float3( 0 ), float3( 0 ), float3( 0 ), float3( 0 ), float3( 0 ), float3( 0 ) };
};

} // namespace rtmi
