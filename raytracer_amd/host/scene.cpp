// scene.cpp -- host scene container (see scene.h).  Build with -O2 -ffp-contract=off.
#include "scene.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace rtmi {

Scene::~Scene() { delete bvh; }

void Scene::InitMaterials() // scene.h:42-48
{
	def_mat = Material( Material::MaterialType::DIFFUSE, float3( 0.93f ), 0 );
	red_mat = Material( Material::MaterialType::DIFFUSE, float3( 0.93f, 0.21f, 0.21f ), 0 );
	cyan_mat = Material( Material::MaterialType::DIFFUSE, float3( 0.11f, 0.95f, 0.91f ), 0 );
	mirror_mat = Material( Material::MaterialType::MIRROR, float3( 0.93f ), 0 );
	absorb_all_but_blue_mat = Material( Material::MaterialType::GLASS, float3( 1 ), 0, 1.52, float3( 8.0f, 2.0f, 1.0f ) );
	glass_mat = Material( Material::MaterialType::GLASS, float3( 0.93f ), 0, 1.52, float3( 0 ) );
	light_mat = Material( Material::MaterialType::LIGHT, float3( 0.93f ), 0, 0, float3( 0 ), float3( 2.0f ) );
}

void Scene::AddQuad( float s, const Material& m, const mat4& T ) { vq.push_back( Quad( 0, s, m, T ) ); }
void Scene::AddSphere( float3 p, float r, const Material& m ) { vs.push_back( Sphere( 0, p, r, m ) ); }
void Scene::AddCube( float3 pos, float3 size, const Material& m, const mat4& T ) { vc.push_back( Cube( 0, pos, size, m, T ) ); }
void Scene::AddPlane( float3 N, float d, const Material& m ) { vp.push_back( Plane( 0, N, d, m ) ); }
void Scene::AddTriangles( const float* v, size_t count, const Material& m, float3 off )
{
	const size_t base = vt.size();
	vt.resize( base + count );
	// no offset: the vertices are taken as they are (x + 0 would turn a -0 coordinate into +0)
	const bool shift = !( off.x == 0 && off.y == 0 && off.z == 0 );
#pragma omp parallel for schedule( static )
	for (long long i = 0; i < (long long)count; i++)
	{
		const float* q = v + 9 * i;
		const float3 a( q[0], q[1], q[2] ), b( q[3], q[4], q[5] ), c( q[6], q[7], q[8] );
		vt[base + i] = shift ? Triangle( 0, a + off, b + off, c + off, m ) : Triangle( 0, a, b, c, m );
	}
}

// scale 1 is the reference room.  The reference's triangle test rejects |det| < 1e-4 (primitive.h:59), i.e. every
// triangle with 2 * area * |cos(angle to the ray)| < 1e-4: a 10^6-triangle mesh that fits the reference room
// (three objects of ~3 m^2 each) is invisible to every ray.  The mesh configs therefore use the SAME scene in larger
// units: room, light, objects and camera multiplied by `scale` (solid angles, hence the image, are unchanged).
void Scene::MakeRoom( float scale )
{
	InitMaterials();
	const float k = scale;
	// scene.h:51-52, 58: light quad; scene.h:70-75: the six walls
	mat4 M1base = mat4::Translate( float3( 0, 2.6f * k, 2 * k ) );
	mat4 M1 = M1base * mat4::RotateZ( sinf( 0 * 0.6f ) * 0.1f ) * mat4::Translate( float3( 0, -0.9 * k, 0 ) );
	AddQuad( 3 * k, light_mat, M1 );
	AddPlane( float3( 1, 0, 0 ), 3 * k, mirror_mat );
	AddPlane( float3( -1, 0, 0 ), 2.99f * k, def_mat );
	AddPlane( float3( 0, 1, 0 ), 1 * k, def_mat );
	AddPlane( float3( 0, -1, 0 ), 2 * k, def_mat );
	AddPlane( float3( 0, 0, 1 ), 3 * k, def_mat );
	AddPlane( float3( 0, 0, -1 ), 3.99f * k, def_mat );
}

bool Scene::MakeDefault( const std::string& assets, BVH::Builder builder )
{
	MakeRoom();
	// scene.h:61-63
	AddSphere( float3( -1.4f, -0.5f, 2 ), 0.5f, absorb_all_but_blue_mat );
	AddSphere( float3( 0, 2.5f, -3.07f ), 0.5f, def_mat );
	// scene.h:54-55, 66
	mat4 M2base = mat4::RotateX( PI / 4 ) * mat4::RotateZ( PI / 4 );
	mat4 M2 = mat4::Translate( float3( 1.4f, 0, 2 ) ) * mat4::RotateY( 0 * 0.5f ) * M2base;
	AddCube( float3( 0 ), float3( 0.75f ), def_mat, M2 );
	// scene.h:80-82
	if (!LoadObject( assets + "/monkey.obj", glass_mat, float3( 0, 0, 1.5 ) )) return false;
	if (!LoadObject( assets + "/monkey.obj", cyan_mat, float3( 1.5, 0, 1.5 ) )) return false;
	if (!LoadObject( assets + "/monkey.obj", red_mat, float3( -1.5, 0, 1.5 ) )) return false;
	if (!LoadTextureRaw( assets + "/logo_128x64.rgb32", 128, 64 )) return false;
	Finalize( builder );
	return true;
}

void Scene::SetTexture( const uint32_t* px, int w, int h ) { texture.assign( px, px + (size_t)w * h ), tex_w = w, tex_h = h; }
bool Scene::LoadTextureRaw( const std::string& file, int w, int h )
{
	std::ifstream f( file, std::ios::binary );
	if (!f) return false;
	texture.resize( (size_t)w * h );
	f.read( (char*)texture.data(), (std::streamsize)texture.size() * 4 );
	tex_w = w, tex_h = h;
	return (size_t)f.gcount() == texture.size() * 4;
}

// Decimal text -> double the way tinyobjloader v2.0.0 does it (tiny_obj_loader.h:887-1017, the parser the
// reference's LoadObject goes through): digits are accumulated into a double mantissa, fractional digits are
// added as digit * 10^-k from a small table, and a decimal exponent is applied as ldexp(m * 5^e, e).  Restated
// (not strtod) so that vertices come out bit-identical to the reference's.
static bool ParseReal( const char*& s, float& out )
{
	while (*s == ' ' || *s == '\t') s++;
	const char* c = s;
	double m = 0.0;
	int exponent = 0, read = 0;
	char sign = '+', esign = '+';
	bool leadingDot = false;
	if (*c == '+' || *c == '-') { sign = *c++; if (*c == '.') leadingDot = true; }
	else if (*c >= '0' && *c <= '9') {}
	else if (*c == '.') leadingDot = true;
	else return false;
	if (!leadingDot)
	{
		while (*c >= '0' && *c <= '9') m *= 10, m += (int)( *c - '0' ), c++, read++;
		if (read == 0) return false;
	}
	if (*c == '.')
	{
		static const double lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
		c++, read = 1;
		while (*c >= '0' && *c <= '9') m += (int)( *c - '0' ) * ( read < 8 ? lut[read] : std::pow( 10.0, -read ) ), read++, c++;
	}
	if (*c == 'e' || *c == 'E')
	{
		c++;
		if (*c == '+' || *c == '-') esign = *c++;
		else if (!( *c >= '0' && *c <= '9' )) return false;
		read = 0;
		while (*c >= '0' && *c <= '9') { if (exponent > 214748364) return false; exponent = exponent * 10 + ( *c - '0' ), c++, read++; }
		exponent *= ( esign == '+' ? 1 : -1 );
		if (read == 0) return false;
	}
	const double r = ( sign == '+' ? 1 : -1 ) * ( exponent ? std::ldexp( m * std::pow( 5.0, exponent ), exponent ) : m );
	out = (float)r;
	s = c;
	return true;
}

// ---- OBJ ingest (Scene::LoadObject, scene.h:222-291, which goes through tinyobjloader v2.0.0) ------------------------------
// The whole file is read into memory and cut into chunks at line ends, one per thread.  Pass 1 counts, per chunk, the
// vertices, the triangles its faces produce and its `usemtl` statements; prefix sums give every chunk its place in the
// global arrays (file order is kept: scene.h:246-268 walks shapes and faces in file order) and the material in force at its
// first line.  Pass 2a parses the vertices, pass 2b the faces.  Decimal text -> float is ParseReal above (tinyobj's own
// algorithm), a quad is split along its shorter diagonal as tinyobj does (tiny_obj_loader.h:1485-1580); polygons with more
// than four corners are fanned (tinyobj ear-clips them; the reference's assets hold triangles only).
namespace {
struct ObjChunk { const char* b, *e; size_t nv = 0, ntri = 0, v0 = 0, t0 = 0; std::vector<std::pair<size_t, std::string>> mtl; std::string mtlAtStart; };
inline const char* line_end( const char* p, const char* e ) { while (p < e && *p != '\n') p++; return p; }
inline const char* skip_ws( const char* s ) { while (*s == ' ' || *s == '\t') s++; return s; }
inline bool is_ws_or_end( char c ) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == 0; }
// the vertex index of one face corner "v", "v/vt", "v/vt/vn", "v//vn" (1-based; negative = relative to the vertices read so far)
inline bool corner_index( const char*& s, const char* e, long long nvBefore, long long& out )
{
	s = skip_ws( s );
	if (s >= e || *s == '\r' || *s == '\n') return false;
	char* end;
	const long i = strtol( s, &end, 10 );
	if (end == s) return false;
	out = i > 0 ? i - 1 : nvBefore + i;
	s = end;
	while (s < e && !is_ws_or_end( *s )) s++; // skip /vt/vn
	return true;
}
std::string token_after( const char* s, const char* e )
{
	s = skip_ws( s );
	const char* t = s;
	while (t < e && *t != '\r' && *t != '\n') t++;
	while (t > s && ( t[-1] == ' ' || t[-1] == '\t' )) t--;
	return std::string( s, t );
}
} // namespace

// Wavefront .mtl -> Material (no reference counterpart: the reference parses the .mtl through tinyobj and then ignores it,
// scene.h:266).  Kd -> color, Ke -> emission, Ni -> refractive index, Tf -> Beer absorption -ln(Tf) per unit length.
// Type: LIGHT if Ke > 0; else GLASS for illum 4, 6, 7, 9 or d / Tr making it transparent; else MIRROR for illum 3, 5, 8;
// else DIFFUSE.
bool Scene::LoadMaterialLibrary( const std::string& file, std::vector<std::pair<std::string, Material>>& out )
{
	std::ifstream f( file );
	if (!f) return false;
	std::string line, name;
	Material m;
	int illum = 2;
	float d = 1;
	bool have = false;
	auto flush = [&]() {
		if (!have) return;
		const bool emits = m.emission.x > 0 || m.emission.y > 0 || m.emission.z > 0;
		m.type = emits ? Material::MaterialType::LIGHT : ( illum == 4 || illum == 6 || illum == 7 || illum == 9 || d < 1 ) ? Material::MaterialType::GLASS
			: ( illum == 3 || illum == 5 || illum == 8 ) ? Material::MaterialType::MIRROR : Material::MaterialType::DIFFUSE;
		out.push_back( { name, m } ); };
	while (std::getline( f, line ))
	{
		const char* s = skip_ws( line.c_str() ), *e = line.c_str() + line.size();
		auto three = [&]( const char* p, float3& v ) { float x = 0, y = 0, z = 0; if (ParseReal( p, x )) { y = z = x; if (ParseReal( p, y )) ParseReal( p, z ); else y = x; } v = float3( x, y, z ); };
		if (!strncmp( s, "newmtl", 6 ) && is_ws_or_end( s[6] )) { flush(); name = token_after( s + 6, e ), m = Material(), m.color = float3( 0.93f ), illum = 2, d = 1, have = true; }
		else if (!strncmp( s, "Kd", 2 ) && is_ws_or_end( s[2] )) three( s + 2, m.color );
		else if (!strncmp( s, "Ke", 2 ) && is_ws_or_end( s[2] )) three( s + 2, m.emission );
		else if (!strncmp( s, "Tf", 2 ) && is_ws_or_end( s[2] )) { float3 tf; three( s + 2, tf ); m.absorption = float3( -logf( tmax( tf.x, 1e-6f ) ), -logf( tmax( tf.y, 1e-6f ) ), -logf( tmax( tf.z, 1e-6f ) ) ); }
		else if (!strncmp( s, "Ni", 2 ) && is_ws_or_end( s[2] )) { const char* p = s + 2; ParseReal( p, m.refractive_index ); }
		else if (!strncmp( s, "illum", 5 ) && is_ws_or_end( s[5] )) illum = atoi( s + 5 );
		else if (s[0] == 'd' && is_ws_or_end( s[1] )) { const char* p = s + 1; ParseReal( p, d ); }
		else if (!strncmp( s, "Tr", 2 ) && is_ws_or_end( s[2] )) { const char* p = s + 2; float tr = 0; if (ParseReal( p, tr )) d = 1 - tr; }
	}
	flush();
	return true;
}

bool Scene::LoadObject( const std::string& file, const Material& material, float3 transform ) { return LoadObjectImpl( file, material, transform, false ); }
bool Scene::LoadObjectWithMaterials( const std::string& file, const Material& fallback, float3 transform ) { return LoadObjectImpl( file, fallback, transform, true ); }

bool Scene::LoadObjectImpl( const std::string& file, const Material& material, float3 transform, bool useMtl )
{
	std::string text;
	{
		FILE* f = fopen( file.c_str(), "rb" );
		if (!f) { fprintf( stderr, "rtmi::Scene::LoadObject: cannot open %s\n", file.c_str() ); return false; }
		fseek( f, 0, SEEK_END );
		const long size = ftell( f );
		fseek( f, 0, SEEK_SET );
		text.resize( size > 0 ? (size_t)size : 0 );
		const size_t got = text.empty() ? 0 : fread( &text[0], 1, text.size(), f );
		fclose( f );
		if (got != text.size()) return false;
	}
	const char* B = text.c_str(), *E = B + text.size();
	int nth = 1;
#ifdef _OPENMP
	nth = std::max( 1, std::min( omp_get_max_threads(), (int)( text.size() / ( 1u << 20 ) ) + 1 ) );
#endif
	std::vector<ObjChunk> ch( nth );
	for (int k = 0; k < nth; k++)
	{
		const char* b = k == 0 ? B : ch[k - 1].e;
		const char* e = k == nth - 1 ? E : line_end( std::max( b, B + text.size() * (size_t)( k + 1 ) / nth ), E );
		if (e < E) e++; // past the newline
		ch[k].b = b, ch[k].e = e;
	}
	std::vector<std::string> mtllibs;
	// pass 1: counts
#pragma omp parallel for schedule( static, 1 ) num_threads( nth )
	for (int k = 0; k < nth; k++)
	{
		ObjChunk& c = ch[k];
		for (const char* p = c.b; p < c.e; )
		{
			const char* le = line_end( p, c.e );
			const char* s = skip_ws( p );
			if (s[0] == 'v' && ( s[1] == ' ' || s[1] == '\t' )) c.nv++;
			else if (s[0] == 'f' && ( s[1] == ' ' || s[1] == '\t' ))
			{
				int corners = 0;
				long long dummy;
				for (const char* q = s + 2; corner_index( q, le, 0, dummy ); ) corners++;
				if (corners >= 3) c.ntri += (size_t)( corners - 2 );
			}
			else if (useMtl && !strncmp( s, "usemtl", 6 ) && is_ws_or_end( s[6] )) c.mtl.push_back( { c.ntri, token_after( s + 6, le ) } );
			p = le + 1;
		}
	}
	if (useMtl) // (mtllib lines are rare: one sequential scan for them)
		for (const char* p = B; p < E; )
		{
			const char* le = line_end( p, E );
			const char* s = skip_ws( p );
			if (!strncmp( s, "mtllib", 6 ) && is_ws_or_end( s[6] )) mtllibs.push_back( token_after( s + 6, le ) );
			p = le + 1;
		}
	size_t nv = 0, ntri = 0;
	std::string cur;
	for (int k = 0; k < nth; k++)
	{
		ch[k].v0 = nv, ch[k].t0 = ntri, ch[k].mtlAtStart = cur;
		nv += ch[k].nv, ntri += ch[k].ntri;
		if (!ch[k].mtl.empty()) cur = ch[k].mtl.back().second;
	}
	if (ntri == 0) return false;
	// materials by name
	std::vector<std::pair<std::string, Material>> lib;
	if (useMtl)
	{
		const size_t slash = file.find_last_of( '/' );
		const std::string dir = slash == std::string::npos ? "" : file.substr( 0, slash + 1 );
		for (const std::string& m : mtllibs) LoadMaterialLibrary( dir + m, lib );
	}
	auto material_named = [&]( const std::string& name ) -> const Material& {
		for (const auto& m : lib) if (m.first == name) return m.second;
		return material; };
	// pass 2a: vertices
	std::vector<float> v( nv * 3 );
#pragma omp parallel for schedule( static, 1 ) num_threads( nth )
	for (int k = 0; k < nth; k++)
	{
		size_t at = ch[k].v0;
		for (const char* p = ch[k].b; p < ch[k].e; )
		{
			const char* le = line_end( p, ch[k].e );
			const char* s = skip_ws( p );
			if (s[0] == 'v' && ( s[1] == ' ' || s[1] == '\t' ))
			{
				s += 2;
				float x = 0, y = 0, z = 0;
				ParseReal( s, x ), ParseReal( s, y ), ParseReal( s, z );
				v[3 * at] = x, v[3 * at + 1] = y, v[3 * at + 2] = z, at++;
			}
			p = le + 1;
		}
	}
	// pass 2b: faces -> triangles, in file order
	const size_t before = vt.size();
	vt.resize( before + ntri );
	// (Differences from tinyobj kept on purpose: a face with an index outside the file fails the whole load here -- tinyobj drops that
	// face with a warning, tiny_obj_loader.h:1497-1503 -- and indices are checked against the file's total vertex count, so a face may
	// name a vertex that is defined further down; the reference's own assets and the generated meshes do neither.)
	bool bad = false;
#pragma omp parallel for schedule( static, 1 ) num_threads( nth ) reduction( || : bad )
	for (int k = 0; k < nth; k++)
	{
		size_t at = before + ch[k].t0, seenV = ch[k].v0, nextMtl = 0;
		const Material* mat = useMtl && !ch[k].mtlAtStart.empty() ? &material_named( ch[k].mtlAtStart ) : &material;
		std::vector<long long> idx;
		for (const char* p = ch[k].b; p < ch[k].e; )
		{
			const char* le = line_end( p, ch[k].e );
			const char* s = skip_ws( p );
			if (s[0] == 'v' && ( s[1] == ' ' || s[1] == '\t' )) seenV++;
			else if (s[0] == 'f' && ( s[1] == ' ' || s[1] == '\t' ))
			{
				while (nextMtl < ch[k].mtl.size() && ch[k].mtl[nextMtl].first <= at - before - ch[k].t0) mat = &material_named( ch[k].mtl[nextMtl++].second );
				idx.clear();
				long long i;
				for (const char* q = s + 2; corner_index( q, le, (long long)seenV, i ); ) idx.push_back( i );
				auto P = [&]( long long a ) { return float3( v[3 * a], v[3 * a + 1], v[3 * a + 2] ); };
				auto ok = [&]( long long a ) { return a >= 0 && (size_t)a < nv; };
				auto emit = [&]( long long a, long long b, long long c ) {
					if (!ok( a ) || !ok( b ) || !ok( c )) { bad = true; a = b = c = 0; }
					vt[at++] = Triangle( 0, P( a ) + transform, P( b ) + transform, P( c ) + transform, *mat ); }; // scene.h:266
				if (idx.size() == 4 && ok( idx[0] ) && ok( idx[1] ) && ok( idx[2] ) && ok( idx[3] ))
				{
					// tinyobj: the shorter diagonal splits the quad (tiny_obj_loader.h:1537-1580), binary32 arithmetic
					const float3 e02 = P( idx[2] ) - P( idx[0] ), e13 = P( idx[3] ) - P( idx[1] );
					const float sqr02 = e02.x * e02.x + e02.y * e02.y + e02.z * e02.z, sqr13 = e13.x * e13.x + e13.y * e13.y + e13.z * e13.z;
					if (sqr02 < sqr13) emit( idx[0], idx[1], idx[2] ), emit( idx[0], idx[2], idx[3] );
					else emit( idx[0], idx[1], idx[3] ), emit( idx[1], idx[2], idx[3] );
				}
				else for (size_t c = 2; c < idx.size(); c++) emit( idx[0], idx[c - 1], idx[c] );
			}
			p = le + 1;
		}
	}
	if (bad) { vt.resize( before ); return false; }
	return true;
}

void Scene::Finalize( BVH::Builder builder )
{
	quads = vq.data(), quads_size = (int)vq.size();
	spheres = vs.data(), spheres_size = (int)vs.size();
	cubes = vc.data(), cubes_size = (int)vc.size();
	planes = vp.data(), planes_size = (int)vp.size();
	triangles = vt.data(), triangles_size = (int)vt.size();
	// global object indices in class order (scene.h:58-75, 266-270)
	id = 0;
	for (auto& q : vq) q.objIdx = id++;
	for (auto& s : vs) s.objIdx = id++;
	for (auto& c : vc) c.objIdx = id++;
	for (auto& p : vp) p.objIdx = id++;
	{
		const int first = id;
#pragma omp parallel for schedule( static )
		for (long long k = 0; k < (long long)vt.size(); k++) vt[k].objIdx = first + (int)k;
		id += (int)vt.size();
	}
	delete bvh;
	bvh = new BVH( spheres, spheres_size, planes, planes_size, triangles, triangles_size, builder );

	const int total = TotalPrimitives();
	albedos.assign( total, float4( 0 ) ), absorptions.assign( total, float4( 0 ) ), primitives.assign( total, float4( 0 ) );
	primMaterials.assign( total, 0 ), refractiveIndices.assign( total, 0 );
	quadMatrices.resize( quads_size ), quadSizes.resize( quads_size );
	sphereInfos.resize( spheres_size ), sphereInvrs.resize( spheres_size );
	cubeMatrices.resize( cubes_size ), cubeInvMatrices.resize( cubes_size ), cubeBs.resize( (size_t)cubes_size * 2 );
	triangleInfos.resize( (size_t)triangles_size * 3 );
	int i = 0;
	auto mat = [&]( int k, const Material& m ) {
		primMaterials[k] = (int)m.type, albedos[k] = float4( m.color, 0 );
		refractiveIndices[k] = m.refractive_index, absorptions[k] = float4( m.absorption, 0 ); };
	for (int k = 0; k < quads_size; k++, i++) // scene.h:116-124
		mat( i, quads[k].material ), quadMatrices[k] = quads[k].invT, quadSizes[k] = quads[k].size, primitives[i] = float4( quads[k].N, 0 );
	for (int k = 0; k < spheres_size; k++, i++) // scene.h:131-139
		mat( i, spheres[k].material ), sphereInfos[k] = float4( spheres[k].pos, spheres[k].r2 ), primitives[i] = float4( spheres[k].pos, 0 ), sphereInvrs[k] = spheres[k].invr;
	for (int k = 0; k < cubes_size; k++, i++) // scene.h:147-157
		mat( i, cubes[k].material ), cubeInvMatrices[k] = cubes[k].invM, cubeMatrices[k] = cubes[k].M,
		cubeBs[2 * k] = float4( cubes[k].b[0], 0 ), cubeBs[2 * k + 1] = float4( cubes[k].b[1], 0 ), primitives[i] = float4( 0 );
	for (int k = 0; k < planes_size; k++, i++) // scene.h:165-171
		mat( i, planes[k].material ), primitives[i] = float4( planes[k].N, planes[k].d );
	const int triBase = i;
#pragma omp parallel for schedule( static )
	for (int k = 0; k < triangles_size; k++) // scene.h:176-185
	{
		mat( triBase + k, triangles[k].material );
		triangleInfos[3 * (size_t)k] = float4( triangles[k].pos1, 0.0f ), triangleInfos[3 * (size_t)k + 1] = float4( triangles[k].pos2, 0.0f );
		triangleInfos[3 * (size_t)k + 2] = float4( triangles[k].pos3, 0.0f ), primitives[triBase + k] = float4( triangles[k].N, 0 );
	}

	desc = rtmi_scene_desc{};
	desc.n_quads = quads_size, desc.n_spheres = spheres_size, desc.n_cubes = cubes_size, desc.n_planes = planes_size, desc.n_triangles = triangles_size;
	desc.quad_inv_T = quads_size ? quadMatrices[0].cell : 0, desc.quad_size = quadSizes.data();
	desc.sphere_info = (const float*)sphereInfos.data(), desc.sphere_invr = sphereInvrs.data();
	desc.cube_M = cubes_size ? cubeMatrices[0].cell : 0, desc.cube_inv_M = cubes_size ? cubeInvMatrices[0].cell : 0, desc.cube_b = (const float*)cubeBs.data();
	desc.tri_verts = (const float*)triangleInfos.data();
	desc.primitives = (const float*)primitives.data(), desc.albedo = (const float*)albedos.data();
	desc.material_type = primMaterials.data(), desc.ior = refractiveIndices.data(), desc.absorption = (const float*)absorptions.data();
	desc.nodes = (const rtmi_bvh_node*)bvh->gpuBvhNode, desc.n_nodes = (int)bvh->nodesUsed, desc.prim_idx = bvh->primitiveIdx;
	if (quads_size)
	{
		const Quad& q = quads[0]; // scene.h:208, renderer.cpp:483
		const float4 l[3] = { float4( q.c1, 0 ), float4( q.c2, 0 ), float4( q.c3, 0 ) };
		memcpy( desc.light_corners, l, 48 );
		desc.light_A = q.A, desc.light_s = q.s;
		desc.light_emission[0] = q.material.emission.x, desc.light_emission[1] = q.material.emission.y, desc.light_emission[2] = q.material.emission.z, desc.light_emission[3] = 0;
		// Scene::GetLightPos (scene.h:343-350), by value instead of the reference's dangling pointer
		const float3 corner1 = TransformPosition( float3( -q.size, 0, -q.size ), q.T );
		const float3 corner2 = TransformPosition( float3( q.size, 0, q.size ), q.T );
		const float3 l0 = ( corner1 + corner2 ) * 0.5f - float3( 0, 0.01f, 0 );
		desc.whitted_light_pos[0][0] = l0.x, desc.whitted_light_pos[0][1] = l0.y, desc.whitted_light_pos[0][2] = l0.z;
	}
	MarshalLights();
	desc.texture = texture.empty() ? 0 : texture.data(), desc.tex_w = tex_w, desc.tex_h = tex_h;
	isInitialized = true;
}

// Scene::GetLightPos()[1] / GetLightColor() / intensity (scene.h:343-358, 562-564) into the descriptor.  The reference's setters
// (scene.h:536-551) are read by DirectIllumination at the next Trace; here Renderer::Tick hands a changed set to the device
// (rtmi_set_whitted_lights) before the next frame, without a scene upload.
void Scene::MarshalLights()
{
	desc.whitted_light_pos[1][0] = lightPos.x, desc.whitted_light_pos[1][1] = lightPos.y, desc.whitted_light_pos[1][2] = lightPos.z;
	for (int k = 0; k < 2; k++)
		desc.whitted_light_color[k][0] = lightColors[k].x, desc.whitted_light_color[k][1] = lightColors[k].y, desc.whitted_light_color[k][2] = lightColors[k].z;
	desc.whitted_intensity = intensity;
	lightsChanged = true;
}

// (p,q) torus knot centre line c(t) = ((R + r cos qt) cos pt, r sin qt, (R + r cos qt) sin pt), swept by a circle of
// radius `tube` in the (normal, binormal) frame obtained from finite differences.  Evaluated in double, stored as
// float; closed form, no RNG, so every rank generates the identical mesh.
void MakeTorusKnot( int nu, int nv, float R, float r, float tube, int p, int q, float3 centre, std::vector<float>& out )
{
	const double TWO_PI = 6.283185307179586476925;
	auto curve = [&]( double t, double* c ) {
		const double w = R + r * cos( q * t );
		c[0] = w * cos( p * t ), c[1] = r * sin( q * t ), c[2] = w * sin( p * t ); };
	std::vector<float> P( (size_t)nu * nv * 3 );
#pragma omp parallel for schedule( static )
	for (int i = 0; i < nu; i++)
	{
		const double t = TWO_PI * i / nu, h = 1e-4;
		double c[3], a[3], b[3];
		curve( t, c ), curve( t - h, a ), curve( t + h, b );
		double T[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
		double A[3] = { b[0] - 2 * c[0] + a[0], b[1] - 2 * c[1] + a[1], b[2] - 2 * c[2] + a[2] };
		double tl = sqrt( T[0] * T[0] + T[1] * T[1] + T[2] * T[2] );
		for (int k = 0; k < 3; k++) T[k] /= tl;
		double ad = A[0] * T[0] + A[1] * T[1] + A[2] * T[2];
		double N[3] = { A[0] - ad * T[0], A[1] - ad * T[1], A[2] - ad * T[2] };
		double nl = sqrt( N[0] * N[0] + N[1] * N[1] + N[2] * N[2] );
		if (nl < 1e-12) { N[0] = 0, N[1] = 1, N[2] = 0, nl = 1; }
		for (int k = 0; k < 3; k++) N[k] /= nl;
		double B[3] = { T[1] * N[2] - T[2] * N[1], T[2] * N[0] - T[0] * N[2], T[0] * N[1] - T[1] * N[0] };
		for (int j = 0; j < nv; j++)
		{
			const double s = TWO_PI * j / nv, cs = cos( s ) * tube, sn = sin( s ) * tube;
			float* o = &P[( (size_t)i * nv + j ) * 3];
			o[0] = (float)( c[0] + cs * N[0] + sn * B[0] + centre.x );
			o[1] = (float)( c[1] + cs * N[1] + sn * B[1] + centre.y );
			o[2] = (float)( c[2] + cs * N[2] + sn * B[2] + centre.z );
		}
	}
	out.resize( (size_t)nu * nv * 18 );
#pragma omp parallel for schedule( static )
	for (int i = 0; i < nu; i++) for (int j = 0; j < nv; j++)
	{
		const int i1 = ( i + 1 ) % nu, j1 = ( j + 1 ) % nv;
		const float* p00 = &P[( (size_t)i * nv + j ) * 3], *p10 = &P[( (size_t)i1 * nv + j ) * 3];
		const float* p01 = &P[( (size_t)i * nv + j1 ) * 3], *p11 = &P[( (size_t)i1 * nv + j1 ) * 3];
		float* o = &out[( (size_t)i * nv + j ) * 18];
		memcpy( o, p00, 12 ), memcpy( o + 3, p10, 12 ), memcpy( o + 6, p11, 12 );
		memcpy( o + 9, p00, 12 ), memcpy( o + 12, p11, 12 ), memcpy( o + 15, p01, 12 );
	}
}

bool WriteObj( const std::string& file, const float* v, size_t ntri )
{
	FILE* f = fopen( file.c_str(), "w" );
	if (!f) return false;
	fprintf( f, "# rtmi synthetic mesh, %zu triangles\n", ntri );
	for (size_t i = 0; i < ntri * 3; i++) fprintf( f, "v %.9g %.9g %.9g\n", v[3 * i], v[3 * i + 1], v[3 * i + 2] );
	for (size_t i = 0; i < ntri; i++) fprintf( f, "f %zu %zu %zu\n", 3 * i + 1, 3 * i + 2, 3 * i + 3 );
	fclose( f );
	return true;
}

} // namespace rtmi
