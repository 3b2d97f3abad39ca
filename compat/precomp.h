// compat/precomp.h -- what the reference's host sources expect from template/precomp.h, as far as the trace path needs it.
//
// Code written against the reference (`#include "precomp.h"`, namespace Tmpl8, `using namespace Tmpl8`) compiles against this
// directory with `-I compat -I include` and links librtmi_host.so + librtmi.so: the types are the rtmi:: host types
// (raytracer_amd/host/rt_types.h, which keep the reference's field names and derived values bit for bit), the per-object methods
// (Intersect, FindNearest, IsOccluded, GetNormal, GetAlbedo) run on the device behind include/rtmi.h.  There is no CPU trace path
// in here: without a GPU the first such call ends in FatalError.
//   float3 / float4 / mat4 / math   <- template/precomp.h:252-271, 488-491, 822-875, 976-1100
//   MALLOC64 / FREE64               <- template/precomp.h:177-178
//   FatalError                      <- template/precomp.h:479, template.cpp:1099
//   RandomUInt / RandomFloat / Rand <- template.cpp:869-893
//   Surface (pixels, width, height) <- template/surface.h (the part Renderer uses: screen->pixels)
//   SCRWIDTH / SCRHEIGHT            <- template/common.h
#pragma once
#include "../raytracer_amd/host/scene.h"
#include "../include/rtmi.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>

#ifndef SCRWIDTH
#define SCRWIDTH 1280
#define SCRHEIGHT 720
#endif
#define MALLOC64( x ) ( ( x ) == 0 ? 0 : aligned_alloc( 64, ( ( (size_t)( x ) ) + 63 ) & ~(size_t)63 ) )
#define FREE64( x ) free( x )

namespace Tmpl8 {

using rtmi::uint;
using rtmi::float3;
using rtmi::float4;
using rtmi::mat4;
using rtmi::PI;
using rtmi::dot;
using rtmi::cross;
using rtmi::normalize;
using rtmi::magnitude;
using rtmi::sqrf;
using rtmi::TransformPosition;
using rtmi::TransformVector;
using rtmi::Ray;

[[noreturn]] inline void FatalError( const char* fmt, ... )
{
	va_list args;
	va_start( args, fmt );
	vfprintf( stderr, fmt, args );
	va_end( args );
	fputc( '\n', stderr );
	exit( 1 );
}

// Marsaglia's xor32 with the template's start state: the seed table of Renderer::Init comes out as the reference's
inline uint& TemplateSeed() { static uint seed = 0x12345678; return seed; }
inline uint RandomUInt( uint& s ) { s ^= s << 13, s ^= s >> 17, s ^= s << 5; return s; }
inline uint RandomUInt() { return RandomUInt( TemplateSeed() ); }
inline float RandomFloat() { return RandomUInt() * 2.3283064365387e-10f; }
inline float RandomFloat( uint& s ) { return RandomUInt( s ) * 2.3283064365387e-10f; }
inline float Rand( float range ) { return RandomFloat() * range; }

class Surface
{
public:
	Surface( int w, int h ) : pixels( (uint*)MALLOC64( (size_t)w * h * sizeof( uint ) ) ), width( w ), height( h ), owner( true ) { memset( pixels, 0, (size_t)w * h * 4 ); }
	Surface( int w, int h, uint* buffer ) : pixels( buffer ), width( w ), height( h ) {}
	~Surface() { if (owner) FREE64( pixels ); }
	Surface( const Surface& ) = delete;
	Surface& operator=( const Surface& ) = delete;
	void Clear( uint c ) { for (int i = 0; i < width * height; i++) pixels[i] = c; }
	uint* pixels = 0;
	int width = 0, height = 0;
private:
	bool owner = false;
};

// the device the per-object methods of primitive.h run on: one small handle per process, created at the first call
inline rtmi_ctx* CompatDevice()
{
	static rtmi_ctx* ctx = [] {
		rtmi_ctx* c = 0;
		if (rtmi_create( &c, 0, 64, 64 ) != RTMI_OK) FatalError( "compat: no MI355X / librtmi (the per-object methods run on the device)" );
		return c;
	}();
	return ctx;
}

} // namespace Tmpl8

using namespace Tmpl8; // template/precomp.h:168
