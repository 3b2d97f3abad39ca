// capi.cpp -- plain-C handles over the C++ host mirror (Scene, BVH, Camera, Renderer) for the Python harness
// (tests/, bench.py, __graft_entry__.py) via ctypes.  No arithmetic of its own.
#include "renderer.h"
#include "frames.h"
#include <chrono>
#include <cstdio>

using namespace rtmi;

extern "C" {

struct rth_material { int type; float color[3]; float ior; float absorption[3]; float emission[3]; };

static Material to_mat( const rth_material* m )
{
	return Material( (Material::MaterialType)m->type, float3( m->color[0], m->color[1], m->color[2] ), 0, m->ior,
		float3( m->absorption[0], m->absorption[1], m->absorption[2] ), float3( m->emission[0], m->emission[1], m->emission[2] ) );
}

void* rth_scene_create() { return new Scene(); }
void rth_scene_destroy( void* s ) { delete (Scene*)s; }
int rth_scene_make_default( void* s, const char* assets, int builder ) { return ( (Scene*)s )->MakeDefault( assets, (BVH::Builder)builder ) ? 0 : -1; }
void rth_scene_make_room( void* s, float scale ) { ( (Scene*)s )->MakeRoom( scale ); }
void rth_scene_add_sphere( void* s, float x, float y, float z, float r, const rth_material* m ) { ( (Scene*)s )->AddSphere( float3( x, y, z ), r, to_mat( m ) ); }
void rth_scene_add_triangles( void* s, const float* v9, long long n, const rth_material* m, float ox, float oy, float oz )
{
	( (Scene*)s )->AddTriangles( v9, (size_t)n, to_mat( m ), float3( ox, oy, oz ) );
}
void rth_scene_add_torus_knot( void* s, int nu, int nv, float R, float r, float tube, int p, int q, float cx, float cy, float cz, const rth_material* m )
{
	std::vector<float> v;
	MakeTorusKnot( nu, nv, R, r, tube, p, q, float3( cx, cy, cz ), v );
	( (Scene*)s )->AddTriangles( v.data(), v.size() / 9, to_mat( m ) );
}
void rth_scene_add_plane( void* s, float nx, float ny, float nz, float d, const rth_material* m ) { ( (Scene*)s )->AddPlane( float3( nx, ny, nz ), d, to_mat( m ) ); }
void rth_scene_add_quad( void* s, float size, const float* T16, const rth_material* m )
{
	mat4 T;
	memcpy( T.cell, T16, 64 );
	( (Scene*)s )->AddQuad( size, to_mat( m ), T );
}
void rth_scene_add_cube( void* s, const float* pos3, const float* size3, const float* M16, const rth_material* m )
{
	mat4 M;
	memcpy( M.cell, M16, 64 );
	( (Scene*)s )->AddCube( float3( pos3[0], pos3[1], pos3[2] ), float3( size3[0], size3[1], size3[2] ), to_mat( m ), M );
}
int rth_scene_load_obj( void* s, const char* path, const rth_material* m, float ox, float oy, float oz )
{
	return ( (Scene*)s )->LoadObject( path, to_mat( m ), float3( ox, oy, oz ) ) ? 0 : -1;
}
int rth_scene_load_obj_mtl( void* s, const char* path, const rth_material* fallback, float ox, float oy, float oz )
{
	return ( (Scene*)s )->LoadObjectWithMaterials( path, to_mat( fallback ), float3( ox, oy, oz ) ) ? 0 : -1;
}
int rth_scene_load_texture_raw( void* s, const char* path, int w, int h ) { return ( (Scene*)s )->LoadTextureRaw( path, w, h ) ? 0 : -1; }
// Scene::SetLightIntensity / SetLightPos / SetLightColor / SetRoofLightColor (scene.h:536-551)
void rth_scene_set_lights( void* s, float intensity, const float* pos3, const float* color3, const float* roof3 )
{
	Scene* sc = (Scene*)s;
	sc->SetLightIntensity( intensity );
	if (pos3) sc->SetLightPos( float3( pos3[0], pos3[1], pos3[2] ) );
	if (color3) sc->SetLightColor( float3( color3[0], color3[1], color3[2] ) );
	if (roof3) sc->SetRoofLightColor( float3( roof3[0], roof3[1], roof3[2] ) );
}
// BVH::BuildQBVH on the scene's tree (bvh.h:213-249): the number of QBVH nodes (0: the primitive root is a leaf), and, if `out` is
// given, the nodes as 32 words each (SIMD_BVH_Node, bvh.h:18-27)
int rth_scene_build_qbvh( void* s, void* out, int capacity )
{
	Scene* sc = (Scene*)s;
	if (!sc->bvh || !sc->bvh->BuildQBVH()) return 0;
	const int n = (int)sc->bvh->nodesUsedQBVH;
	if (out && capacity >= n) memcpy( out, sc->bvh->qbvhNode.data(), (size_t)n * sizeof( QbvhNode ) );
	return n;
}
void rth_scene_finalize( void* s, int builder ) { ( (Scene*)s )->Finalize( (BVH::Builder)builder ); }
const rtmi_scene_desc* rth_scene_desc( void* s ) { return ( (Scene*)s )->Desc(); }
void rth_scene_info( void* s, int* out )
{
	Scene* sc = (Scene*)s;
	out[0] = sc->quads_size, out[1] = sc->spheres_size, out[2] = sc->cubes_size, out[3] = sc->planes_size, out[4] = sc->triangles_size;
	out[5] = sc->bvh ? (int)sc->bvh->nodesUsed : 0, out[6] = sc->bvh ? sc->bvh->maxDepth : 0;
}
void rth_quad_T( void* s, float* T16 ) { memcpy( T16, ( (Scene*)s )->quads[0].T.cell, 64 ); }

// The strong-scaling frame loop of a C++ host (frames.h): `frames` frames of total_spp samples per pixel each, this
// rank's share of every frame (global sample indices frame * total_spp + rank + k * world) rendered on handles[frame % n_handles].
// Per frame, the calls of the unpipelined loop: rtmi_reset_accumulator, rtmi_render, rtmi_flush; then, in frame order on every
// rank, rtmi_reduce_accumulator( comm, 0 ) if comm != 0; then on rank 0 rtmi_finalize( finalize_spp ) into pixels_out + frame * W * H
// (pixels_out may be 0: the frame is finished and dropped).  Returns the first error status of any frame.
int rth_render_frames( rtmi_ctx** handles, int n_handles, int frames, int total_spp, int rank, int world, void* comm, int finalize_spp, int W, int H, uint32_t* pixels_out )
{
	if (!handles || n_handles < 1 || frames < 0 || total_spp < 1 || world < 1 || rank < 0 || rank >= world) return RTMI_ERR_INVALID;
	FramePipeline pipe( n_handles );
	const int mine = total_spp > rank ? ( total_spp - rank + world - 1 ) / world : 0;
	for (int f = 0; f < frames; f++)
		pipe.Submit( [&, mine]( int k, uint64_t frame ) -> int {
			rtmi_ctx* c = handles[k];
			int r = rtmi_reset_accumulator( c );
			if (!r && mine > 0) r = rtmi_render( c, RTMI_PATHTRACING, (int)frame * total_spp + rank, mine, world, 5, 0 );
			if (!r) r = rtmi_flush( c ); // the end of the frame is enqueued: it runs beside the next frames' bulk
			if (!r && comm) { FramePipeline::Ordered turn( pipe, frame ); r = rtmi_reduce_accumulator( c, comm, 0 ); }
			if (!r && rank == 0) r = pixels_out ? rtmi_finalize( c, pixels_out + frame * (uint64_t)W * H, finalize_spp, nullptr ) : rtmi_synchronize( c );
			return r;
		} );
	return pipe.Finish();
}

// frames.h without a GPU (tests/test_sample_split.py): `frames` jobs on `handles` threads that reach their collective after delays
// out of frame order; jobs whose bit in skip_mask is set have no collective, the one numbered `fail_at` returns 7 before it.
// order_out receives the frames in the order in which their ordered sections ran; returns Finish()'s status, or -100 - k if handle k
// was ever given two frames at once or a frame that is not its own.
int rth_frames_selftest( int handles, int frames, unsigned skip_mask, int fail_at, int* order_out, int* n_order )
{
	FramePipeline pipe( handles );
	std::mutex m;
	std::vector<int> busy( handles, 0 ), order;
	int bad = 0;
	for (int f = 0; f < frames; f++)
		pipe.Submit( [&]( int k, uint64_t frame ) -> int {
			{ std::lock_guard<std::mutex> lock( m ); if (busy[k]++ || (int)( frame % handles ) != k) bad = -100 - k; }
			std::this_thread::sleep_for( std::chrono::milliseconds( 3 * ( ( 7 * frame ) % 4 ) ) );
			int r = 0;
			if ((int)frame == fail_at) r = 7;
			else if (!( skip_mask >> frame & 1u ))
			{
				FramePipeline::Ordered turn( pipe, frame );
				std::lock_guard<std::mutex> lock( m );
				order.push_back( (int)frame );
			}
			{ std::lock_guard<std::mutex> lock( m ); busy[k]--; }
			return r;
		} );
	const int status = pipe.Finish();
	*n_order = (int)order.size();
	for (size_t i = 0; i < order.size(); i++) order_out[i] = order[i];
	return bad ? bad : status;
}

void rth_fill_seeds( uint32_t* out, long long n ) { FillSeedTable( out, (size_t)n ); }
void rth_camera_default( int w, int h, float* cam16 ) { Camera c( w, h ); memcpy( cam16, c.gpuCamProp, 64 ); }
void rth_camera_primary_rays( int w, int h, float* O3, float* D3, float* rD3 )
{
	Camera c( w, h );
	for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
	{
		Ray r = c.GetPrimaryRay( x, y );
		const size_t p = (size_t)x + (size_t)y * w;
		memcpy( O3 + 3 * p, &r.O, 12 ), memcpy( D3 + 3 * p, &r.D, 12 ), memcpy( rD3 + 3 * p, &r.rD, 12 );
	}
}

// standalone BVH build over caller triangles (9 floats) and spheres (pos, r); nodes_out has room for 2*(nt+ns)+1 nodes
int rth_bvh_build( const float* v9, int nt, const float* pr4, int ns, int builder, rtmi_bvh_node* nodes_out, int* prim_idx_out, int* info )
{
	std::vector<Triangle> tris( nt );
	std::vector<Sphere> sph( ns );
	Material m;
	for (int i = 0; i < ns; i++) sph[i] = Sphere( i, float3( pr4[4 * i], pr4[4 * i + 1], pr4[4 * i + 2] ), pr4[4 * i + 3], m );
	for (int i = 0; i < nt; i++)
	{
		const float* v = v9 + 9 * (size_t)i;
		tris[i] = Triangle( ns + i, float3( v[0], v[1], v[2] ), float3( v[3], v[4], v[5] ), float3( v[6], v[7], v[8] ), m );
	}
	BVH bvh( sph.data(), ns, 0, 0, tris.data(), nt, (BVH::Builder)builder );
	memcpy( nodes_out, bvh.gpuBvhNode, sizeof( GPUBVHNode ) * bvh.nodesUsed );
	memcpy( prim_idx_out, bvh.primitiveIdx, sizeof( int ) * ( nt + ns ) );
	info[0] = (int)bvh.nodesUsed, info[1] = bvh.maxDepth;
	return 0;
}
void rth_triangle_derived( const float* v9, int n, float* N3, float* centroid3 )
{
	Material m;
	for (int i = 0; i < n; i++)
	{
		const float* v = v9 + 9 * (size_t)i;
		Triangle t( 0, float3( v[0], v[1], v[2] ), float3( v[3], v[4], v[5] ), float3( v[6], v[7], v[8] ), m );
		memcpy( N3 + 3 * i, &t.N, 12 ), memcpy( centroid3 + 3 * i, &t.centroid, 12 );
	}
}
long long rth_torus_knot( int nu, int nv, float R, float r, float tube, int p, int q, float cx, float cy, float cz, float* v9_out )
{
	std::vector<float> v;
	MakeTorusKnot( nu, nv, R, r, tube, p, q, float3( cx, cy, cz ), v );
	if (v9_out) memcpy( v9_out, v.data(), v.size() * 4 );
	return (long long)( v.size() / 9 );
}
int rth_write_obj( const char* path, const float* v9, long long n ) { return WriteObj( path, v9, (size_t)n ) ? 0 : -1; }

// rtmi::Camera driven by a script of events (host only; the tests compare it with the oracle's restatement of camera.h):
// ev[2k] = kind, ev[2k+1] = key: 0 KeyDown, 1 KeyUp (through a Renderer that is never initialised: KeyDown/KeyUp only touch the
// camera), 2 Camera::Update, 3 SetFov( val[k] ), 4 SetAspectRatio( val[k] ); returns the number of updates
int rth_camera_script( int w, int h, int n, const int* ev, const float* val, float* cam_out, int* moved_out )
{
	Renderer r( w, h, 0 );
	int updates = 0;
	for (int k = 0; k < n; k++)
	{
		const int kind = ev[2 * k], key = ev[2 * k + 1];
		if (kind == 0) r.KeyDown( key );
		else if (kind == 1) r.KeyUp( key );
		else if (kind == 2) { moved_out[updates] = r.camera->Update() ? 1 : 0; memcpy( cam_out + 16 * updates, r.camera->gpuCamProp, 64 ); updates++; }
		else if (kind == 3) r.camera->SetFov( (int)val[k] );
		else if (kind == 4) r.camera->SetAspectRatio( val[k] );
	}
	return updates;
}

// ---- Renderer ------------------------------------------------------------------------------------------------
void* rth_renderer_create( int w, int h, int device ) { return new Renderer( w, h, device ); }
void rth_renderer_destroy( void* r ) { delete (Renderer*)r; }
void* rth_renderer_scene( void* r ) { return &( (Renderer*)r )->scene; }
int rth_renderer_init( void* r ) { return ( (Renderer*)r )->Init() ? 0 : -1; }
int rth_renderer_upload_scene( void* r ) { return ( (Renderer*)r )->UploadScene() ? 0 : -1; }
void rth_renderer_tick( void* r, float dt ) { ( (Renderer*)r )->Tick( dt ); }
void rth_renderer_set_mode( void* r, int mode ) { ( (Renderer*)r )->SetVisualizationMode( (Renderer::VisualizationMode)mode ); }
void rth_renderer_set_aa( void* r, int aa ) { ( (Renderer*)r )->SetAntiAliasing( aa != 0 ); }
void rth_renderer_set_depth( void* r, int d ) { ( (Renderer*)r )->SetRecusionDepth( d ); }
void rth_renderer_set_spp_per_tick( void* r, int n ) { ( (Renderer*)r )->samplesPerTick = n > 0 ? n : 1; }
void rth_renderer_set_postfx( void* r, const rtmi_postfx* fx, int on ) { ( (Renderer*)r )->SetPostFx( *fx, on != 0 ); }
void rth_renderer_key( void* r, int key, int down ) { if (down) ( (Renderer*)r )->KeyDown( key ); else ( (Renderer*)r )->KeyUp( key ); }
void rth_renderer_set_fov( void* r, int fov ) { ( (Renderer*)r )->camera->SetFov( fov ); }               // template.cpp:356-357
void rth_renderer_set_aspect( void* r, float a ) { ( (Renderer*)r )->camera->SetAspectRatio( a ); }     // template.cpp:370-371
void rth_renderer_camera( void* r, float* cam16 ) { memcpy( cam16, ( (Renderer*)r )->camera->gpuCamProp, 64 ); }
unsigned rth_renderer_frames( void* r ) { return ( (Renderer*)r )->accumulatedFrames; }
const uint32_t* rth_renderer_pixels( void* r ) { return ( (Renderer*)r )->screenPixels; }
const float* rth_renderer_accumulator( void* r ) { Renderer* R = (Renderer*)r; return R->ReadAccumulator() ? &R->accumulator[0].x : 0; }
void* rth_renderer_ctx( void* r ) { return ( (Renderer*)r )->ctx; }
const char* rth_renderer_error( void* r ) { return ( (Renderer*)r )->LastError(); }
float rth_renderer_last_ms( void* r ) { return ( (Renderer*)r )->lastFrameMs; }

} // extern "C"
