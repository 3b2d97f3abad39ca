// integration_tu.cpp -- INTEGRATION.md sections 1-3 as a translation unit: the three code blocks below are those sections VERBATIM
// (tests/test_abi_and_host.py checks that they are, character for character), compiled against compat/ the way a maintainer of the
// reference would compile renderer.cpp after the switch: `#include "precomp.h"`, namespace Tmpl8, the reference's member names.
// __graft_entry__.build() builds it into tests/compat/libcompat_tu.so; tests/test_gpu_compat.py drives it on the GPU and compares
// with the library's own Python path and the reference's golden vectors.
#include "precomp.h"
#include "renderer.h"

// ---- INTEGRATION.md 1 ----
#include "rtmi.h"
// members: rtmi_ctx* ctx = nullptr;   (instead of the 6 Kernel* and ~25 Buffer* members, renderer.h:80-123)
void Renderer::Init()
{
	accumulator = (float4*)MALLOC64( SCRWIDTH * SCRHEIGHT * 16 );
	memset( accumulator, 0, SCRWIDTH * SCRHEIGHT * 16 );
	if (rtmi_create( &ctx, 0, SCRWIDTH, SCRHEIGHT ) != RTMI_OK) FatalError( "no MI355X / librtmi" );
	seeds = new uint[SCRWIDTH * SCRHEIGHT];
	for (int i = 0; i < SCRWIDTH * SCRHEIGHT; i++) seeds[i] = RandomUInt();       // renderer.cpp:51-55, unchanged
	rtmi_set_seeds( ctx, seeds );                                                  // was seedBuffer->CopyToDevice
	rtmi_set_camera( ctx, &camera->gpuCamProp[0].x );                              // was cameraPropBuffer->CopyToDevice (:63)
}
// ---- end 1 ----

void Renderer::Tick( float deltaTime )
{
	(void)deltaTime;
	if (!sceneUploaded)
	{
// ---- INTEGRATION.md 2 ----
rtmi_scene_desc d = {};
d.n_quads = scene.quads_size; d.n_spheres = scene.spheres_size; d.n_cubes = scene.cubes_size;
d.n_planes = scene.planes_size; d.n_triangles = scene.triangles_size;           // renderer.cpp:457
d.quad_inv_T = scene.quadMatrices[0].cell;  d.quad_size = scene.quadSizes;        // scene.h:117-118
d.sphere_info = &scene.sphereInfos[0].x;    d.sphere_invr = scene.sphereInvrs;    // scene.h:132, 135
d.cube_M = scene.cubeMatrices[0].cell; d.cube_inv_M = scene.cubeInvMatrices[0].cell; d.cube_b = &scene.cubeBs[0].x; // :148-153
d.tri_verts = &scene.triangleInfos[0].x;                                          // scene.h:177-179
d.primitives = &scene.primitives[0].x; d.albedo = &scene.albedos[0].x;            // scene.h:120-182
d.material_type = scene.primMaterials; d.ior = scene.refractiveIndices; d.absorption = &scene.absorptions[0].x;
d.nodes = (const rtmi_bvh_node*)scene.bvh->gpuBvhNode; d.n_nodes = scene.bvh->nodesUsed;   // scene.h:216, bvh.h:11-16
d.prim_idx = scene.bvh->primitiveIdx;                                             // scene.h:217
memcpy( d.light_corners, scene.lightCorners, 48 );                                // scene.h:207-208 (the `lights` local, kept as a member)
d.light_A = scene.quads[0].A; d.light_s = scene.quads[0].s;                       // renderer.cpp:483
memcpy( d.light_emission, &scene.quads[0].material.emission, 12 );
d.texture = scene.texturePixels; d.tex_w = scene.textureWidth; d.tex_h = scene.textureHeight; // scene.h:211-212 (`logo.pixels`, 128 x 64)
for (int i = 0; i < 2; i++)                                                       // Whitted lights, by value: scene.h:343-358, 562-564
	memcpy( d.whitted_light_pos[i], &scene.GetLightPos()[i], 12 ), memcpy( d.whitted_light_color[i], &scene.GetLightColor()[i], 12 );
d.whitted_intensity = scene.intensity;
if (rtmi_upload_scene( ctx, &d ) != RTMI_OK) FatalError( rtmi_last_error( ctx ) );
// ---- end 2 ----
		sceneUploaded = true;
	}
	if (scene.LightsChanged()) rtmi_set_whitted_lights( ctx, &scene.Desc()->whitted_light_pos[0][0], &scene.Desc()->whitted_light_color[0][0], scene.Desc()->whitted_intensity );
// ---- INTEGRATION.md 3 ----
if (visualizationMode == PathTracing) {
	rtmi_render( ctx, RTMI_PATHTRACING, accumulatedFrames, 1, 1, recursionDepth, 0 );   // GeneratePrimaryRays..Connect loop
	accumulatedFrames++;
	rtmi_finalize( ctx, screen->pixels, accumulatedFrames, nullptr );                   // Finalize + deviceBuffer->CopyFromDevice
} else {
	rtmi_render( ctx, visualizationMode, 0, 1, 1, recursionDepth, anti_aliasing );      // the CPU Trace() loop, :578-618
	rtmi_finalize( ctx, screen->pixels, 0, nullptr );
}
if (camera->Update()) { accumulatedFrames = 0; rtmi_reset_accumulator( ctx ); rtmi_set_camera( ctx, &camera->gpuCamProp[0].x ); } // :622-633
// ---- end 3 ----
}

// ---- what the tests call ------------------------------------------------------------------------------------------------------
static Renderer* app = 0;
extern "C" {

int tu_init() // the template's startup: app->screen = new Surface( SCRWIDTH, SCRHEIGHT ); app->Init()
{
	if (app) return 0;
	app = new Renderer();
	app->screen = new Surface( SCRWIDTH, SCRHEIGHT );
	app->Init();
	return 0;
}
int tu_width() { return SCRWIDTH; }
int tu_height() { return SCRHEIGHT; }
const unsigned* tu_seeds() { return app->seeds; }
// `ticks` frames in the given mode; screen->pixels afterwards
int tu_tick( int mode, int ticks, unsigned* pixels_out )
{
	if ((int)app->visualizationMode != mode) app->SetVisualizationMode( (Renderer::VisualizationMode)mode );
	for (int i = 0; i < ticks; i++) app->Tick( 0.0f );
	memcpy( pixels_out, app->screen->pixels, (size_t)SCRWIDTH * SCRHEIGHT * 4 );
	return (int)app->accumulatedFrames;
}
void tu_set_light( float x, float y, float z, float intensity ) { app->scene.SetLightPos( float3( x, y, z ) ); app->scene.SetLightIntensity( intensity ); }
// Scene::FindNearest / IsOccluded / GetNormal / GetAlbedo / GetMaterial, one ray per call as the reference's loops make them
void tu_scene_queries( int n, int use_qbvh, const float* O3, const float* D3, const float* tmax, int* obj, float* t, int* occluded, float* N3, float* albedo3, int* material )
{
	Scene& scene = app->scene;
	scene.SetQBVH( use_qbvh != 0 );
	for (int i = 0; i < n; i++)
	{
		Ray ray( float3( O3[3 * i], O3[3 * i + 1], O3[3 * i + 2] ), float3( D3[3 * i], D3[3 * i + 1], D3[3 * i + 2] ) );
		scene.FindNearest( ray );
		obj[i] = ray.objIdx, t[i] = ray.t;
		const float3 I = ray.IntersectionPoint();
		const float3 N = scene.GetNormal( ray.objIdx, I, ray.D ), a = scene.GetAlbedo( ray.objIdx, I, scene.GetMaterial( ray.objIdx ) );
		memcpy( N3 + 3 * i, &N.x, 12 ), memcpy( albedo3 + 3 * i, &a.x, 12 );
		material[i] = (int)scene.GetMaterial( ray.objIdx ).type;
		Ray shadow( ray.O, ray.D, tmax[i] );
		occluded[i] = scene.IsOccluded( shadow ) ? 1 : 0;
	}
}
// the primary ray of pixel (x, y) against every analytic object of the scene and `n_tri` of its triangles with <Primitive>::Intersect,
// the way Scene::FindNearest looked before the BVH (scene.h:359-368 + a loop): nearest object index and t
void tu_brute_force( int x, int y, int n_tri, int* obj, float* t )
{
	const Scene& scene = app->scene;
	Ray ray = app->camera->GetPrimaryRay( x, y );
	for (int i = 0; i < scene.quads_size; i++) scene.quads[i].Intersect( ray );
	for (int i = 0; i < scene.spheres_size; i++) scene.spheres[i].Intersect( ray );
	for (int i = 0; i < scene.cubes_size; i++) scene.cubes[i].Intersect( ray );
	for (int i = 0; i < scene.planes_size; i++) scene.planes[i].Intersect( ray );
	for (int i = 0; i < n_tri && i < scene.triangles_size; i++) scene.triangles[i].Intersect( ray );
	*obj = ray.objIdx, *t = ray.t;
}

}
