// compat/scene.h -- template/scene.h of the reference: class Scene with the reference's data members (pointer-typed arrays as in
// scene.h:563-629, so `scene.quadSizes`, `&scene.sphereInfos[0].x`, `scene.bvh->gpuBvhNode` ... of INTEGRATION.md 2 compile) and its
// query surface.  The container behind it is rtmi::Scene (raytracer_amd/host/scene.h: reference-exact marshalling, OBJ ingest, BVH
// build); the queries run on the device:
//   FindNearest / IsOccluded  (scene.h:359-379) -> rtmi_trace_rays        (follows useQBVH: option qbvh of the handle)
//   GetNormal / GetAlbedo     (scene.h:380-446) -> rtmi_hit_attributes
//   GetMaterial               (scene.h:448-478)    a table lookup, on the host
// one ray per call like the reference's CPU loops: for compiling and agreeing, not for speed (FindNearest( rays, n ) is the batch
// form).  The CPU renderer's helpers (RandomPointOnLight, DirectIllumination, DiffuseReflection, scene.h:317-341, 490-531) belong to
// Renderer::Trace / Sample, which rtmi_render replaces: they are not here.
// Two members the reference keeps as locals of its constructor are members here so that the upload can name them:
// lightCorners (`lights`, scene.h:207) and texturePixels / textureWidth / textureHeight (`logo`, scene.h:211).
#pragma once
#include "bvh.h"
#include <string>
namespace Tmpl8 {

class Scene
{
public:
	// the reference's hard-coded scene (scene.h:42-217); its assets (monkey.obj, the logo) are looked up in $RTMI_ASSETS, else ./assets
	Scene()
	{
		const char* dir = getenv( "RTMI_ASSETS" );
		if (!impl.MakeDefault( dir ? dir : "assets" )) FatalError( "Scene: assets not found in %s", dir ? dir : "assets" );
		Sync();
	}
	// an empty container: fill through Host() (AddQuad / AddSphere / LoadObject ... of rtmi::Scene), then Finalize()
	struct Empty {};
	explicit Scene( Empty ) {}
	~Scene() { if (queryCtx) rtmi_destroy( queryCtx ); }
	Scene( const Scene& ) = delete;
	Scene& operator=( const Scene& ) = delete;
	rtmi::Scene& Host() { return impl; }
	void Finalize( BVH::Builder builder = BVH::REFERENCE_SAH ) { impl.Finalize( builder ); Sync(); queryUploaded = false; }
	const rtmi_scene_desc* Desc() const { return impl.Desc(); }
	// queries go to `ctx` (a handle this scene has been uploaded on, e.g. the renderer's) instead of a handle of the scene's own
	void UseDevice( rtmi_ctx* ctx ) { shared = ctx; }

	void SetTime( float t ) { animTime = 0; if (isDynamic) animTime = t; } // scene.h:293-315; like the reference's GPU path, the device keeps the transforms it was given
	float3* GetLightPos() const // scene.h:343-350, without the dangling pointer
	{
		const rtmi_scene_desc* d = impl.Desc();
		for (int i = 0; i < 2; i++) lightsOut[i] = float3( d->whitted_light_pos[i][0], d->whitted_light_pos[i][1], d->whitted_light_pos[i][2] );
		return lightsOut;
	}
	float3* GetLightColor() { return lightColors; }
	int GetNLights() const { return 2; }
	void FindNearest( Ray& ray ) const
	{
		int32_t obj = -1;
		float t = ray.t;
		rtmi_ctx* c = QueryDevice();
		if (rtmi_trace_rays( c, 1, &ray.O.x, &ray.D.x, &ray.t, 0, &obj, &t ) != RTMI_OK) FatalError( "FindNearest: %s", rtmi_last_error( c ) );
		if (obj >= 0) ray.t = t, ray.objIdx = obj;
	}
	void FindNearest( Ray* rays, int n ) const // not in the reference: the same query for n rays in one launch
	{
		std::vector<float> O( 3 * (size_t)n ), D( 3 * (size_t)n ), tin( n ), t( n );
		std::vector<int32_t> obj( n );
		for (int i = 0; i < n; i++) memcpy( &O[3 * i], &rays[i].O.x, 12 ), memcpy( &D[3 * i], &rays[i].D.x, 12 ), tin[i] = rays[i].t;
		rtmi_ctx* c = QueryDevice();
		if (rtmi_trace_rays( c, n, O.data(), D.data(), tin.data(), 0, obj.data(), t.data() ) != RTMI_OK) FatalError( "FindNearest: %s", rtmi_last_error( c ) );
		for (int i = 0; i < n; i++) if (obj[i] >= 0) rays[i].t = t[i], rays[i].objIdx = obj[i];
	}
	bool IsOccluded( Ray& ray ) const // the ray itself is left as it was (the reference's leaves the quad's / cube's t in it)
	{
		int32_t occ = 0;
		float t = ray.t;
		rtmi_ctx* c = QueryDevice();
		if (rtmi_trace_rays( c, 1, &ray.O.x, &ray.D.x, &ray.t, 1, &occ, &t ) != RTMI_OK) FatalError( "IsOccluded: %s", rtmi_last_error( c ) );
		return occ != 0;
	}
	float3 GetNormal( int objIdx, float3 I, float3 wo ) const
	{
		if (objIdx == -1) return float3( 0 );
		float3 N, albedo;
		int32_t type;
		Attributes( objIdx, I, wo, N, albedo, type );
		return N;
	}
	// `mat` is what GetMaterial( objIdx ) returns at every call site of the reference; the device reads the same colour from its table
	float3 GetAlbedo( int objIdx, float3 I, Material ) const
	{
		if (objIdx == -1) return float3( 0 );
		float3 N, albedo;
		int32_t type;
		Attributes( objIdx, I, float3( 0, 0, 1 ), N, albedo, type );
		return albedo;
	}
	Material GetMaterial( int objIdx ) const
	{
		if (objIdx < 0 || objIdx >= impl.TotalPrimitives()) return Material();
		int k = objIdx;
		if (k < quads_size) return quads[k].material;
		if (( k -= quads_size ) < spheres_size) return spheres[k].material;
		if (( k -= spheres_size ) < cubes_size) return cubes[k].material;
		if (( k -= cubes_size ) < planes_size) return planes[k].material;
		return triangles[k - planes_size].material;
	}
	float GetReflectivity( int objIdx, float3 ) const { return objIdx == 1 ? 1.0f : objIdx == 6 ? 0.3f : 0.0f; } // scene.h:479-484
	float GetRefractivity( int objIdx, float3 ) const { return objIdx == 3 ? 1.0f : 0.0f; }                      // scene.h:485-488
	void SetIsDynamicScene( bool _isDynamic ) { isDynamic = _isDynamic; }
	// scene.h:536-551.  The Whitted lights reach a device with rtmi_set_whitted_lights( ctx, &Desc()->whitted_light_pos[0][0], ... )
	// -- LightsChanged() tells a Tick when -- no upload needed
	void SetLightIntensity( float newIntensity ) { intensity = newIntensity; impl.SetLightIntensity( newIntensity ); }
	void SetLightPos( float3 newLightPos ) { lightPos = newLightPos; impl.SetLightPos( newLightPos ); }
	void SetLightColor( float3 newLightColor ) { lightColors[1] = newLightColor; impl.SetLightColor( newLightColor ); }
	void SetRoofLightColor( float3 newLightColor ) { lightColors[0] = newLightColor; impl.SetRoofLightColor( newLightColor ); }
	bool LightsChanged() { const bool c = impl.lightsChanged; impl.lightsChanged = false; return c; }
	void SetQBVH( bool mode ) { useQBVH = mode; }

	float animTime = 0;
	bool isDynamic = false;
	float intensity = 24;
	float3 lightPos = float3( -1, 0, 0 );
	float3 lightColors[2] = { float3( 1, 1, 1 ), float3( 0, 0, 0 ) };
	Quad* quads = 0;
	Sphere* spheres = 0;
	Cube* cubes = 0;
	Plane* planes = 0;
	Triangle* triangles = 0;
	int quads_size = 0, spheres_size = 0, cubes_size = 0, planes_size = 0, triangles_size = 0;
	Material def_mat, mirror_mat, glass_mat, light_mat, absorb_all_but_blue_mat, red_mat, cyan_mat;
	BVH* bvh = 0;
	int id = 0;
	bool useQBVH = true; // scene.h:584: the reference's default traversal
	int* primMaterials = 0;
	float4* albedos = 0;
	float* refractiveIndices = 0;
	float4* absorptions = 0;
	mat4* quadMatrices = 0;
	float* quadSizes = 0;
	float4* sphereInfos = 0;
	mat4* cubeMatrices = 0;
	mat4* cubeInvMatrices = 0;
	float4* cubeBs = 0;
	float4* triangleInfos = 0;
	float4* primitives = 0;
	float* sphereInvrs = 0;
	float4 lightCorners[3];
	uint* texturePixels = 0;
	int textureWidth = 0, textureHeight = 0;
	bool isInitialized = false;

private:
	void Sync() // the reference's members, pointing into the container
	{
		quads = (Quad*)impl.quads, spheres = (Sphere*)impl.spheres, cubes = (Cube*)impl.cubes, planes = (Plane*)impl.planes, triangles = (Triangle*)impl.triangles;
		quads_size = impl.quads_size, spheres_size = impl.spheres_size, cubes_size = impl.cubes_size, planes_size = impl.planes_size, triangles_size = impl.triangles_size;
		def_mat = impl.def_mat, mirror_mat = impl.mirror_mat, glass_mat = impl.glass_mat, light_mat = impl.light_mat;
		absorb_all_but_blue_mat = impl.absorb_all_but_blue_mat, red_mat = impl.red_mat, cyan_mat = impl.cyan_mat;
		bvh = impl.bvh, id = impl.id, isInitialized = impl.isInitialized;
		primMaterials = impl.primMaterials.data(), albedos = impl.albedos.data(), refractiveIndices = impl.refractiveIndices.data(), absorptions = impl.absorptions.data();
		quadMatrices = impl.quadMatrices.data(), quadSizes = impl.quadSizes.data(), sphereInfos = impl.sphereInfos.data();
		cubeMatrices = impl.cubeMatrices.data(), cubeInvMatrices = impl.cubeInvMatrices.data(), cubeBs = impl.cubeBs.data();
		triangleInfos = impl.triangleInfos.data(), primitives = impl.primitives.data(), sphereInvrs = impl.sphereInvrs.data();
		memcpy( lightCorners, impl.Desc()->light_corners, 48 );
		texturePixels = impl.texture.data(), textureWidth = impl.tex_w, textureHeight = impl.tex_h;
		intensity = impl.intensity, lightPos = impl.lightPos, lightColors[0] = impl.lightColors[0], lightColors[1] = impl.lightColors[1];
	}
	rtmi_ctx* QueryDevice() const
	{
		if (shared) return shared;
		if (!queryCtx && rtmi_create( &queryCtx, 0, 64, 64 ) != RTMI_OK) FatalError( "Scene: no MI355X / librtmi (the queries run on the device)" );
		if (!queryUploaded || queryQbvh != useQBVH)
		{
			rtmi_set_option( queryCtx, "qbvh", useQBVH ? 1 : 0 );
			if (rtmi_upload_scene( queryCtx, impl.Desc() ) != RTMI_OK) FatalError( "Scene: %s", rtmi_last_error( queryCtx ) );
			queryUploaded = true, queryQbvh = useQBVH;
		}
		return queryCtx;
	}
	void Attributes( int objIdx, const float3& I, const float3& wo, float3& N, float3& albedo, int32_t& type ) const
	{
		rtmi_ctx* c = QueryDevice();
		const int32_t obj = objIdx;
		if (rtmi_hit_attributes( c, 1, &obj, &I.x, &wo.x, &N.x, &albedo.x, &type ) != RTMI_OK) FatalError( "GetNormal / GetAlbedo: %s", rtmi_last_error( c ) );
	}
	rtmi::Scene impl;
	rtmi_ctx* shared = 0;
	mutable rtmi_ctx* queryCtx = 0;
	mutable bool queryUploaded = false, queryQbvh = false;
	mutable float3 lightsOut[2];
};

} // namespace Tmpl8
