// scene.h -- host scene container of the drop-in API (counterpart of the reference's template/scene.h).
//
// Holds the primitives with the reference's types and global object index order (quads, spheres, cubes, planes,
// triangles), builds the BVH on the host (bvh.h) and marshals everything into the flat arrays of
// rtmi_scene_desc -- the same arrays Scene::Scene() prepares for its OpenCL buffers (scene.h:94-217).
// Queries (FindNearest / IsOccluded / GetNormal / GetAlbedo) run on the device behind include/rtmi.h.
#pragma once
#include "bvh.h"
#include "../../include/rtmi.h"
#include <string>
#include <utility>
#include <vector>

namespace rtmi {

class Scene
{
public:
	Scene() = default; // empty; fill with the Add*/LoadObject calls or MakeDefault(), then Finalize()
	~Scene();
	Scene( const Scene& ) = delete;
	Scene& operator=( const Scene& ) = delete;

	// the hard-coded scene of Scene::Scene() (scene.h:42-84): light quad, 2 spheres, cube, 6 planes, 3 x monkey.obj
	bool MakeDefault( const std::string& assetsDir, BVH::Builder builder = BVH::REFERENCE_SAH );
	// the 6-plane room + light quad only (no spheres/cube meshes); used by the mesh configs
	void MakeRoom( float scale = 1.0f );

	void AddQuad( float s, const Material& m, const mat4& T );
	void AddSphere( float3 p, float r, const Material& m );
	void AddCube( float3 pos, float3 size, const Material& m, const mat4& T );
	void AddPlane( float3 N, float d, const Material& m );
	void AddTriangles( const float* v9, size_t count, const Material& m, float3 offset = float3( 0 ) );
	// scene.h:222-291: every face of every shape in file order, vertices + offset, one material per file
	bool LoadObject( const std::string& file, const Material& m, float3 offset = float3( 0 ) );
	// the same reader with per-material support (the reference parses the .mtl and ignores it): every `usemtl`
	// group gets the Material built from its `newmtl` entry of the `mtllib` files, `fallback` where there is none
	bool LoadObjectWithMaterials( const std::string& file, const Material& fallback, float3 offset = float3( 0 ) );
	static bool LoadMaterialLibrary( const std::string& file, std::vector<std::pair<std::string, Material>>& out );
	void SetTexture( const uint32_t* pixels, int w, int h );
	bool LoadTextureRaw( const std::string& file, int w, int h ); // w*h little-endian 0x00RRGGBB words

	// builds the BVH and the marshalled arrays; after this Desc() is valid until the scene changes
	void Finalize( BVH::Builder builder = BVH::REFERENCE_SAH );
	const rtmi_scene_desc* Desc() const { return &desc; }
	int TotalPrimitives() const { return quads_size + spheres_size + cubes_size + planes_size + triangles_size; }

	// scene.h:536-551 (driven every frame by the UI, template.cpp:350-465): the Whitted point lights.  They reach the device at the
	// next Renderer::Tick (rtmi_set_whitted_lights), no Finalize / upload needed.
	void SetLightIntensity( float v ) { intensity = v; MarshalLights(); }
	void SetLightPos( float3 p ) { lightPos = p; MarshalLights(); }
	void SetLightColor( float3 c ) { lightColors[1] = c; MarshalLights(); }
	void SetRoofLightColor( float3 c ) { lightColors[0] = c; MarshalLights(); }
	bool lightsChanged = false;  // the descriptor's light fields are newer than what the device was last given

	float intensity = 24;                                            // scene.h:562
	float3 lightPos = float3( -1, 0, 0 );                            // scene.h:563
	float3 lightColors[2] = { float3( 1, 1, 1 ), float3( 0, 0, 0 ) }; // scene.h:564

	Quad* quads = 0;
	Sphere* spheres = 0;
	Cube* cubes = 0;
	Plane* planes = 0;
	Triangle* triangles = 0;
	int quads_size = 0, spheres_size = 0, cubes_size = 0, planes_size = 0, triangles_size = 0;
	Material def_mat, mirror_mat, glass_mat, light_mat, absorb_all_but_blue_mat, red_mat, cyan_mat;
	BVH* bvh = 0;
	int id = 0;
	bool isInitialized = false;

	// marshalled arrays (scene.h:588-619)
	std::vector<int> primMaterials;
	std::vector<float4> albedos, absorptions, primitives, sphereInfos, cubeBs, triangleInfos;
	std::vector<float> refractiveIndices, quadSizes, sphereInvrs;
	std::vector<mat4> quadMatrices, cubeMatrices, cubeInvMatrices;
	std::vector<uint32_t> texture;
	int tex_w = 0, tex_h = 0;

private:
	bool LoadObjectImpl( const std::string& file, const Material& m, float3 offset, bool useMtl );
	void InitMaterials();
	void MarshalLights();
	std::vector<Quad> vq;
	std::vector<Sphere> vs;
	std::vector<Cube> vc;
	std::vector<Plane> vp;
	std::vector<Triangle> vt;
	rtmi_scene_desc desc = {};
};

// closed-form test meshes for the 10^5..10^7-triangle configs: a (p,q) torus knot tube tessellated
// nu x nv x 2 triangles.  Writes nu*nv*2 triangles as 9 floats each.
void MakeTorusKnot( int nu, int nv, float R, float r, float tube, int p, int q, float3 centre, std::vector<float>& v9 );
bool WriteObj( const std::string& file, const float* v9, size_t ntri );

} // namespace rtmi
