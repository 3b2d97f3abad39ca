// bvh.h -- host-built BVH of the drop-in API (counterpart of the reference's bvh.h).
//
// The node array keeps the reference's conventions so that anything written against BVH::bvhNode /
// gpuBvhNode / primitiveIdx keeps working and so that rtmi_upload_scene receives what scene.h:216-217 uploads:
//   node 0 = dummy root, node 1 = plane leaf (infinite box), node 2 = primitive root, nodesUsed starts at 3
//   (bvh.h:56, 262-274); the two children of an interior node are adjacent (bvh.h:380-381); a node is a leaf
//   iff primitiveCount > 0 (bvh.h:8); primitiveIdx[k] < spheres_size selects a sphere, else triangle
//   k - spheres_size (bvh.h:298-314).
// Two builders fill it:
//   BuildBVH()        the reference's full-sweep SAH (every primitive centroid x 3 axes is a candidate,
//                     bvh.h:318-441), restated so that it yields the SAME tree bit for bit but evaluates all
//                     candidates of a node from one sort + prefix/suffix sweep per axis (O(n log n) per node
//                     instead of the reference's O(n^2)), which makes it usable at 10^5..10^6 triangles;
//   BuildBVHBinned()  a 16-bin SAH builder with the same node format for 10^6..10^7 triangles.
// Traversal lives on the device (HIP kernels behind include/rtmi.h), not here.
#pragma once
#include "rt_types.h"
#include "qbvh.h"
#include <vector>

namespace rtmi {

struct alignas( 32 ) BVHNode // bvh.h:4-9
{
	float3 aabbMin, aabbMax;
	uint leftFirst, primitiveCount;
	bool isLeaf() const { return primitiveCount > 0; }
};
struct GPUBVHNode // bvh.h:11-16 == rtmi_bvh_node
{
	float aabbMinx, aabbMiny, aabbMinz;
	float aabbMaxx, aabbMaxy, aabbMaxz;
	int leftFirst, primitiveCount;
};

class BVH
{
public:
	enum Builder { REFERENCE_SAH = 0, BINNED_SAH = 1 };
	int spheres_size = 0;
	Sphere* spheres = 0;
	int planes_size = 0;
	Plane* planes = 0;
	int triangles_size = 0;
	Triangle* triangles = 0;
	int* primitiveIdx = 0;
	BVHNode* bvhNode = 0;
	GPUBVHNode* gpuBvhNode = 0;
	uint rootNodeIdx = 0, planeRootNodeIdx = 1, primitveRootNodeIdx = 2, nodesUsed = 3;
	int maxDepth = 0; // depth of the deepest leaf below node 2 (the device traversal stack bound)
	// QBVH (bvh.h:58-60): the reference's 4-wide collapse of the binary tree, its default traversal structure.  Built on demand
	// by BuildQBVH(); the device builds its own copy from the node array when a handle's option qbvh is on (rtmi_upload_scene).
	std::vector<QbvhNode> qbvhNode;
	uint nodesUsedQBVH = 0;
	bool BuildQBVH() { const bool ok = BuildQbvh( (const rtmi_bvh_node*)gpuBvhNode, (int)nodesUsed, qbvhNode ); nodesUsedQBVH = (uint)qbvhNode.size(); return ok; } // bvh.h:213-249

	BVH() = default;
	BVH( Sphere* spheres, int spheres_size, Plane* planes, int planes_size, Triangle* triangles, int triangles_size, Builder builder = REFERENCE_SAH );
	~BVH();
	BVH( const BVH& ) = delete;
	BVH& operator=( const BVH& ) = delete;
	void BuildBVH();       // reference-exact tree
	void BuildBVHBinned(); // fast builder, same format
	// N * 2 + 1 nodes as the reference allocates (bvh.h:253), but never fewer than the three fixed nodes 0..2
	int NodeCapacity() const { const int n = ( spheres_size + triangles_size ) * 2 + 1; return n < 3 ? 3 : n; }

private:
	void Allocate();
	void UpdateNodeBounds( uint nodeIdx );
	void FlattenForGPU();
	inline void PrimBounds( int prim, float3& mn, float3& mx ) const;
};

} // namespace rtmi
