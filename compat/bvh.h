// compat/bvh.h -- bvh.h of the reference: BVHNode (bvh.h:4-9), GPUBVHNode (bvh.h:11-16), QBVHNode's data (bvh.h:18-35) and class
// BVH with the members Scene and Renderer read (gpuBvhNode, nodesUsed, primitiveIdx, bvhNode, root indices, qbvhNode / BuildQBVH).
// The builders are the host ones of raytracer_amd/host/bvh.cpp (the reference's tree, bit for bit); traversal is not here:
// IntersectBVH / IntersectQBVH / their shadow variants are what Scene::FindNearest / IsOccluded reach on the device.
#pragma once
#include "primitive.h"
namespace Tmpl8 {
using rtmi::BVHNode;
using rtmi::GPUBVHNode;
using QBVHNode = rtmi::QbvhNode;
using rtmi::BVH;
}
