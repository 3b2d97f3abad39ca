// qbvh.h -- the reference's 4-wide QBVH (bvh.h:18-27, 75-249) from a binary node array in the reference layout.
//
// Shared by the host mirror (BVH::BuildQBVH, bvh.cpp) and by rtmi_upload_scene (rtmi_host.cpp, option "qbvh"), which needs the
// reference's default traversal order and box chains on the device.  Header-only, no dependencies beyond include/rtmi.h.
#pragma once
#include "../../include/rtmi.h"
#include <cmath>
#include <vector>

namespace rtmi {

// == SIMD_BVH_Node, bvh.h:18-27 (128 bytes): six 4-lane rows of box planes, then the children and their primitive counts.
// child[i] == -1: no child; count[i] > 0: a leaf, child[i] = first position in primitiveIdx; else child[i] = QBVH node index.
struct QbvhNode
{
	float bminx[4], bmaxx[4], bminy[4], bmaxy[4], bminz[4], bmaxz[4];
	int child[4], count[4];
};

namespace qbvh_detail {
inline void put( QbvhNode& q, int slot, const rtmi_bvh_node& n )
{
	q.bminx[slot] = n.aabbMinx, q.bminy[slot] = n.aabbMiny, q.bminz[slot] = n.aabbMinz;
	q.bmaxx[slot] = n.aabbMaxx, q.bmaxy[slot] = n.aabbMaxy, q.bmaxz[slot] = n.aabbMaxz;
	q.child[slot] = n.leftFirst, q.count[slot] = n.primitiveCount;
}
// a new node holding the two children of binary node pair (first, first + 1): AssignNodeQBVH, bvh.h:177-211
inline int add_pair( std::vector<QbvhNode>& out, const rtmi_bvh_node* nodes, int first )
{
	QbvhNode q = {};
	put( q, 0, nodes[first] ), put( q, 1, nodes[first + 1] );
	q.child[2] = q.child[3] = -1;
	out.push_back( q );
	return (int)out.size() - 1;
}
// CollapseBVH, bvh.h:75-174: while the node has a free slot, its interior child with the largest box (the reference's measure:
// |dx| |dy| + |dy| |dz| + |dz| |dx|, first of equals) is replaced by its own two children -- the left one in its slot, the right
// one in the free slot; then every interior child becomes a node of its own, depth first in slot order.
inline void collapse( std::vector<QbvhNode>& out, const rtmi_bvh_node* nodes, int at )
{
	for (int slot = 2; slot < 4; slot++) // (slots 0 and 1 are always taken)
	{
		QbvhNode& q = out[at];
		int pick = -1;
		float best = -1.0f;
		for (int j = 0; j < slot; j++) if (q.count[j] == 0)
		{
			const float ex = std::fabs( q.bmaxx[j] - q.bminx[j] ), ey = std::fabs( q.bmaxy[j] - q.bminy[j] ), ez = std::fabs( q.bmaxz[j] - q.bminz[j] );
			const float area = ex * ey + ey * ez + ez * ex;
			if (area > best) best = area, pick = j;
		}
		if (pick < 0) break;
		const int left = q.child[pick];
		put( q, pick, nodes[left] ), put( q, slot, nodes[left + 1] );
	}
	for (int slot = 0; slot < 4; slot++)
	{
		if (out[at].child[slot] == -1 || out[at].count[slot] != 0) continue;
		const int made = add_pair( out, nodes, out[at].child[slot] );
		out[at].child[slot] = made;
		collapse( out, nodes, made );
	}
}
} // namespace qbvh_detail

// BuildQBVH, bvh.h:213-249: node 0 = {plane leaf, primitive root}, node 1 = the primitive root's children, collapsed.  False (and
// nothing built) when the primitive root is a leaf: there the reference collapses node 0 itself and, without planes, takes the
// empty plane slot for an interior node; callers keep the binary traversal for such scenes.
inline bool BuildQbvh( const rtmi_bvh_node* nodes, int n_nodes, std::vector<QbvhNode>& out )
{
	out.clear();
	if (!nodes || n_nodes < 5 || nodes[2].primitiveCount > 0) return false;
	QbvhNode root = {};
	qbvh_detail::put( root, 0, nodes[1] ), qbvh_detail::put( root, 1, nodes[2] );
	root.child[2] = root.child[3] = -1;
	out.push_back( root );
	out[0].child[1] = qbvh_detail::add_pair( out, nodes, nodes[2].leftFirst );
	qbvh_detail::collapse( out, nodes, 1 );
	return true;
}

} // namespace rtmi
