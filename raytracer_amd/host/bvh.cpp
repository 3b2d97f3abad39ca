// bvh.cpp -- host BVH builders (see bvh.h).  Build with -O2 -ffp-contract=off -fopenmp.
#include "bvh.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <numeric>

namespace rtmi {

namespace {

// growing box with the reference's start values and ternary min/max (bvh.h:29-38)
struct SahBox
{
	float3 bmin = float3( 1e30f ), bmax = float3( -1e30f );
	void grow( const float3& p ) { bmin = fminf3( bmin, p ), bmax = fmaxf3( bmax, p ); }
	void grow( const SahBox& b ) { bmin = fminf3( bmin, b.bmin ), bmax = fmaxf3( bmax, b.bmax ); }
	float area() const
	{
		float3 e = bmax - bmin;
		return e.x * e.y + e.y * e.z + e.z * e.x;
	}
};

// temporary tree node used while subtrees are built concurrently; renumbered into the reference's
// allocation order (children pair taken from nodesUsed at split time, left subtree completed before the
// right one is started, bvh.h:380-392) by Renumber().
struct TmpNode
{
	float3 mn, mx;
	uint first, count; // primitive range
	int left = -1;     // index of left child in the pool (right = left + 1), -1 = leaf
};

struct BuildCtx
{
	const BVH* bvh;
	int* primIdx;
	TmpNode* pool;
	std::atomic<int> used;
};

const int PAR_THRESHOLD = 8192;  // spawn an OpenMP task per child above this many primitives
const int SWEEP_THRESHOLD = 48;  // below this the literal O(n^2) evaluation is cheaper than sorting

} // namespace

BVH::BVH( Sphere* s, int ns, Plane* p, int np, Triangle* t, int nt, Builder builder )
	: spheres_size( ns ), spheres( s ), planes_size( np ), planes( p ), triangles_size( nt ), triangles( t )
{
	if (builder == BINNED_SAH) BuildBVHBinned(); else BuildBVH();
}
BVH::~BVH()
{
	free( primitiveIdx ), free( bvhNode ), free( gpuBvhNode );
}

inline void BVH::PrimBounds( int prim, float3& mn, float3& mx ) const
{
	if (prim < spheres_size)
	{
		const Sphere& s = spheres[prim];
		mn = s.pos - s.r, mx = s.pos + s.r;
	}
	else
	{
		const Triangle& t = triangles[prim - spheres_size];
		mn = fminf3( fminf3( t.pos1, t.pos2 ), t.pos3 ), mx = fmaxf3( fmaxf3( t.pos1, t.pos2 ), t.pos3 );
	}
}

void BVH::Allocate()
{
	free( primitiveIdx ), free( bvhNode ), free( gpuBvhNode );
	const int n = spheres_size + triangles_size, nNodes = NodeCapacity();
	bvhNode = (BVHNode*)aligned_alloc( 64, ( (size_t)nNodes * sizeof( BVHNode ) + 63 ) & ~(size_t)63 );
	gpuBvhNode = (GPUBVHNode*)aligned_alloc( 64, ( (size_t)nNodes * sizeof( GPUBVHNode ) + 63 ) & ~(size_t)63 );
	primitiveIdx = (int*)aligned_alloc( 64, ( (size_t)( n > 0 ? n : 1 ) * sizeof( int ) + 63 ) & ~(size_t)63 );
	memset( (void*)bvhNode, 0, (size_t)nNodes * sizeof( BVHNode ) );
	for (int i = 0; i < n; i++) primitiveIdx[i] = i;
	nodesUsed = 3;
	// node 0: dummy root; node 1: plane leaf; node 2: primitive root (bvh.h:262-271)
	bvhNode[0].leftFirst = 0, bvhNode[0].primitiveCount = 0;
	bvhNode[0].aabbMin = float3( -1e30f ), bvhNode[0].aabbMax = float3( 1e30f );
	bvhNode[1].leftFirst = 0, bvhNode[1].primitiveCount = planes_size;
	bvhNode[1].aabbMin = float3( -1e30f ), bvhNode[1].aabbMax = float3( 1e30f );
	bvhNode[2].leftFirst = 0, bvhNode[2].primitiveCount = n;
}

// bounds of a primitive range, grown in index order from +-1e30 (bvh.h:290-316)
static void RangeBounds( const BVH* b, const int* primIdx, uint first, uint count, float3& mn, float3& mx )
{
	mn = float3( 1e30f ), mx = float3( -1e30f );
	for (uint i = 0; i < count; i++)
	{
		const int leaf = primIdx[first + i];
		if (leaf < b->spheres_size)
		{
			const Sphere& s = b->spheres[leaf];
			mn = fminf3( mn, s.pos - s.r ), mx = fmaxf3( mx, s.pos + s.r );
		}
		else
		{
			const Triangle& t = b->triangles[leaf - b->spheres_size];
			mn = fminf3( mn, t.pos1 ), mn = fminf3( mn, t.pos2 ), mn = fminf3( mn, t.pos3 );
			mx = fmaxf3( mx, t.pos1 ), mx = fmaxf3( mx, t.pos2 ), mx = fmaxf3( mx, t.pos3 );
		}
	}
}

void BVH::UpdateNodeBounds( uint nodeIdx )
{
	BVHNode& node = bvhNode[nodeIdx];
	RangeBounds( this, primitiveIdx, node.leftFirst, node.primitiveCount, node.aabbMin, node.aabbMax );
}

static inline float PrimCentroid( const BVH* b, int prim, int axis )
{
	return prim < b->spheres_size ? b->spheres[prim].pos[axis] : b->triangles[prim - b->spheres_size].centroid[axis];
}
static inline void GrowPrim( const BVH* b, int prim, SahBox& box )
{
	if (prim < b->spheres_size)
	{
		const Sphere& s = b->spheres[prim];
		box.grow( s.pos - s.r ), box.grow( s.pos + s.r );
	}
	else
	{
		const Triangle& t = b->triangles[prim - b->spheres_size];
		box.grow( t.pos1 ), box.grow( t.pos2 ), box.grow( t.pos3 );
	}
}
// SAH cost of "k primitives left of the plane" exactly as bvh.h:439-440 computes it (an empty side makes the
// area overflow to inf, 0*inf = NaN, and the `cost > 0` test turns that into 1e30f)
static inline float SplitCost( int leftCount, const SahBox& l, int rightCount, const SahBox& r )
{
	float cost = leftCount * l.area() + rightCount * r.area();
	return cost > 0 ? cost : 1e30f;
}

// Finds the reference's (bestAxis, bestPos, bestCost) for a primitive range: candidates are visited axis by
// axis, primitive by primitive in range order, and a candidate replaces the incumbent only when strictly
// cheaper (bvh.h:329-342).  For every candidate the left set is { centroid[axis] < pos } (bvh.h:405, 421).
static void BestSplit( const BVH* b, const int* primIdx, uint first, uint count, int& bestAxis, float& bestPos, float& bestCost )
{
	bestAxis = -1, bestPos = 0, bestCost = 1e30f;
	if (count < (uint)SWEEP_THRESHOLD)
	{
		for (int axis = 0; axis < 3; axis++) for (uint i = 0; i < count; i++)
		{
			const float pos = PrimCentroid( b, primIdx[first + i], axis );
			SahBox l, r;
			int lc = 0, rc = 0;
			for (uint k = 0; k < count; k++)
			{
				const int prim = primIdx[first + k];
				if (PrimCentroid( b, prim, axis ) < pos) lc++, GrowPrim( b, prim, l ); else rc++, GrowPrim( b, prim, r );
			}
			const float cost = SplitCost( lc, l, rc, r );
			if (cost < bestCost) bestPos = pos, bestAxis = axis, bestCost = cost;
		}
		return;
	}
	// one sort per axis; cost[k] = cost of the split that puts the k smallest centroids left
	std::vector<uint> order( count );
	std::vector<float> keys( count ), cost( count + 1 );
	std::vector<SahBox> suffix( count + 1 );
	for (int axis = 0; axis < 3; axis++)
	{
		std::iota( order.begin(), order.end(), 0u );
		std::sort( order.begin(), order.end(), [&]( uint a, uint c ) {
			return PrimCentroid( b, primIdx[first + a], axis ) < PrimCentroid( b, primIdx[first + c], axis ); } );
		for (uint k = 0; k < count; k++) keys[k] = PrimCentroid( b, primIdx[first + order[k]], axis );
		suffix[count] = SahBox();
		for (uint k = count; k-- > 0;) suffix[k] = suffix[k + 1], GrowPrim( b, primIdx[first + order[k]], suffix[k] );
		SahBox prefix;
		for (uint k = 0; k <= count; k++)
		{
			cost[k] = SplitCost( (int)k, prefix, (int)( count - k ), suffix[k] );
			if (k < count) GrowPrim( b, primIdx[first + order[k]], prefix );
		}
		for (uint i = 0; i < count; i++)
		{
			const float pos = PrimCentroid( b, primIdx[first + i], axis );
			const uint k = (uint)( std::lower_bound( keys.begin(), keys.end(), pos ) - keys.begin() ); // #centroids < pos
			if (cost[k] < bestCost) bestPos = pos, bestAxis = axis, bestCost = cost[k];
		}
	}
}

static void SubdivideTmp( BuildCtx& c, int nodeIdx )
{
	TmpNode& node = c.pool[nodeIdx];
	const float3 e = node.mx - node.mn;
	const float parentArea = e.x * e.y + e.y * e.z + e.z * e.x;
	const float parentCost = node.count * parentArea;
	if (node.count <= 2) return;
	int axis; float splitPos, bestCost;
	BestSplit( c.bvh, c.primIdx, node.first, node.count, axis, splitPos, bestCost );
	if (bestCost >= parentCost) return;
	// in-place partition, bvh.h:357-375
	int i = node.first, j = i + node.count - 1;
	while (i <= j)
	{
		if (PrimCentroid( c.bvh, c.primIdx[i], axis ) < splitPos) i++;
		else std::swap( c.primIdx[i], c.primIdx[j--] );
	}
	const int leftCount = i - node.first;
	if (leftCount == 0 || leftCount == (int)node.count) return;
	const int l = c.used.fetch_add( 2 ), r = l + 1;
	c.pool[l].first = node.first, c.pool[l].count = leftCount;
	c.pool[r].first = i, c.pool[r].count = node.count - leftCount;
	node.left = l;
	RangeBounds( c.bvh, c.primIdx, c.pool[l].first, c.pool[l].count, c.pool[l].mn, c.pool[l].mx );
	RangeBounds( c.bvh, c.primIdx, c.pool[r].first, c.pool[r].count, c.pool[r].mn, c.pool[r].mx );
	if (node.count >= (uint)PAR_THRESHOLD)
	{
#pragma omp task shared( c ) firstprivate( l )
		SubdivideTmp( c, l );
#pragma omp task shared( c ) firstprivate( r )
		SubdivideTmp( c, r );
#pragma omp taskwait
	}
	else SubdivideTmp( c, l ), SubdivideTmp( c, r );
}

// copies the temporary tree into bvhNode[] in the reference's numbering, iteratively (deep trees)
static void Renumber( BVH* b, const TmpNode* pool )
{
	struct Item { int tmp; uint dst; int depth; };
	std::vector<Item> stack;
	stack.push_back( { 0, b->primitveRootNodeIdx, 0 } );
	b->maxDepth = 0;
	while (!stack.empty())
	{
		Item it = stack.back();
		stack.pop_back();
		const TmpNode& t = pool[it.tmp];
		BVHNode& n = b->bvhNode[it.dst];
		n.aabbMin = t.mn, n.aabbMax = t.mx;
		if (it.depth > b->maxDepth) b->maxDepth = it.depth;
		if (t.left < 0) { n.leftFirst = t.first, n.primitiveCount = t.count; continue; }
		// The reference takes the children pair from nodesUsed when Subdivide() reaches the parent and finishes
		// the left subtree before it starts the right one (bvh.h:380-392), i.e. pairs are numbered in pre-order.
		// Popping the left child before the right one reproduces that numbering.
		const uint l = b->nodesUsed;
		b->nodesUsed += 2;
		n.leftFirst = l, n.primitiveCount = 0;
		stack.push_back( { t.left + 1, l + 1, it.depth + 1 } );
		stack.push_back( { t.left, l, it.depth + 1 } );
	}
}

void BVH::FlattenForGPU()
{
	const int nNodes = NodeCapacity();
	for (int i = 0; i < nNodes; i++) // bvh.h:277-287
	{
		gpuBvhNode[i].aabbMinx = bvhNode[i].aabbMin.x, gpuBvhNode[i].aabbMiny = bvhNode[i].aabbMin.y, gpuBvhNode[i].aabbMinz = bvhNode[i].aabbMin.z;
		gpuBvhNode[i].aabbMaxx = bvhNode[i].aabbMax.x, gpuBvhNode[i].aabbMaxy = bvhNode[i].aabbMax.y, gpuBvhNode[i].aabbMaxz = bvhNode[i].aabbMax.z;
		gpuBvhNode[i].leftFirst = bvhNode[i].leftFirst, gpuBvhNode[i].primitiveCount = bvhNode[i].primitiveCount;
	}
}

void BVH::BuildBVH()
{
	Allocate();
	const int n = spheres_size + triangles_size;
	std::vector<TmpNode> pool( (size_t)n * 2 + 2 );
	BuildCtx c{ this, primitiveIdx, pool.data(), { 1 } };
	pool[0].first = 0, pool[0].count = n;
	RangeBounds( this, primitiveIdx, 0, n, pool[0].mn, pool[0].mx );
#pragma omp parallel
#pragma omp single
	SubdivideTmp( c, 0 );
	Renumber( this, pool.data() );
	FlattenForGPU();
}

// ---------------------------------------------------------------------------------------------------------
// Binned SAH (16 bins on centroid bounds), object-median fallback.  Same node conventions; leaves hold <= 4
// primitives unless all centroids coincide.
static void SubdivideBinnedTmp( BuildCtx& c, int nodeIdx, int depth )
{
	const int BINS = 16;
	TmpNode& node = c.pool[nodeIdx];
	if (node.count <= 2) return;
	const BVH* b = c.bvh;
	float3 cmn( 1e30f ), cmx( -1e30f );
	for (uint i = 0; i < node.count; i++)
	{
		const int prim = c.primIdx[node.first + i];
		float3 ctr( PrimCentroid( b, prim, 0 ), PrimCentroid( b, prim, 1 ), PrimCentroid( b, prim, 2 ) );
		cmn = fminf3( cmn, ctr ), cmx = fmaxf3( cmx, ctr );
	}
	int bestAxis = -1, bestBin = 0;
	float bestCost = 1e30f;
	for (int axis = 0; axis < 3; axis++)
	{
		const float lo = cmn[axis], hi = cmx[axis];
		if (!( hi > lo )) continue;
		const float scale = BINS / ( hi - lo );
		SahBox box[BINS];
		int cnt[BINS] = { 0 };
		for (uint i = 0; i < node.count; i++)
		{
			const int prim = c.primIdx[node.first + i];
			int bin = (int)( ( PrimCentroid( b, prim, axis ) - lo ) * scale );
			bin = bin < 0 ? 0 : ( bin > BINS - 1 ? BINS - 1 : bin );
			cnt[bin]++, GrowPrim( b, prim, box[bin] );
		}
		float rightArea[BINS];
		int rightCnt[BINS];
		SahBox acc;
		int n = 0;
		for (int k = BINS - 1; k > 0; k--) acc.grow( box[k] ), n += cnt[k], rightArea[k] = acc.area(), rightCnt[k] = n;
		acc = SahBox(), n = 0;
		for (int k = 1; k < BINS; k++)
		{
			acc.grow( box[k - 1] ), n += cnt[k - 1];
			if (n == 0 || rightCnt[k] == 0) continue;
			const float cost = n * acc.area() + rightCnt[k] * rightArea[k];
			if (cost < bestCost) bestCost = cost, bestAxis = axis, bestBin = k;
		}
	}
	const float3 e = node.mx - node.mn;
	const float parentCost = node.count * ( e.x * e.y + e.y * e.z + e.z * e.x );
	int i = node.first;
	if (bestAxis >= 0 && ( bestCost < parentCost || node.count > 4 ))
	{
		const float lo = cmn[bestAxis], scale = BINS / ( cmx[bestAxis] - lo );
		int j = i + node.count - 1;
		while (i <= j)
		{
			int bin = (int)( ( PrimCentroid( b, c.primIdx[i], bestAxis ) - lo ) * scale );
			bin = bin < 0 ? 0 : ( bin > BINS - 1 ? BINS - 1 : bin );
			if (bin < bestBin) i++; else std::swap( c.primIdx[i], c.primIdx[j--] );
		}
	}
	else if (node.count <= 4) return;
	int leftCount = i - node.first;
	if (leftCount == 0 || leftCount == (int)node.count)
	{
		if (node.count <= 4) return;
		// all centroids in one bin (or coincident): object median on the widest centroid axis
		int axis = 0;
		const float3 ce = cmx - cmn;
		if (ce.y > ce.x) axis = 1;
		if (ce.z > ce[axis]) axis = 2;
		int* f = c.primIdx + node.first;
		leftCount = node.count / 2;
		std::nth_element( f, f + leftCount, f + node.count, [&]( int a, int d ) {
			const float ka = PrimCentroid( b, a, axis ), kd = PrimCentroid( b, d, axis );
			return ka < kd || ( ka == kd && a < d ); } );
		i = node.first + leftCount;
	}
	const int l = c.used.fetch_add( 2 ), r = l + 1;
	c.pool[l].first = node.first, c.pool[l].count = leftCount;
	c.pool[r].first = i, c.pool[r].count = node.count - leftCount;
	node.left = l;
	RangeBounds( b, c.primIdx, c.pool[l].first, c.pool[l].count, c.pool[l].mn, c.pool[l].mx );
	RangeBounds( b, c.primIdx, c.pool[r].first, c.pool[r].count, c.pool[r].mn, c.pool[r].mx );
	if (node.count >= (uint)PAR_THRESHOLD)
	{
#pragma omp task shared( c ) firstprivate( l, depth )
		SubdivideBinnedTmp( c, l, depth + 1 );
#pragma omp task shared( c ) firstprivate( r, depth )
		SubdivideBinnedTmp( c, r, depth + 1 );
#pragma omp taskwait
	}
	else SubdivideBinnedTmp( c, l, depth + 1 ), SubdivideBinnedTmp( c, r, depth + 1 );
}

void BVH::BuildBVHBinned()
{
	Allocate();
	const int n = spheres_size + triangles_size;
	std::vector<TmpNode> pool( (size_t)n * 2 + 2 );
	BuildCtx c{ this, primitiveIdx, pool.data(), { 1 } };
	pool[0].first = 0, pool[0].count = n;
	RangeBounds( this, primitiveIdx, 0, n, pool[0].mn, pool[0].mx );
#pragma omp parallel
#pragma omp single
	SubdivideBinnedTmp( c, 0, 0 );
	Renumber( this, pool.data() );
	FlattenForGPU();
}

} // namespace rtmi
