// bvh.cpp -- host BVH builders (see bvh.h).  Build with -O2 -ffp-contract=off -fopenmp.
#include "bvh.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <numeric>
#include <cstdio>
#include <omp.h>

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
#pragma omp parallel for schedule( static )
	for (int i = 0; i < nNodes; i++) memset( (void*)( bvhNode + i ), 0, sizeof( BVHNode ) );
#pragma omp parallel for schedule( static )
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
#pragma omp parallel for schedule( static )
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
// The builder works on an array of 40-byte records {bounds, centroid, primitive id} that is kept in node order (a
// partition moves the records), so every pass over a node streams memory instead of gathering 116-byte Triangles
// through primitiveIdx.  A node is processed in three passes: centroid bounds; bin boxes and counts of all three axes;
// partition + bounds of the two children.  Nodes above BIG_NODE primitives (the top levels of a 10^6..10^7-triangle
// tree, where there are fewer nodes than cores) run each pass as OpenMP tasks over chunks and merge the chunk results:
// min / max / counts do not depend on the order, so the tree (split planes, boxes, primitive sets per node) is the same
// whatever the thread count; only the order of primitives inside a leaf range can differ.
namespace {
const int BINS = 16;
const int BIG_NODE = 1 << 16, CHUNK = 1 << 14;
struct Rec { float3 mn, mx, c; int id; };
struct AxisBins { SahBox box[3][BINS]; int cnt[3][BINS]; };
struct BinnedCtx { Rec* rec; TmpNode* pool; float3* cb; std::atomic<int> used; }; // cb[2*node], cb[2*node+1]: centroid bounds of the node
struct Side { SahBox box, cen; void add( const Rec& q ) { box.bmin = fminf3( box.bmin, q.mn ), box.bmax = fmaxf3( box.bmax, q.mx ), cen.grow( q.c ); } void add( const Side& o ) { box.grow( o.box ), cen.grow( o.cen ); } };

inline int BinOf( float c, float lo, float scale )
{
	int bin = (int)( ( c - lo ) * scale );
	return bin < 0 ? 0 : ( bin > BINS - 1 ? BINS - 1 : bin );
}
void CentroidBounds( const Rec* r, uint n, float3& cmn, float3& cmx )
{
	cmn = float3( 1e30f ), cmx = float3( -1e30f );
	for (uint i = 0; i < n; i++) cmn = fminf3( cmn, r[i].c ), cmx = fmaxf3( cmx, r[i].c );
}
void FillBins( const Rec* r, uint n, const float3& cmn, const float* scale, const bool* valid, AxisBins& out )
{
	for (int a = 0; a < 3; a++) for (int k = 0; k < BINS; k++) out.cnt[a][k] = 0, out.box[a][k] = SahBox();
	for (uint i = 0; i < n; i++)
		for (int a = 0; a < 3; a++)
		{
			if (!valid[a]) continue;
			const int bin = BinOf( r[i].c[a], cmn[a], scale[a] );
			SahBox& bx = out.box[a][bin];
			out.cnt[a][bin]++, bx.bmin = fminf3( bx.bmin, r[i].mn ), bx.bmax = fmaxf3( bx.bmax, r[i].mx ); // mn <= mx per record
		}
}
void RecBounds( const Rec* r, uint n, float3& mn, float3& mx )
{
	mn = float3( 1e30f ), mx = float3( -1e30f );
	for (uint i = 0; i < n; i++) mn = fminf3( mn, r[i].mn ), mx = fmaxf3( mx, r[i].mx );
}

void SubdivideBinnedTmp( BinnedCtx& c, int nodeIdx, int depth )
{
	TmpNode& node = c.pool[nodeIdx];
	if (node.count <= 2) return;
	Rec* rec = c.rec + node.first;
	const uint n = node.count;
	const bool big = n >= (uint)BIG_NODE;
	const int nch = big ? (int)( ( n + CHUNK - 1 ) / CHUNK ) : 1;
	auto lo_of = [&]( int ch ) { return (uint)( (size_t)n * ch / nch ); };
	// centroid bounds: computed by the pass that created this node (the parent's partition, or the record pass for the root)
	const float3 cmn = c.cb[2 * nodeIdx], cmx = c.cb[2 * nodeIdx + 1];
	// pass 2: bins of the three axes
	float scale[3] = { 0, 0, 0 };
	bool valid[3];
	for (int a = 0; a < 3; a++) { valid[a] = cmx[a] > cmn[a]; if (valid[a]) scale[a] = BINS / ( cmx[a] - cmn[a] ); }
	AxisBins bins;
	if (!big) FillBins( rec, n, cmn, scale, valid, bins );
	else
	{
		std::vector<AxisBins> part( nch );
#pragma omp taskloop default( shared ) grainsize( 1 )
		for (int ch = 0; ch < nch; ch++) FillBins( rec + lo_of( ch ), lo_of( ch + 1 ) - lo_of( ch ), cmn, scale, valid, part[ch] );
		for (int a = 0; a < 3; a++) for (int k = 0; k < BINS; k++) bins.cnt[a][k] = 0, bins.box[a][k] = SahBox();
		for (int ch = 0; ch < nch; ch++) for (int a = 0; a < 3; a++) for (int k = 0; k < BINS; k++)
			bins.cnt[a][k] += part[ch].cnt[a][k], bins.box[a][k].grow( part[ch].box[a][k] );
	}
	int bestAxis = -1, bestBin = 0;
	float bestCost = 1e30f;
	for (int axis = 0; axis < 3; axis++)
	{
		if (!valid[axis]) continue;
		const SahBox* box = bins.box[axis];
		const int* cnt = bins.cnt[axis];
		float rightArea[BINS];
		int rightCnt[BINS];
		SahBox acc;
		int m = 0;
		for (int k = BINS - 1; k > 0; k--) acc.grow( box[k] ), m += cnt[k], rightArea[k] = acc.area(), rightCnt[k] = m;
		acc = SahBox(), m = 0;
		for (int k = 1; k < BINS; k++)
		{
			acc.grow( box[k - 1] ), m += cnt[k - 1];
			if (m == 0 || rightCnt[k] == 0) continue;
			const float cost = m * acc.area() + rightCnt[k] * rightArea[k];
			if (cost < bestCost) bestCost = cost, bestAxis = axis, bestBin = k;
		}
	}
	const float3 e = node.mx - node.mn;
	const float parentCost = node.count * ( e.x * e.y + e.y * e.z + e.z * e.x );
	int leftCount = 0;
	bool haveBounds = false;
	Side L, R;
	if (bestAxis >= 0 && ( bestCost < parentCost || node.count > 4 ))
	{
		const float lo = cmn[bestAxis], sc = scale[bestAxis];
		if (!big)
		{
			int i = 0, j = (int)n - 1;
			while (i <= j) // every record is classified exactly once: the children's bounds come out of the same pass
			{
				if (BinOf( rec[i].c[bestAxis], lo, sc ) < bestBin) L.add( rec[i] ), i++; else R.add( rec[i] ), std::swap( rec[i], rec[j--] );
			}
			leftCount = i, haveBounds = true;
		}
		else
		{
			// pass 3 in chunks: count, then scatter (left part in chunk order, right part behind it) with the children's bounds
			std::vector<uint> nl( nch + 1, 0 );
#pragma omp taskloop default( shared ) grainsize( 1 )
			for (int ch = 0; ch < nch; ch++)
			{
				uint k = 0;
				for (uint i = lo_of( ch ); i < lo_of( ch + 1 ); i++) k += BinOf( rec[i].c[bestAxis], lo, sc ) < bestBin;
				nl[ch + 1] = k;
			}
			for (int ch = 0; ch < nch; ch++) nl[ch + 1] += nl[ch];
			leftCount = (int)nl[nch];
			Rec* tmp = (Rec*)malloc( (size_t)n * sizeof( Rec ) );
			std::vector<Side> lb( nch ), rb( nch );
#pragma omp taskloop default( shared ) grainsize( 1 )
			for (int ch = 0; ch < nch; ch++)
			{
				uint l = nl[ch], r = (uint)leftCount + ( lo_of( ch ) - nl[ch] );
				for (uint i = lo_of( ch ); i < lo_of( ch + 1 ); i++)
				{
					const Rec& q = rec[i];
					if (BinOf( q.c[bestAxis], lo, sc ) < bestBin) tmp[l++] = q, lb[ch].add( q );
					else tmp[r++] = q, rb[ch].add( q );
				}
			}
#pragma omp taskloop default( shared ) grainsize( 1 )
			for (int ch = 0; ch < nch; ch++) memcpy( (void*)( rec + lo_of( ch ) ), (const void*)( tmp + lo_of( ch ) ), ( lo_of( ch + 1 ) - lo_of( ch ) ) * sizeof( Rec ) );
			free( tmp );
			for (int ch = 0; ch < nch; ch++) L.add( lb[ch] ), R.add( rb[ch] );
			haveBounds = true;
		}
	}
	else if (node.count <= 4) return;
	if (leftCount == 0 || leftCount == (int)node.count)
	{
		if (node.count <= 4) return;
		// all centroids in one bin (or coincident): object median on the widest centroid axis
		int axis = 0;
		const float3 ce = cmx - cmn;
		if (ce.y > ce.x) axis = 1;
		if (ce.z > ce[axis]) axis = 2;
		leftCount = node.count / 2;
		std::nth_element( rec, rec + leftCount, rec + n, [&]( const Rec& a, const Rec& d ) {
			return a.c[axis] < d.c[axis] || ( a.c[axis] == d.c[axis] && a.id < d.id ); } );
		haveBounds = false;
	}
	const int l = c.used.fetch_add( 2 ), r = l + 1;
	c.pool[l].first = node.first, c.pool[l].count = leftCount, c.pool[l].left = -1;
	c.pool[r].first = node.first + leftCount, c.pool[r].count = node.count - leftCount, c.pool[r].left = -1;
	node.left = l;
	if (!haveBounds)
	{
		L = Side(), R = Side();
		for (int i = 0; i < leftCount; i++) L.add( rec[i] );
		for (uint i = (uint)leftCount; i < n; i++) R.add( rec[i] );
	}
	c.pool[l].mn = L.box.bmin, c.pool[l].mx = L.box.bmax, c.pool[r].mn = R.box.bmin, c.pool[r].mx = R.box.bmax;
	c.cb[2 * l] = L.cen.bmin, c.cb[2 * l + 1] = L.cen.bmax, c.cb[2 * r] = R.cen.bmin, c.cb[2 * r + 1] = R.cen.bmax;
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
} // namespace

void BVH::BuildBVHBinned()
{
	const bool timing = getenv( "RTMI_BUILD_TIMING" ) != 0;
	double t0 = omp_get_wtime(), t1;
	auto lap = [&]( const char* what ) { t1 = omp_get_wtime(); if (timing) fprintf( stderr, "rtmi build: %-10s %.3f s\n", what, t1 - t0 ); t0 = t1; };
	Allocate();
	lap( "allocate" );
	const int n = spheres_size + triangles_size;
	// untouched memory: a node's fields are written when the node is created
	TmpNode* pool = (TmpNode*)malloc( ( (size_t)n * 2 + 2 ) * sizeof( TmpNode ) );
	Rec* rec = (Rec*)malloc( (size_t)( n > 0 ? n : 1 ) * sizeof( Rec ) );
	const int nch = std::max( 1, n / CHUNK );
	std::vector<float3> mn( nch ), mx( nch ), cn( nch ), cx( nch );
	float3* cb = (float3*)malloc( ( (size_t)n * 2 + 2 ) * 2 * sizeof( float3 ) );
#pragma omp parallel for schedule( dynamic, 1 )
	for (int ch = 0; ch < nch; ch++)
	{
		const int lo = (int)( (size_t)n * ch / nch ), hi = (int)( (size_t)n * ( ch + 1 ) / nch );
		for (int i = lo; i < hi; i++)
		{
			Rec& q = rec[i];
			PrimBounds( i, q.mn, q.mx );
			q.c = float3( PrimCentroid( this, i, 0 ), PrimCentroid( this, i, 1 ), PrimCentroid( this, i, 2 ) ), q.id = i;
		}
		RecBounds( rec + lo, (uint)( hi - lo ), mn[ch], mx[ch] );
		CentroidBounds( rec + lo, (uint)( hi - lo ), cn[ch], cx[ch] );
	}
	pool[0].first = 0, pool[0].count = n, pool[0].left = -1;
	pool[0].mn = float3( 1e30f ), pool[0].mx = float3( -1e30f );
	cb[0] = float3( 1e30f ), cb[1] = float3( -1e30f );
	for (int ch = 0; ch < nch; ch++) pool[0].mn = fminf3( pool[0].mn, mn[ch] ), pool[0].mx = fmaxf3( pool[0].mx, mx[ch] ), cb[0] = fminf3( cb[0], cn[ch] ), cb[1] = fmaxf3( cb[1], cx[ch] );
	lap( "records" );
	BinnedCtx c{ rec, pool, cb, { 1 } };
#pragma omp parallel
#pragma omp single
	SubdivideBinnedTmp( c, 0, 0 );
	lap( "subdivide" );
#pragma omp parallel for schedule( static )
	for (int i = 0; i < n; i++) primitiveIdx[i] = rec[i].id;
	free( rec ), free( cb );
	Renumber( this, pool );
	free( pool );
	lap( "renumber" );
	FlattenForGPU();
	lap( "flatten" );
}

} // namespace rtmi
