// rtmi.hip -- librtmi.so: the C ABI of include/rtmi.h over hand-written gfx950 kernels.
//
// Wavefront path tracer (north star): per render batch of S samples x W*H pixels
//   k_primary           ray generation: camera ray, per-(pixel, sample) rng state, analytic part of FindNearest,
//                       root box test                                                      (camera.h:28-35, scene.h:359-368)
//   per bounce:
//   k_trace<closest>    persistent waves walk the BVH for the rays that entered its root box (bvh.h:507-530): 4-wide nodes
//                       with quantised boxes built on upload, every hit confirmed against the exact box of its leaf
//                       (rt_device.h: wide_step); rays with an infinite reciprocal direction walk the reference's pairs
//   k_shade             material evaluation, NEE, Russian roulette (renderer.cpp:246-365); CREATES the next ray and the
//                       shadow ray and runs their analytic intersections + root box tests; survivors are written in
//                       queue order to the other state set, BVH walkers are appended to the walk / shadow queues with
//                       one atomic per 1024-thread block (ballot + mbcnt prefix per wave, LDS scan across waves)
//   k_trace<any>        any-hit walk of the shadow queue, E[slot] += contribution          (scene.h:369-379, renderer.cpp:339-344);
//                       on a side stream, beside the next bounce's k_trace<closest> (render_batch)
//   k_carry_out / _in   the paths still alive when a batch leaves the bounce loop are late (>= 7 vertices, META_AGE_LATE: they add to an
//                       exact per-pixel sum, see Sinks) and join the next batch's queue; k_tail runs what is left of a
//                       render's paths to their ends when the caller reads the image
//   k_accumulate        accumulator[p] += E[sample][p] in sample order                    (renderer.cpp:566-567)
//   k_finalize, k_gamma, k_chroma, k_vignette                                             (precomp.h:431-465, postprocessing.h)
//   k_whitted           Renderer::Trace for the non-path-tracing modes, one thread per pixel with an explicit
//                       frame stack instead of recursion                                  (renderer.cpp:80-244)
// No host read-back inside a bounce loop iteration: grids are sized from queue counts that arrive two bounces late
// (counts never grow), kernels read the live count from device memory.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared   (see raytracer_amd/build.py)
#include "rt_device.h"
#include "../../include/rtmi.h"
#include <string>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <dlfcn.h>
#include <chrono>

#define BLOCK 256   // traversal and per-pixel kernels
#define SBLOCK 1024 // shading / ray-generation kernels: one queue-append atomic per 1024 paths (see block_append)

// ---- per-path state in QUEUE ORDER (entry i of the current bounce), ping-ponged between bounces: k_shade reads entry i
// with fully coalesced loads and writes the survivors to consecutive entries of the other set ---------------------------
struct PathState
{
	float4* o_t;    // ray origin, hit distance
	float4* d_h;    // ray direction, hit object index (int bits)
	float4* T_s;    // throughput, rng state (uint bits)
	uint32_t* meta; // slot (= sample_in_batch * W*H + pixel, RTMI_SLOT_BITS bits) | age | META_SUSPENDED | flags << 30 (see below)
};
struct ShadowQueue // only the shadow rays that have to walk the BVH
{
	float4* o_t;   // origin, tmax
	float4* d_s;   // direction, slot (uint bits)
	float4* c;     // contribution if unoccluded
};
struct CamProp { float4 pos, tl, tr, bl; };

struct DevCounters // the first 64 bytes are zeroed per render batch
{
	uint32_t n_state[2];   // ping-pong path queue lengths
	uint32_t n_walk[2];    // rays of that queue that enter the BVH root box (indices in wq[])
	uint32_t n_shadow[2];  // ping-pong shadow queue lengths
	uint32_t n_rec[2];     // suspend records written by k_trace<closest> of that bounce parity
	uint32_t n_srec[2];    // same for k_trace<any>
	uint32_t pool_extend;  // next unclaimed entry of the running k_trace<closest> launch (persistent waves pull chunks from it)
	uint32_t pool_connect; // same for k_trace<any>
	uint32_t pending;      // paths + carried shadow rays that the next bounce starts with (host loop termination)
	uint32_t pool_tail;    // next unclaimed entry of the running k_tail round
	uint32_t tail_mode;    // | TAIL_WAITING: the host waits for the tails (the pinned flag, read once per round); which kernel the running tail round belongs to (k_tail_decide): 0 k_tail, 1 k_tail_team, 2 k_tail_wave
	uint32_t pad[1];
	unsigned long long extend_rays, shadow_rays, extend_pairs, extend_prims, shadow_pairs, shadow_prims, extend_walkers, shadow_walkers, suspended;
	uint32_t max_steps[2]; // longest single walk (node pairs + leaf records), closest / any; counting instantiation only
	unsigned long long tail_prof[12]; // option tail_profile: see rtmi_stats.tail_profile
};

#define FLAG_SPECULAR 1u
#define FLAG_INSIDE 2u
#define TAIL_WAITING 0x100u
// The meta word: slot | age << RTMI_SLOT_BITS | META_SUSPENDED (bit 29) | flags << 30.  26 slot bits = batches of up to 2^26 paths
// (32 spp at 1080p) and 3 bits of age: a path is LATE from its 7th vertex.  (25 bits: 16 spp, late from the 15th vertex -- round 1's
// layout.  Measured, 20 bench steps: one lane 9 547 -> 10 360 Mrays/s, two lanes 10 945 -> 11 165; the larger launches lift the
// closest-hit walk from 0.60 to 0.73-0.77 of its algorithmic roofline; twice the queue memory.)
#ifndef RTMI_SLOT_BITS
#define RTMI_SLOT_BITS 26
#endif
#define META_SLOT_MASK ( ( 1u << RTMI_SLOT_BITS ) - 1u )  // a batch holds at most 2^RTMI_SLOT_BITS paths
#define META_AGE_SHIFT RTMI_SLOT_BITS           // vertices behind the path, saturating at META_AGE_LATE
#define META_AGE_LATE ( ( 1u << ( 29 - RTMI_SLOT_BITS ) ) - 1u ) // the age field fills the bits between the slot and META_SUSPENDED
#define SQ_LATE 0x40000000u         // shadow queue slot word: the shadow ray left a late vertex
#define META_SUSPENDED 0x20000000u // the closest-hit walk of this path was parked (k_trace); k_shade passes it through
#define RESUME_BIT 0x80000000u     // walk queue entry / shadow slot field: resume from a suspend record
#define REC_HEADER 6               // suspend record: cur, sp, k, h.obj, h.pos, -, then the stack

// ---- queue append.  One device-scope atomic on a single word retires at ~88 per microsecond on this chip, so a
// per-wave atomic (half a million waves per bounce) is the bottleneck of a wavefront path tracer; here a whole
// 1024-thread block (16 waves) reserves its entries with ONE atomic per queue: ballot + mbcnt prefix inside the wave,
// a 16-entry scan in LDS across waves. ----------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void block_append( const bool ( &push )[NQ], uint32_t* const ( &counters )[NQ], uint32_t ( &idx )[NQ], uint32_t* lds /* NQ * 16 */ )
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	uint32_t rank[NQ];
#pragma unroll
	for (int q = 0; q < NQ; q++)
	{
		const unsigned long long mask = __ballot( push[q] );
		rank[q] = __builtin_amdgcn_mbcnt_hi( (uint32_t)( mask >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)mask, 0 ) );
		if (lane == 0) lds[q * 16 + wave] = (uint32_t)__popcll( mask );
	}
	__syncthreads();
	if (threadIdx.x < NQ)
	{
		const int q = threadIdx.x;
		uint32_t total = 0;
		for (int w = 0; w < nw; w++) { const uint32_t c = lds[q * 16 + w]; lds[q * 16 + w] = total; total += c; }
		const uint32_t base = total ? atomicAdd( counters[q], total ) : 0u;
		for (int w = 0; w < nw; w++) lds[q * 16 + w] += base;
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < NQ; q++) idx[q] = lds[q * 16 + wave] + rank[q];
}
__device__ __forceinline__ void wave_add64( unsigned long long v, unsigned long long* dst )
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down( v, o, 64 );
	if (( threadIdx.x & 63 ) == 0 && v) atomicAdd( dst, v );
}

// ---- primary ray, camera.h:28-35 / generatePrimaryRays.cl:4-33 -------------------------------------------------
__device__ __forceinline__ void primary_ray( const CamProp& cam, float x, float y, float invW, float invH, f3& O, f3& D )
{
	const float u = x * invW, v = y * invH;
	const f3 TL = mk3( cam.tl ), TR = mk3( cam.tr ), BL = mk3( cam.bl );
	const f3 P = TL + u * ( TR - TL ) + v * ( BL - TL );
	O = mk3( cam.pos );
	D = normalize3( P - O );
}

// The part of Scene::FindNearest that needs no BVH walk, done by the kernel that CREATES the ray while it still has it
// in registers (every lane busy): quads[0], cubes[0], planes (scene.h:361-362, bvh.h:498-502), the reciprocal
// direction and the root box test.  Returns true when the ray has to walk the BVH.
__device__ __forceinline__ bool begin_closest( const DScene& s, f3 O, f3 D, RayHit& h, float t0 = 1e34f )
{
	h.t = t0, h.obj = -1, h.pos = -1;
	isect_analytic( s, O, D, h );
	if (s.n_bvh_prims == 0) return false;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	return ( inexact ? bvh_enter<true>( s, O, rD, h.t ) : bvh_enter<false>( s, O, rD, h.t ) ) != RT_DONE;
}
// Same for Scene::IsOccluded (scene.h:369-379): quad and cube with the shrinking ray.t, then the planes with early out.
__device__ __forceinline__ void begin_any( const DScene& s, f3 O, f3 D, float tmaxRay, bool& occluded, bool& walk )
{
	RayHit h; h.t = tmaxRay, h.obj = -1, h.pos = -1;
	if (s.n_quads > 0) isect_quad( s, 0, O, D, h );
	if (s.n_cubes > 0) isect_cube( s, 0, O, D, h );
	if (h.obj == -1) isect_planes( s, O, D, h );
	occluded = h.obj != -1, walk = false;
	if (occluded || s.n_bvh_prims == 0) return;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	walk = ( inexact ? bvh_enter<true>( s, O, rD, h.t ) : bvh_enter<false>( s, O, rD, h.t ) ) != RT_DONE;
}

// ---- ray generation (bounce 0): camera ray, per-(pixel, sample) rng state, analytic part of FindNearest --------------
__global__ __launch_bounds__( SBLOCK ) void k_primary( DScene s, PathState st, float4* __restrict__ E, float4* __restrict__ Es, uint32_t N, CamProp cam, int W, int H, float invW, float invH,
	const uint32_t* __restrict__ seeds, int sample_begin, int sample_stride, uint32_t* __restrict__ wq, DevCounters* ctr )
{
	__shared__ uint32_t lds[16];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	bool walk = false;
	if (i < N)
	{
		const uint32_t WH = (uint32_t)( W * H ), p = i % WH, sidx = (uint32_t)sample_begin + ( i / WH ) * (uint32_t)sample_stride;
		f3 O, D;
		primary_ray( cam, (float)(int)( p % (uint32_t)W ), (float)(int)( p / (uint32_t)W ), invW, invH, O, D );
		RayHit h;
		walk = begin_closest( s, O, D, h );
		st.o_t[i] = make_float4( O.x, O.y, O.z, h.t );
		st.d_h[i] = make_float4( D.x, D.y, D.z, __int_as_float( h.obj ) );
		st.T_s[i] = make_float4( 1, 1, 1, __uint_as_float( sample_seed( seeds[p], sidx ) ) );
		st.meta[i] = i | ( FLAG_SPECULAR << 30 );
		E[i] = make_float4( 0, 0, 0, 0 );
		Es[2 * (size_t)i] = make_float4( 0, 0, 0, 0 ), Es[2 * (size_t)i + 1] = make_float4( 0, 0, 0, 0 ); // 4 x int64 zero
	}
	const bool push[1] = { walk };
	uint32_t* const counters[1] = { &ctr->n_walk[0] };
	uint32_t idx[1];
	block_append<1>( push, counters, idx, lds );
	if (walk) wq[idx[0]] = i;
	if (i == 0) ctr->n_state[0] = N;
}

// ---- BVH walk: persistent waves with lane refill -----------------------------------------------------------------
// Most rays of a bounce never enter the BVH (they end on a wall, the light or nothing) and the ones that do have very
// different traversal lengths, so one-ray-per-thread launches leave most lanes of a wave idle.  Here the rays that
// passed the root box test (queued by the kernel that created them) are pulled by a fixed number of resident waves:
// a wave reserves a CHUNK of the queue with one atomic, and whenever `refill` of its lanes are idle they take the next
// rays of the chunk, so lanes stay busy and the pool counter sees one atomic per few hundred rays.
//   ANY = false: closest hit (IntersectBVH, bvh.h:507-530); entry = index into the path state; the analytic hit found
//                by the creating kernel is the starting t.
//   ANY = true:  any hit (IntersectBVHShadowRay, bvh.h:640-688) over the shadow queue; E[slot] += contribution when
//                nothing is hit (renderer.cpp:339-344).
// Contribution of a shadow ray that had to walk the BVH.  These arrive in an order that depends on scheduling (and, with
// parked rays, a bounce late), so they are not added into the float chain E[slot] but summed EXACTLY: 64-bit fixed point,
// 2^-40 units, integer atomics.  Integer addition is associative, so the sum does not depend on arrival order and no
// read-modify-write of a float is shared between lanes; k_accumulate adds float(Es) to the path's E once.
// (A contribution below 2^-40 ~ 1e-12 is dropped.)
#define ES_SCALE 1099511627776.0f // 2^40
// (Range +-2^22.  A contribution that is not finite or larger goes to odd[pixel], a float4 per pixel that is touched by
// float atomics ONLY -- no plain read-modify-write of another kernel can lose it -- and is added to the accumulator by
// k_fold_late at the next read, when nothing else is in flight: it stays visible, as NaN / inf / a huge value.)
__device__ __forceinline__ void add_odd( float4* odd, uint32_t pixel, float x, float y, float z )
{
	atomicAdd( &odd[pixel].x, x ), atomicAdd( &odd[pixel].y, y ), atomicAdd( &odd[pixel].z, z );
}
__device__ __forceinline__ void add_shadow_contribution( long long* Es, float4* odd, uint32_t WH, uint32_t slot, float4 c )
{
	const bool finite = fabsf( c.x ) < 4194304.0f && fabsf( c.y ) < 4194304.0f && fabsf( c.z ) < 4194304.0f; // false for NaN
	if (finite)
	{
		unsigned long long* p = (unsigned long long*)( Es + 4 * (size_t)slot );
		const long long x = (long long)( c.x * ES_SCALE ), y = (long long)( c.y * ES_SCALE ), z = (long long)( c.z * ES_SCALE );
		if (x) atomicAdd( p, (unsigned long long)x );
		if (y) atomicAdd( p + 1, (unsigned long long)y );
		if (z) atomicAdd( p + 2, (unsigned long long)z );
	}
	else add_odd( odd, slot % WH, c.x, c.y, c.z );
}

// Where a path's radiance goes.  Up to its 7th vertex (META_AGE_LATE) a path adds into its own slot: E[slot] is the float chain of
// Renderer::Sample for that sample (light hits, shadow rays that needed no walk), Es[slot] the exact sum of its walked
// shadow rays; k_accumulate adds both to the pixel in sample order.  From the 7th vertex on (a LATE vertex: the age is
// a property of the path, not of the launch that happens to process it) every contribution goes to late[pixel], an exact
// 64-bit fixed-point sum per pixel that is folded into the accumulator when the caller next reads it.  A late path owes
// its batch nothing any more, so the batch's slot can be accumulated and reused while the path is still bouncing -- in
// the queues of the following batches (k_carry_out / k_carry_in), at wavefront efficiency, instead of in a tail of its own.
struct Sinks { float4* E; long long* Es; long long* late; float4* odd; uint32_t WH; };
__device__ __forceinline__ void add_late( const Sinks& k, uint32_t slot, f3 c )
{
	const uint32_t p = slot % k.WH;
	const bool finite = fabsf( c.x ) < 4194304.0f && fabsf( c.y ) < 4194304.0f && fabsf( c.z ) < 4194304.0f; // false for NaN
	if (finite)
	{
		unsigned long long* q = (unsigned long long*)( k.late + 4 * (size_t)p );
		const long long x = (long long)( c.x * ES_SCALE ), y = (long long)( c.y * ES_SCALE ), z = (long long)( c.z * ES_SCALE );
		if (x) atomicAdd( q, (unsigned long long)x );
		if (y) atomicAdd( q + 1, (unsigned long long)y );
		if (z) atomicAdd( q + 2, (unsigned long long)z );
	}
	else add_odd( k.odd, p, c.x, c.y, c.z );
}
// a contribution of the float chain (light hit, unoccluded shadow ray without a walk)
__device__ __forceinline__ void add_chain( const Sinks& k, uint32_t slot, bool late, f3 c )
{
	if (late) { add_late( k, slot, c ); return; }
	float4 e = k.E[slot];
	e.x = e.x + c.x, e.y = e.y + c.y, e.z = e.z + c.z;
	k.E[slot] = e;
}
// a walked shadow ray's contribution
__device__ __forceinline__ void add_walked( const Sinks& k, uint32_t slot, bool late, float4 c )
{
	if (late) add_late( k, slot, mk3( c ) );
	else add_shadow_contribution( k.Es, k.odd, k.WH, slot, c );
}

// Tail control: a launch lasts as long as its longest ray, and a few rays of this scene walk > 1500 steps against a
// mean of 90, so from the third bounce on most of a launch was one wave finishing one ray.  Once a wave has no new
// rays to fetch, a CLOSEST-HIT ray that is still walking `budget` steps later is SUSPENDED: its traversal state (node,
// leaf progress, stack, best hit) goes to a record in HBM and the ray resumes in the next bounce's launch of this
// kernel, next to that bounce's new rays; its path simply lags one bounce.  Per-path arithmetic and event order are
// unchanged, so results do not depend on which rays were parked.  A parked SHADOW ray is carried into the next
// bounce's shadow queue; its contribution then arrives late and may meet the same path's next shadow ray in one launch,
// which is why the contributions of walked shadow rays are summed order-independently (Es, below).
struct TraceArgs
{
	PathState st;            // closest: the rays' state set
	ShadowQueue sq, sq_out;  // any: this bounce's queue; the next bounce's queue (receives the parked shadow rays)
	Sinks sinks;             // any: where the contributions of unoccluded shadow rays go
	const uint32_t* wq;      // closest: walk queue (entry | RESUME_BIT)
	const uint32_t* count_ptr;
	const uint32_t* rays_ptr;
	uint32_t* pool;
	const uint32_t* rec_in;  // records written by the previous bounce
	uint32_t* rec_out;       // records written by this launch
	uint32_t* n_rec_out;
	const uint32_t* n_rec_in;
	uint32_t* n_shadow_out;
	uint32_t rec_cap;
	int rec_stride, refill, budget; // budget 0: never suspend
};

template <bool ANY, bool COUNT>
__global__ __launch_bounds__( BLOCK ) void k_trace( DScene s, TraceArgs a, DevCounters* ctr )
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t* stack = lds_stack + threadIdx.x;
	const uint32_t n = *a.count_ptr;
	const int lane = threadIdx.x & 63;
	const int refill = a.refill;
	// interior nodes: the 4-wide quantised records where they were built; rays with an infinite reciprocal direction
	// component (`inexact`) and the counting pass (which counts the reference's nodes) walk the pairs
	const bool use_wide = !COUNT && s.wide != nullptr;
	// chunk: large enough that the pool counter is not the bottleneck, small enough that the last chunks balance
	const uint32_t waves = gridDim.x * ( BLOCK / 64 );
	uint32_t chunk = n / ( waves * 4u );
	chunk = chunk < 64u ? 64u : ( chunk > 512u ? 512u : chunk );
	uint32_t chunk_next = 0, chunk_end = 0; // wave-uniform
	// lane state: !has = empty; has && cur == RT_DONE = finished, waiting to be retired; else walking
	bool has = false, exhausted = false, inexact = false, occluded = false, resumed = false;
	uint32_t cur = RT_DONE, entry = 0, slot = 0;
	int sp = 0, k = 0; // stack pointer; primitives of the current leaf already tested
	int tail_steps = 0; // steps since this wave ran out of new rays
	f3 O = mk3( 0, 0, 0 ), D = mk3( 0, 0, 1 ), rD = mk3( 0, 0, 1 );
	RayHit h; h.t = 0, h.obj = -1, h.pos = -1;
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	uint32_t steps = 0, maxSteps = 0, nSusp = 0; // COUNT only (nSusp always)
	for (;;)
	{
		const unsigned long long walking = __ballot( has && cur != RT_DONE );
		if (walking == 0 || 64 - __popcll( walking ) >= refill || ( exhausted && a.budget > 0 && tail_steps >= a.budget ))
		{
			// retire finished rays
			if (has && cur == RT_DONE)
			{
				if (ANY)
				{
					if (!occluded) add_walked( a.sinks, slot & META_SLOT_MASK, ( slot & SQ_LATE ) != 0, a.sq.c[entry] ); // E += T * (...) * BRDF * emission
				}
				else if (h.pos >= 0 || resumed) // a BVH primitive beat the analytic hit (a resumed ray's fields held its record)
				{
					a.st.o_t[entry].w = h.t;
					a.st.d_h[entry].w = __int_as_float( h.obj );
				}
				has = false, inexact = false;
				if (COUNT) maxSteps = steps > maxSteps ? steps : maxSteps, steps = 0;
			}
			// hand the free lanes new rays
			if (!exhausted)
			{
				const unsigned long long idle = ~walking;
				const int nIdle = __popcll( idle );
				if (chunk_next == chunk_end)
				{
					uint32_t base = 0;
					if (lane == 0) base = atomicAdd( a.pool, chunk );
					base = __shfl( base, 0, 64 );
					if (base >= n) exhausted = true, chunk_next = chunk_end = 0;
					else chunk_next = base, chunk_end = ( n - base < chunk ) ? n : base + chunk;
				}
				if (!exhausted)
				{
					const uint32_t my = chunk_next + __builtin_amdgcn_mbcnt_hi( (uint32_t)( idle >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)idle, 0 ) );
					if (!has && my < chunk_end)
					{
						has = true, sp = 0, k = 0, occluded = false, h.pos = -1, cur = s.root_ref; // the root box test passed in the creating kernel
						uint32_t rec = 0;
						if (ANY)
						{
							entry = my;
							const float4 ot = a.sq.o_t[entry], ds = a.sq.d_s[entry];
							O = mk3( ot ), D = mk3( ds ), h.t = ot.w, h.obj = -1, slot = __float_as_uint( ds.w );
							resumed = ( slot & RESUME_BIT ) != 0, slot &= ~RESUME_BIT;
							if (resumed) rec = __float_as_uint( a.sq.c[entry].w );
						}
						else
						{
							const uint32_t e = a.wq[my];
							entry = e & ~RESUME_BIT, resumed = ( e & RESUME_BIT ) != 0;
							const float4 ot = a.st.o_t[entry], dh = a.st.d_h[entry];
							O = mk3( ot ), D = mk3( dh ), h.t = ot.w, h.obj = __float_as_int( dh.w ); // analytic hit so far
							if (resumed) rec = __float_as_uint( dh.w ), a.st.meta[entry] &= ~META_SUSPENDED;
						}
						if (resumed)
						{
							const uint32_t* r = a.rec_in + (size_t)rec * a.rec_stride;
							cur = r[0], sp = (int)r[1], k = (int)r[2];
							if (!ANY) h.obj = (int)r[3], h.pos = (int)r[4];
							for (int i = 0; i < sp; i++) stack[i * BLOCK] = r[REC_HEADER + i];
						}
						rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
						inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
						if (use_wide && !inexact && !resumed) cur = s.wide_root;
					}
					chunk_next = ( chunk_end - chunk_next < (uint32_t)nIdle ) ? chunk_end : chunk_next + (uint32_t)nIdle;
					if (chunk_next == chunk_end && chunk_end == n) exhausted = true;
				}
				if (exhausted) tail_steps = 0;
			}
			else if (a.budget > 0 && tail_steps >= a.budget)
			{
				// park the rays that are still walking
				const bool want = has && cur != RT_DONE;
				const unsigned long long m = __ballot( want );
				if (m)
				{
					const int leader = __ffsll( (long long)m ) - 1;
					const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( m >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)m, 0 ) );
					uint32_t base = 0;
					if (lane == leader) base = atomicAdd( a.n_rec_out, (uint32_t)__popcll( m ) );
					const uint32_t idx = __shfl( base, leader, 64 ) + rank;
					const bool park = want && idx < a.rec_cap; // no room: keep walking for another budget
					uint32_t si = 0;
					if (ANY) // the parked shadow rays join the next bounce's shadow queue
					{
						const unsigned long long mp = __ballot( park );
						if (mp)
						{
							const int l2 = __ffsll( (long long)mp ) - 1;
							uint32_t b2 = 0;
							if (lane == l2) b2 = atomicAdd( a.n_shadow_out, (uint32_t)__popcll( mp ) );
							si = __shfl( b2, l2, 64 ) + __builtin_amdgcn_mbcnt_hi( (uint32_t)( mp >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)mp, 0 ) );
						}
					}
					if (park)
					{
						uint32_t* r = a.rec_out + (size_t)idx * a.rec_stride;
						r[0] = cur, r[1] = (uint32_t)sp, r[2] = (uint32_t)k, r[3] = (uint32_t)h.obj, r[4] = (uint32_t)h.pos;
						for (int i = 0; i < sp; i++) r[REC_HEADER + i] = stack[i * BLOCK];
						if (ANY)
						{
							const float4 c = a.sq.c[entry];
							a.sq_out.o_t[si] = make_float4( O.x, O.y, O.z, h.t );
							a.sq_out.d_s[si] = make_float4( D.x, D.y, D.z, __uint_as_float( slot | RESUME_BIT ) );
							a.sq_out.c[si] = make_float4( c.x, c.y, c.z, __uint_as_float( idx ) );
						}
						else
						{
							a.st.o_t[entry].w = h.t;
							a.st.d_h[entry].w = __uint_as_float( idx );
							a.st.meta[entry] |= META_SUSPENDED;
						}
						has = false, inexact = false, nSusp++;
						if (COUNT) maxSteps = steps > maxSteps ? steps : maxSteps, steps = 0;
					}
					tail_steps = 0;
				}
			}
			if (__ballot( has ) == 0)
			{
				if (exhausted) break;
				continue;
			}
		}
		tail_steps++;
		// one step per iteration for the majority kind: the lanes at an interior node test its two children, or the
		// lanes in a leaf test one primitive.  The minority waits; it becomes the majority as the others arrive.
		const bool atNode = has && !( cur & RT_LEAF ), atLeaf = has && ( cur & RT_LEAF ) && cur != RT_DONE;
		if (__popcll( __ballot( atNode ) ) >= __popcll( __ballot( atLeaf ) ))
		{
			if (use_wide)
			{
				if (atNode && !inexact) cur = wide_step<ANY>( s, cur, O, rD, h.t, stack, BLOCK, sp );
				if (__ballot( atNode && inexact ) != 0) { if (atNode && inexact) cur = interior_step<ANY, true>( s, cur, O, rD, h.t, stack, BLOCK, sp ); }
			}
			else if (__ballot( has && inexact ) != 0)
			{
				if (atNode) { if (COUNT) tc.pairs++, steps++; cur = interior_step<ANY, true>( s, cur, O, rD, h.t, stack, BLOCK, sp ); }
			}
			else if (atNode) { if (COUNT) tc.pairs++, steps++; cur = interior_step<ANY, false>( s, cur, O, rD, h.t, stack, BLOCK, sp ); }
		}
		else if (atLeaf)
		{
			const int first = (int)( cur & RT_LEAF_FIRST_MASK ), count = (int)( ( cur >> RT_LEAF_FIRST_BITS ) & ( RT_LEAF_MAX_COUNT - 1 ) ) + 1;
			if (COUNT) tc.prims++, steps++;
			const float t0 = h.t;
			const int obj0 = h.obj, pos0 = h.pos;
			bool hit = isect_prim<ANY>( s, first + k, O, D, h );
			if (hit && use_wide && !inexact)
			{
				// reached through quantised boxes: the hit counts only if the leaf's exact box passes the reference's test
				const float4 b0 = s.leafbox[2 * first], b1 = s.leafbox[2 * first + 1];
				float tmn, tmx;
				slab<false>( b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, O, rD, tmn, tmx );
				if (!( tmx >= tmn && ( ANY ? tmn < t0 : tmn <= t0 ) && tmx > 0 )) hit = false, h.t = t0, h.obj = obj0, h.pos = pos0;
			}
			k++;
			if (ANY && hit) occluded = true, cur = RT_DONE, k = 0;
			else if (k == count) k = 0, cur = sp > 0 ? stack[--sp * BLOCK] : RT_DONE;
		}
	}
	if (COUNT)
	{
		wave_add64( tc.pairs, ANY ? &ctr->shadow_pairs : &ctr->extend_pairs );
		wave_add64( tc.prims, ANY ? &ctr->shadow_prims : &ctr->extend_prims );
		for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_down( maxSteps, o, 64 ); maxSteps = v > maxSteps ? v : maxSteps; }
		if (lane == 0 && maxSteps) atomicMax( &ctr->max_steps[ANY ? 1 : 0], maxSteps );
	}
	if (__ballot( nSusp != 0 )) wave_add64( nSusp, &ctr->suspended );
	// every closest-hit ray of the bounce counts as an extend ray, whether it had to walk the BVH or not; a resumed
	// ray is not a new ray
	if (threadIdx.x == 0 && blockIdx.x == 0)
	{
		const uint32_t nres = *a.n_rec_in < a.rec_cap ? *a.n_rec_in : a.rec_cap; // resumed rays in this launch
		if (!ANY) atomicAdd( &ctr->extend_rays, (unsigned long long)( *a.rays_ptr - nres ) );
		atomicAdd( ANY ? &ctr->shadow_walkers : &ctr->extend_walkers, (unsigned long long)( n - nres ) );
	}
}

// ---- one path vertex: Renderer::Sample loop body after FindNearest (renderer.cpp:254-361) --------------------------
// In: the ray (O, D), its hit (t, obj), the path's throughput T, rng state, flags.  Adds a light hit to E[slot].
// Out: the next ray (push_next) and the NEE shadow ray (shadow_any) with its contribution.
struct Vertex
{
	bool push_next, shadow_any;
	f3 newO, newD, shO, shD, contrib;
	float shT;
};
// `emit`: this lane adds the light hit to the sinks (false for the three lanes of a k_tail_team team that only mirror the leader's arithmetic)
__device__ __forceinline__ void shade_vertex( const DScene& s, f3 O, f3 D, float t, int obj, f3& T, uint32_t& rng, uint32_t& flags, const Sinks& sinks, uint32_t slot, bool late, Vertex& v, bool emit = true )
{
	v.push_next = false, v.shadow_any = false, v.shT = 0;
	if (obj == -1) return; // miss: the path ends, Sample returns E (renderer.cpp:254, 363)
	const f3 I = O + t * D;
	HitAttr a;
	hit_attributes( s, obj, I, D, a );
	const f3 N = a.N;
	const int type = a.mat->type;
	if (type == RTMI_LIGHT)
	{
		if (flags & FLAG_SPECULAR) // renderer.cpp:262-263: E + T * emission
		{
			if (emit) add_chain( sinks, slot, late, T * mk3( s.light_emission ) );
		}
		return;
	}
	if (type == RTMI_MIRROR) // renderer.cpp:266-271
	{
		const f3 refl = D - ( 2 * dot3( D, N ) ) * N;
		v.newO = I + refl * 0.001f, v.newD = refl;
		flags = FLAG_SPECULAR, v.push_next = true;
		return;
	}
	if (type == RTMI_GLASS) // renderer.cpp:272-321
	{
		const float air = 1.0003f, ior = a.mat->ior;
		float n1, n2;
		f3 absorption = mk3( 1, 1, 1 );
		if (flags & FLAG_INSIDE)
		{
			const float4 ab = a.mat->absorption;
			absorption = mk3( expf( -ab.x * t ), expf( -ab.y * t ), expf( -ab.z * t ) );
			n1 = ior, n2 = air;
		}
		else n1 = air, n2 = ior;
		const float ratio = n1 / n2;
		const float inc = dot3( N, -D );
		const float k = 1.0f - sqrf_( ratio ) * ( 1.0f - sqrf_( inc ) );
		const f3 refr = ratio * D + N * ( ratio * inc - sqrtf( k ) );
		const float outc = dot3( -N, refr );
		const double left = sqrf_( ( n1 * inc - ior * outc ) / ( n1 * inc + n2 * outc ) );
		const double right = sqrf_( ( n1 * outc - ior * inc ) / ( n1 * outc + n2 * inc ) );
		const float Fr = (float)( 0.5 * ( left + right ) );
		T = T * ( a.albedo * absorption );
		bool reflect = k < 0;
		if (!reflect) reflect = rng_float( rng ) <= Fr; // short-circuit: no draw when k < 0
		if (reflect)
		{
			const f3 refl = D - ( 2.0f * dot3( D, N ) ) * N;
			v.newO = I + refl * 0.001f, v.newD = refl, flags = FLAG_SPECULAR;
		}
		else v.newO = I + refr * 0.001f, v.newD = refr, flags = FLAG_SPECULAR | FLAG_INSIDE; // renderer.cpp:317-318
		v.push_next = true;
	}
	else // diffuse (and SUBSTANCE), renderer.cpp:322-361
	{
		const f3 BRDF = a.albedo / 3.14159265358979323846264f;
		// Scene::RandomPointOnLight, scene.h:317-341
		const f3 c2 = mk3( s.light_c2 ), Nl = mk3( s.light_N );
		const f3 c1c2 = mk3( s.light_u ); // normalize( c1 - c2 ), rtmi_upload_scene
		float len = rng_float( rng ) * s.light_s;
		const f3 u = c1c2 * len;
		const f3 c2c3 = mk3( s.light_w ); // normalize( c3 - c2 )
		len = rng_float( rng ) * s.light_s;
		const f3 w = c2c3 * len;
		const f3 lp = c2 + u + w - mk3( 0, 0.01f, 0 );
		f3 L = lp - I;
		const float dist = magnitude3( L );
		L = normalize3( L );
		if (dot3( N, L ) > 0 && dot3( Nl, -L ) > 0)
		{
			const float solidAngle = ( dot3( Nl, -L ) * s.light_A ) / sqrf_( dist );
			const float lightPDF = 1.0f / solidAngle;
			v.contrib = T * ( dot3( N, L ) / lightPDF ) * BRDF * mk3( s.light_emission );
			v.shO = I + L * 0.001f, v.shD = L, v.shT = dist - 2.0f * 0.001f;
			v.shadow_any = true;
		}
		// Russian roulette, renderer.cpp:347-352 (template clamp = fmaxf(a, fminf(f, b)) with ternaries)
		float p = smax( a.albedo.z, smax( a.albedo.x, a.albedo.y ) );
		p = p < 1.0f ? p : 1.0f;
		p = 0.0f > p ? 0.0f : p;
		if (!( p < rng_float( rng ) ))
		{
			T = T * ( 1.0f / p );
			// Scene::DiffuseReflection, scene.h:517-532 (draw order x, y, z)
			f3 R;
			do
			{
				const float rx = rng_float( rng ) * 2.0f - 1.0f;
				const float ry = rng_float( rng ) * 2.0f - 1.0f;
				const float rz = rng_float( rng ) * 2.0f - 1.0f;
				R = mk3( rx, ry, rz );
			} while (magnitude3( R ) > 1.0f);
			R = normalize3( R );
			if (dot3( N, R ) < 0) R = -R;
			const float hemiPDF = 1.0f / ( 3.14159265358979323846264f * 2.0f );
			v.newO = I + R * 0.001f, v.newD = R;
			T = T * ( ( dot3( N, R ) / hemiPDF ) * BRDF );
			flags = 0, v.push_next = true;
		}
	}
	// A path whose throughput is exactly zero can never change E again (every later term is T * finite): ending it here
	// returns the same bits as following it.  The reference has no such test and follows glass-trapped paths for the
	// ~1400 interactions it takes 0.93^n to underflow, and then for as long as they stay trapped (renderer.cpp:306).
	if (T.x == 0.0f && T.y == 0.0f && T.z == 0.0f) v.push_next = false;
}

// ---- shade kernel: shade_vertex fused with the creation of the next ray and of the shadow ray: their analytic
// intersections and root box tests run here, so that a bounce is
//   k_trace<closest>  ->  k_shade  ->  k_trace<any>
// and rays that never reach the BVH cost no queue traffic at all. ------------------------------------------------------
__global__ __launch_bounds__( SBLOCK, 8 ) void k_shade( DScene s, PathState in, PathState out, ShadowQueue sq, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, uint32_t* __restrict__ wq_out, uint32_t* n_walk_out, uint32_t* n_shadow, DevCounters* ctr )
{
	__shared__ uint32_t lds[3 * 16 + 1];
	if (threadIdx.x == 0) lds[48] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n = *n_in;
	bool walk_next = false, push_shadow = false, parked = false;
	uint32_t slot = 0, flags = 0, rng = 0, age = 0;
	bool late = false;
	f3 T;
	Vertex v; v.push_next = false, v.shadow_any = false, v.shT = 0;
	RayHit nh; nh.t = 1e34f, nh.obj = -1, nh.pos = -1;
	if (i < n)
	{
		const float4 ot = in.o_t[i], dh = in.d_h[i], ts = in.T_s[i];
		const uint32_t meta = in.meta[i];
		slot = meta & META_SLOT_MASK, flags = meta >> 30, parked = ( meta & META_SUSPENDED ) != 0;
		age = ( meta >> META_AGE_SHIFT ) & META_AGE_LATE, late = age == META_AGE_LATE;
		const f3 O = mk3( ot ), D = mk3( dh );
		const float t = ot.w;
		const int obj = __float_as_int( dh.w );
		T = mk3( ts ), rng = __float_as_uint( ts.w );
		if (parked)
		{
			// the closest-hit walk of this path is not finished (k_trace parked it): hand the entry on unchanged; t holds
			// the best distance so far and the object field the index of its suspend record
			v.newO = O, v.newD = D, nh.t = t, nh.obj = obj, v.push_next = true, walk_next = true;
		}
		else
		{
			shade_vertex( s, O, D, t, obj, T, rng, flags, sinks, slot, late, v );
			age = age < META_AGE_LATE ? age + 1 : age; // the next vertex of this path
			// the rays created here start their life here: everything of FindNearest / IsOccluded before the BVH walk
			if (v.push_next) walk_next = begin_closest( s, v.newO, v.newD, nh );
			if (v.shadow_any)
			{
				bool occluded;
				begin_any( s, v.shO, v.shD, v.shT, occluded, push_shadow );
				if (!occluded && !push_shadow) add_chain( sinks, slot, late, v.contrib ); // unoccluded without a walk: E += contribution now (renderer.cpp:343)
			}
		}
	}
	{
		const unsigned long long m = __ballot( v.shadow_any );
		if (( threadIdx.x & 63 ) == 0 && m) atomicAdd( &lds[48], (uint32_t)__popcll( m ) );
	}
	const bool push[3] = { v.push_next, walk_next, push_shadow };
	uint32_t* const counters[3] = { n_out, n_walk_out, n_shadow };
	uint32_t idx[3];
	block_append<3>( push, counters, idx, lds );
	if (v.push_next)
	{
		const uint32_t j = idx[0];
		out.o_t[j] = make_float4( v.newO.x, v.newO.y, v.newO.z, nh.t );
		out.d_h[j] = make_float4( v.newD.x, v.newD.y, v.newD.z, __int_as_float( nh.obj ) );
		out.T_s[j] = make_float4( T.x, T.y, T.z, __uint_as_float( rng ) );
		out.meta[j] = slot | ( age << META_AGE_SHIFT ) | ( flags << 30 ) | ( parked ? META_SUSPENDED : 0u );
		if (walk_next) wq_out[idx[1]] = j | ( parked ? RESUME_BIT : 0u );
	}
	if (push_shadow)
	{
		const uint32_t si = idx[2];
		sq.o_t[si] = make_float4( v.shO.x, v.shO.y, v.shO.z, v.shT );
		sq.d_s[si] = make_float4( v.shD.x, v.shD.y, v.shD.z, __uint_as_float( slot | ( late ? SQ_LATE : 0u ) ) );
		sq.c[si] = make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 );
	}
	if (threadIdx.x == 0 && lds[48]) atomicAdd( &ctr->shadow_rays, (unsigned long long)lds[48] );
}

// ---- tail: once only a few thousand paths are alive a bounce is pure latency (three launches, each as long as one
// ray's walk), and paths trapped inside the mirror or glass object live for thousands of bounces (the reference has
// no bounce limit, renderer.cpp:251).  This kernel runs each remaining path for up to `max_bounces` bounces in one
// launch: one lane per path loops FindNearest -> shade_vertex -> IsOccluded.  Paths still alive are written compacted
// to the other state set for the next round, so that a round's waves are dense again (nearly every wave of the first
// round holds one long path, and a thousand almost-empty resident waves would starve the next batches' bulk kernels).
// Same device functions and the same per-path order of operations as the wavefront kernels: results do not depend on
// when the host switches over or on the round lengths. ------------------------------------------------------------------
template <bool PROF>
__global__ __launch_bounds__( 1024 ) void k_tail( DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, int max_bounces, uint32_t dense_above, uint32_t mid_above, int min_lanes, uint32_t lds_words, const uint32_t* draining, uint32_t team_below, DevCounters* ctr,
	int stop_late )
{
	extern __shared__ uint32_t lds_stack[];
	// per_wave paths per 64-lane wave, on every (64 / per_wave)-th lane.  A wave full of long paths pays for the divergence
	// of 64 different walks at every bounce (about 2.4x the latency per bounce); a wave with a few of them runs each at close
	// to single-lane latency but issues a whole wave instruction for 4 lanes: with 4 per wave throughout, the tails of this
	// scene (8 M path-bounces per 16-spp batch, 4 % of the bulk's) issued 42 % of all vector instructions on the chip and
	// the next batches' bulk kernels ran 1.6x longer beside them.  So the round decides from its live path count: full waves
	// while thousands are alive (throughput), 16 and then min_lanes per wave for the last few hundred (latency).
	// The grid is a bounded number of resident waves that pull per_wave paths at a time.
	const uint32_t n = *n_in;
	// `draining` (pinned host memory): the host has nothing more to enqueue and waits for the tails (rtmi_flush, read-backs):
	// the chip is this kernel's, so a round (round 0 excepted) spreads its paths over all the waves of the grid, as few to a
	// wave as that allows: a path's bounce waits for the walks of the others in its wave
	const uint32_t mode = ctr->tail_mode; // k_tail_decide's word for this round: the flag is read from host memory once, not by every wave
	const bool hurry = dense_above != 0 && ( mode & TAIL_WAITING ) != 0;
	if (team_below && ( mode & 3u ) != 0) return; // this round is k_tail_team's or k_tail_wave's (k_tail_decide; launched right after this kernel)
	unsigned long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const unsigned long long prof_t0 = PROF ? wall_clock64() : 0, prof_c0 = PROF ? __builtin_readcyclecounter() : 0;
	int per_wave;
	if (hurry)
	{
		const uint32_t waves = gridDim.x * ( blockDim.x >> 6 );
		const uint32_t need = ( n + waves - 1 ) / waves;
		per_wave = 1;
		while ((uint32_t)per_wave < need && per_wave < 64) per_wave *= 2;
	}
	else per_wave = n > dense_above ? 64 : ( n > mid_above ? ( min_lanes > 16 ? min_lanes : 16 ) : min_lanes );
	const uint32_t stride = 64u / (uint32_t)per_wave;
	// Workgroups of several waves: the waves of this kernel live for milliseconds, and a CU that hosts one serves the
	// blocks of the next batches' k_shade at about half speed (the old wave wins the issue arbitration; the block's barriers
	// wait for its slowest wave): 250 tail waves spread one to a CU slowed every k_shade launch by 1.5-2x.  Waves that the
	// round's live path count does not need leave at once, so the ones that stay are the first blocks of the grid: a few CUs.
	const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
	if (( blockIdx.x * waves_per_block + wave_in_block ) * (uint32_t)per_wave >= n) return; // wave-uniform
	uint32_t* stack = lds_stack + wave_in_block * ( lds_words / waves_per_block ) + lane / stride; // one column per active lane
	unsigned long long nExt = 0, nSh = 0, nExtWalk = 0, nShWalk = 0;
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	for (;;)
	{
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd( &ctr->pool_tail, (uint32_t)per_wave );
		base = __shfl( base, 0, 64 );
		if (base >= n) break; // wave-uniform
		const uint32_t i = base + lane / stride;
		bool alive = false;
		uint32_t slot = 0, flags = 0, rng = 0, age = 0;
		f3 O = mk3( 0, 0, 0 ), D = mk3( 0, 0, 1 ), T = mk3( 0, 0, 0 );
		if (i < n && lane % stride == 0)
		{
			const float4 ot = in.o_t[i], dh = in.d_h[i], ts = in.T_s[i];
			const uint32_t meta = in.meta[i];
			slot = meta & META_SLOT_MASK, flags = meta >> 30, rng = __float_as_uint( ts.w ), age = ( meta >> META_AGE_SHIFT ) & META_AGE_LATE;
			O = mk3( ot ), D = mk3( dh ), T = mk3( ts );
			alive = true;
			for (int nb = 0; nb < max_bounces; nb++)
			{
				// Scene::FindNearest from scratch (the analytic part stored with the entry is recomputed: same bits)
				RayHit h;
				unsigned long long tp = PROF ? wall_clock64() : 0;
				const bool walk = begin_closest( s, O, D, h );
				if (PROF) { const unsigned long long x = wall_clock64(); prof[0] += x - tp, tp = x, prof[4]++; }
				nExt++;
				if (walk)
				{
					nExtWalk++;
					const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
					const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
					if (inexact) traverse<false, true, false>( s, O, D, rD, h, stack, per_wave, tc );
					else if (s.wide) traverse_wide<false>( s, O, D, rD, h, stack, per_wave );
					else traverse<false, false, false>( s, O, D, rD, h, stack, per_wave, tc );
				}
				if (PROF) { const unsigned long long x = wall_clock64(); prof[1] += x - tp, tp = x; }
				Vertex v;
				const bool late = age == META_AGE_LATE;
				shade_vertex( s, O, D, h.t, h.obj, T, rng, flags, sinks, slot, late, v );
				age = age < META_AGE_LATE ? age + 1 : age;
				if (PROF) { const unsigned long long x = wall_clock64(); prof[2] += x - tp, tp = x; }
				if (v.shadow_any)
				{
					nSh++;
					bool occluded, swalk;
					begin_any( s, v.shO, v.shD, v.shT, occluded, swalk );
					if (!occluded && !swalk) add_chain( sinks, slot, late, v.contrib );
					else if (swalk)
					{
						nShWalk++;
						RayHit sh; sh.t = v.shT, sh.obj = -1, sh.pos = -1;
						const f3 rD = mk3( 1 / v.shD.x, 1 / v.shD.y, 1 / v.shD.z );
						const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
						const bool occ = inexact ? traverse<true, true, false>( s, v.shO, v.shD, rD, sh, stack, per_wave, tc )
							: s.wide ? traverse_wide<true>( s, v.shO, v.shD, rD, sh, stack, per_wave ) : traverse<true, false, false>( s, v.shO, v.shD, rD, sh, stack, per_wave, tc );
						if (!occ) add_walked( sinks, slot, late, make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
					}
				}
				if (PROF) { const unsigned long long x = wall_clock64(); prof[3] += x - tp, tp = x; }
				if (!v.push_next) { alive = false; break; }
				O = v.newO, D = v.newD;
				// (stop_late: the round of a batch's stragglers -- paths that lag behind their batch because walks of theirs were
				// parked -- ends for a path as soon as it is late: from then on it owes its batch nothing and rides the next ones)
				if (stop_late && age == META_AGE_LATE) break;
			}
		}
		// survivors of this round, compacted (one atomic per wave)
		const unsigned long long m = __ballot( alive );
		if (m)
		{
			const int leader = __ffsll( (long long)m ) - 1;
			uint32_t base = 0;
			if ((int)lane == leader) base = atomicAdd( n_out, (uint32_t)__popcll( m ) );
			const uint32_t j = __shfl( base, leader, 64 ) + __builtin_amdgcn_mbcnt_hi( (uint32_t)( m >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)m, 0 ) );
			if (alive)
			{
				out.o_t[j] = make_float4( O.x, O.y, O.z, 1e34f );
				out.d_h[j] = make_float4( D.x, D.y, D.z, __int_as_float( -1 ) );
				out.T_s[j] = make_float4( T.x, T.y, T.z, __uint_as_float( rng ) );
				out.meta[j] = slot | ( age << META_AGE_SHIFT ) | ( flags << 30 );
			}
		}
	} // next per_wave paths
	wave_add64( nExt, &ctr->extend_rays ), wave_add64( nSh, &ctr->shadow_rays );
	wave_add64( nExtWalk, &ctr->extend_walkers ), wave_add64( nShWalk, &ctr->shadow_walkers );
	if (PROF && hurry && n <= 16384u) // the profile covers the regime k_tail_team is for: draining rounds with few paths left
	{
		for (int k = 0; k < 5; k++) wave_add64( prof[k], &ctr->tail_prof[k] );
		if (lane == 0) atomicAdd( &ctr->tail_prof[7], wall_clock64() - prof_t0 ), atomicAdd( &ctr->tail_prof[8], __builtin_readcyclecounter() - prof_c0 );
	}
}

// Which kernel runs a tail round is decided ONCE per round, here, from the live path count and from whether the host is
// waiting (`draining`, pinned host memory that the host may flip at any time -- between two launches of one round too, which
// is why the three kernels do not each look for themselves: a round must have exactly one taker).
__global__ void k_tail_decide( DevCounters* ctr, const uint32_t* __restrict__ n_in, const uint32_t* draining, uint32_t team_below, uint32_t wave_below )
{
	const uint32_t n = *n_in;
	const bool waiting = *(const volatile uint32_t*)draining != 0;
	ctr->tail_mode = ( waiting && n <= wave_below ? 2u : waiting && n <= team_below ? 1u : 0u ) | ( waiting ? TAIL_WAITING : 0u );
}

// ---- the end of a render: teams of four lanes per path -------------------------------------------------------------------
// When the caller waits for the image the chip is empty except for the last paths, a few thousand at most, a handful of
// which bounce thousands of times inside the closed mirror object (the reference has no bounce limit, renderer.cpp:251): the
// wait is (bounces of the longest path) x (latency of one bounce).  One path alone in a wave pays about 4 cycles for every
// instruction of its walk, so here four neighbouring lanes walk ONE path together (rt_device.h: team_traverse_wide): lane j
// tests child j of every 4-wide node and primitive j of every leaf, the rest of the bounce (analytic tests, shading, the
// random stream) is computed identically by the four lanes, and lane 0 alone writes.  Same visiting order, same
// arithmetic, same results as k_tail, which stays in charge of every round that is not a draining round with few paths.
template <bool PROF>
__global__ __launch_bounds__( 256 ) void k_tail_team( DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, int max_bounces, const uint32_t* draining, uint32_t team_below, uint32_t wave_below, DevCounters* ctr )
{
	extern __shared__ uint32_t lds_stack[];
	const uint32_t n = *n_in;
	if (( ctr->tail_mode & 3u ) != 1) return; // k_tail's or k_tail_wave's round (k_tail_decide)
	// teams per wave: as few as the grid allows (a team's bounce waits for the divergence of the other teams of its wave)
	const uint32_t waves_per_block = blockDim.x >> 6, waves = gridDim.x * waves_per_block;
	const uint32_t need = ( n + waves - 1 ) / waves;
	uint32_t per_wave = 1;
	while (per_wave < need && per_wave < 16) per_wave *= 2;
	const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
	if (( blockIdx.x * waves_per_block + wave_in_block ) * per_wave >= n) return; // wave-uniform: surplus waves leave at once
	const int j = (int)( lane & 3u );
	const uint32_t tm = lane >> 2;
	const int stride = (int)( waves_per_block * 16u );
	uint32_t* stack = lds_stack + wave_in_block * 16u + tm; // one column per team
	unsigned long long nExt = 0, nSh = 0, nExtWalk = 0, nShWalk = 0;
	unsigned long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, probe[2] = { 0, 0 };
	const unsigned long long prof_t0 = PROF ? wall_clock64() : 0, prof_c0 = PROF ? __builtin_readcyclecounter() : 0;
	uint32_t nSteps = 0, nPrims = 0, nStepsAny = 0, nPrimsAny = 0;
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	for (;;)
	{
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd( &ctr->pool_tail, per_wave );
		base = __shfl( base, 0, 64 );
		if (base >= n) break; // wave-uniform
		const uint32_t i = base + tm;
		bool alive = false;
		uint32_t slot = 0, flags = 0, rng = 0, age = 0;
		f3 O = mk3( 0, 0, 0 ), D = mk3( 0, 0, 1 ), T = mk3( 0, 0, 0 );
		if (i < n && tm < per_wave) // the same for the four lanes of a team
		{
			const float4 ot = in.o_t[i], dh = in.d_h[i], ts = in.T_s[i];
			const uint32_t meta = in.meta[i];
			slot = meta & META_SLOT_MASK, flags = meta >> 30, rng = __float_as_uint( ts.w ), age = ( meta >> META_AGE_SHIFT ) & META_AGE_LATE;
			O = mk3( ot ), D = mk3( dh ), T = mk3( ts );
			alive = true;
			for (int nb = 0; nb < max_bounces; nb++)
			{
				RayHit h;
				unsigned long long tp = PROF ? wall_clock64() : 0;
				const bool walk = begin_closest( s, O, D, h );
				if (PROF) { const unsigned long long x = wall_clock64(); prof[0] += x - tp, tp = x, prof[4]++; }
				if (j == 0) nExt++;
				if (walk)
				{
					if (j == 0) nExtWalk++;
					const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
					const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
					// rays that the 4-wide nodes cannot take walk the pairs, the four lanes in step (same values to the same column)
					if (inexact) traverse<false, true, false>( s, O, D, rD, h, stack, stride, tc );
					else if (s.wide) team_traverse_wide<false, PROF>( s, j, O, D, rD, h, stack, stride, nSteps, nPrims, probe );
					else traverse<false, false, false>( s, O, D, rD, h, stack, stride, tc );
				}
				if (PROF) { const unsigned long long x = wall_clock64(); prof[1] += x - tp, tp = x; }
				Vertex v;
				const bool late = age == META_AGE_LATE;
				shade_vertex( s, O, D, h.t, h.obj, T, rng, flags, sinks, slot, late, v, j == 0 );
				age = age < META_AGE_LATE ? age + 1 : age;
				if (PROF) { const unsigned long long x = wall_clock64(); prof[2] += x - tp, tp = x; }
				if (v.shadow_any)
				{
					if (j == 0) nSh++;
					bool occluded, swalk;
					begin_any( s, v.shO, v.shD, v.shT, occluded, swalk );
					if (!occluded && !swalk) { if (j == 0) add_chain( sinks, slot, late, v.contrib ); }
					else if (swalk)
					{
						if (j == 0) nShWalk++;
						RayHit sh; sh.t = v.shT, sh.obj = -1, sh.pos = -1;
						const f3 rD = mk3( 1 / v.shD.x, 1 / v.shD.y, 1 / v.shD.z );
						const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
						const bool occ = inexact ? traverse<true, true, false>( s, v.shO, v.shD, rD, sh, stack, stride, tc )
							: s.wide ? team_traverse_wide<true>( s, j, v.shO, v.shD, rD, sh, stack, stride, nStepsAny, nPrimsAny )
							: traverse<true, false, false>( s, v.shO, v.shD, rD, sh, stack, stride, tc );
						if (!occ && j == 0) add_walked( sinks, slot, late, make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
					}
				}
				if (PROF) { const unsigned long long x = wall_clock64(); prof[3] += x - tp, tp = x; }
				if (!v.push_next) { alive = false; break; }
				O = v.newO, D = v.newD;
			}
		}
		// survivors of this round, compacted (one atomic per wave); a team's leader writes
		const bool keep = alive && j == 0;
		const unsigned long long m = __ballot( keep );
		if (m)
		{
			const int leader = __ffsll( (long long)m ) - 1;
			uint32_t b0 = 0;
			if ((int)lane == leader) b0 = atomicAdd( n_out, (uint32_t)__popcll( m ) );
			const uint32_t k = __shfl( b0, leader, 64 ) + __builtin_amdgcn_mbcnt_hi( (uint32_t)( m >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)m, 0 ) );
			if (keep)
			{
				out.o_t[k] = make_float4( O.x, O.y, O.z, 1e34f );
				out.d_h[k] = make_float4( D.x, D.y, D.z, __int_as_float( -1 ) );
				out.T_s[k] = make_float4( T.x, T.y, T.z, __uint_as_float( rng ) );
				out.meta[k] = slot | ( age << META_AGE_SHIFT ) | ( flags << 30 );
			}
		}
	} // next per_wave paths
	wave_add64( nExt, &ctr->extend_rays ), wave_add64( nSh, &ctr->shadow_rays );
	wave_add64( nExtWalk, &ctr->extend_walkers ), wave_add64( nShWalk, &ctr->shadow_walkers );
	if (PROF)
	{
		prof[5] = nSteps + nStepsAny, prof[6] = nPrims + nPrimsAny;
		wave_add64( j == 0 ? (unsigned long long)nSteps : 0ull, &ctr->tail_prof[11] );
		for (int k = 0; k < 7; k++) wave_add64( j == 0 ? prof[k] : 0ull, &ctr->tail_prof[k] );
		wave_add64( j == 0 ? probe[0] : 0ull, &ctr->tail_prof[9] ), wave_add64( j == 0 ? probe[1] : 0ull, &ctr->tail_prof[10] );
		if (lane == 0) atomicAdd( &ctr->tail_prof[7], wall_clock64() - prof_t0 ), atomicAdd( &ctr->tail_prof[8], __builtin_readcyclecounter() - prof_c0 );
	}
}

// ---- the very end of a render: one path per wave ---------------------------------------------------------------------------
// Below a few thousand live paths even a team leaves 60 lanes of its wave idle while its path's walk is a chain of ~30
// dependent steps.  Here the whole wave walks the one path breadth-first (rt_device.h: wave_traverse_wide), the analytic
// primitives are tested one per lane, and shading runs replicated as in k_tail_team; lane 0 writes.
__device__ __forceinline__ void wave_find_nearest( const DScene& s, int lane, f3 O, f3 D, float t0, RayHit& h, WaveLds& L, bool& walked, uint32_t& nIter )
{
	wave_analytic( s, lane, O, D, t0, h );
	walked = false;
	if (s.n_bvh_prims == 0) return;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (( inexact ? bvh_enter<true>( s, O, rD, h.t ) : bvh_enter<false>( s, O, rD, h.t ) ) == RT_DONE) return;
	walked = true;
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	// rays the 4-wide nodes cannot take, and rays whose frontier outgrows the LDS, walk as in k_tail: the 64 lanes in step,
	// one stack column (the same values to the same addresses)
	if (inexact) { traverse<false, true, false>( s, O, D, rD, h, L.nref, 1, tc ); return; }
	if (!s.wide) { traverse<false, false, false>( s, O, D, rD, h, L.nref, 1, tc ); return; }
	bool overflow;
	wave_traverse_wide<false>( s, lane, O, D, rD, h, L, overflow, nIter );
	if (overflow) traverse_wide<false>( s, O, D, rD, h, L.nref, 1 );
}
__device__ __forceinline__ bool wave_is_occluded( const DScene& s, int lane, f3 O, f3 D, float tmaxRay, WaveLds& L, bool& walked, uint32_t& nIter )
{
	walked = false;
	if (wave_analytic_any( s, lane, O, D, tmaxRay )) return true;
	if (s.n_bvh_prims == 0) return false;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (( inexact ? bvh_enter<true>( s, O, rD, tmaxRay ) : bvh_enter<false>( s, O, rD, tmaxRay ) ) == RT_DONE) return false;
	walked = true;
	RayHit sh; sh.t = tmaxRay, sh.obj = -1, sh.pos = -1;
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	if (inexact) return traverse<true, true, false>( s, O, D, rD, sh, L.nref, 1, tc );
	if (!s.wide) return traverse<true, false, false>( s, O, D, rD, sh, L.nref, 1, tc );
	bool overflow;
	const bool occ = wave_traverse_wide<true>( s, lane, O, D, rD, sh, L, overflow, nIter );
	return overflow ? traverse_wide<true>( s, O, D, rD, sh, L.nref, 1 ) : occ;
}

#define WAVE_WG 8 // waves per k_tail_wave workgroup: an empty round costs the dispatch of its workgroups, so few large ones
template <bool PROF>
__global__ __launch_bounds__( 64 * WAVE_WG ) void k_tail_wave( DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, int max_bounces, const uint32_t* draining, uint32_t wave_below, DevCounters* ctr )
{
	__shared__ WaveLds lds[WAVE_WG];
	const uint32_t n = *n_in;
	if (( ctr->tail_mode & 3u ) != 2) return; // another tail kernel's round (k_tail_decide)
	const int lane = (int)( threadIdx.x & 63u );
	WaveLds& L = lds[threadIdx.x >> 6];
	if (( blockIdx.x * (uint32_t)WAVE_WG + ( threadIdx.x >> 6 ) ) >= n) return; // surplus waves leave at once
	unsigned long long nExt = 0, nSh = 0, nExtWalk = 0, nShWalk = 0;
	unsigned long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const unsigned long long prof_t0 = PROF ? wall_clock64() : 0, prof_c0 = PROF ? __builtin_readcyclecounter() : 0;
	uint32_t nIter = 0, nIterAny = 0;
	for (;;)
	{
		uint32_t i = 0;
		if (lane == 0) i = atomicAdd( &ctr->pool_tail, 1u );
		i = (uint32_t)__builtin_amdgcn_readfirstlane( (int)i );
		if (i >= n) break;
		const float4 ot = in.o_t[i], dh = in.d_h[i], ts = in.T_s[i];
		const uint32_t meta = in.meta[i];
		uint32_t slot = meta & META_SLOT_MASK, flags = meta >> 30, rng = __float_as_uint( ts.w ), age = ( meta >> META_AGE_SHIFT ) & META_AGE_LATE;
		f3 O = mk3( ot ), D = mk3( dh ), T = mk3( ts );
		bool alive = true;
		for (int nb = 0; nb < max_bounces; nb++)
		{
			RayHit h;
			bool walked;
			unsigned long long tp = PROF ? wall_clock64() : 0;
			wave_find_nearest( s, lane, O, D, 1e34f, h, L, walked, nIter );
			if (PROF) { const unsigned long long x = wall_clock64(); prof[1] += x - tp, tp = x, prof[4]++; }
			nExt++, nExtWalk += walked ? 1 : 0;
			Vertex v;
			const bool late = age == META_AGE_LATE;
			shade_vertex( s, O, D, h.t, h.obj, T, rng, flags, sinks, slot, late, v, lane == 0 );
			age = age < META_AGE_LATE ? age + 1 : age;
			if (PROF) { const unsigned long long x = wall_clock64(); prof[2] += x - tp, tp = x; }
			if (v.shadow_any)
			{
				nSh++;
				const bool occ = wave_is_occluded( s, lane, v.shO, v.shD, v.shT, L, walked, nIterAny );
				nShWalk += walked ? 1 : 0;
				if (!occ && lane == 0)
				{
					if (walked) add_walked( sinks, slot, late, make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
					else add_chain( sinks, slot, late, v.contrib );
				}
			}
			if (PROF) { const unsigned long long x = wall_clock64(); prof[3] += x - tp, tp = x; }
			if (!v.push_next) { alive = false; break; }
			O = v.newO, D = v.newD;
		}
		if (alive && lane == 0) // survivor of this round
		{
			const uint32_t k = atomicAdd( n_out, 1u );
			out.o_t[k] = make_float4( O.x, O.y, O.z, 1e34f );
			out.d_h[k] = make_float4( D.x, D.y, D.z, __int_as_float( -1 ) );
			out.T_s[k] = make_float4( T.x, T.y, T.z, __uint_as_float( rng ) );
			out.meta[k] = slot | ( age << META_AGE_SHIFT ) | ( flags << 30 );
		}
	}
	if (lane == 0)
	{
		atomicAdd( &ctr->extend_rays, nExt ), atomicAdd( &ctr->shadow_rays, nSh ), atomicAdd( &ctr->extend_walkers, nExtWalk ), atomicAdd( &ctr->shadow_walkers, nShWalk );
		if (PROF)
		{
			prof[5] = nIter + nIterAny;
			for (int k = 0; k < 7; k++) if (prof[k]) atomicAdd( &ctr->tail_prof[k], prof[k] );
			atomicAdd( &ctr->tail_prof[7], wall_clock64() - prof_t0 ), atomicAdd( &ctr->tail_prof[8], __builtin_readcyclecounter() - prof_c0 );
			atomicAdd( &ctr->tail_prof[11], (unsigned long long)nIter );
		}
	}
}

// the parity hook through the same code (option hook_kernel = 2): wave w walks caller rays w, w + waves, ...
__global__ __launch_bounds__( 256 ) void k_rays_wave( DScene s, int n, const float* __restrict__ O3, const float* __restrict__ D3, const float* __restrict__ tmax,
	int any_hit, int* __restrict__ obj_out, float* __restrict__ t_out )
{
	__shared__ WaveLds lds[4];
	const int lane = (int)( threadIdx.x & 63u );
	WaveLds& L = lds[threadIdx.x >> 6];
	const int waves = (int)gridDim.x * 4;
	uint32_t nIter = 0;
	for (int i = (int)blockIdx.x * 4 + (int)( threadIdx.x >> 6 ); i < n; i += waves)
	{
		const f3 O = mk3( O3[3 * i], O3[3 * i + 1], O3[3 * i + 2] ), D = mk3( D3[3 * i], D3[3 * i + 1], D3[3 * i + 2] );
		bool walked;
		if (any_hit)
		{
			const bool occ = wave_is_occluded( s, lane, O, D, tmax[i], L, walked, nIter );
			if (lane == 0) obj_out[i] = occ ? 1 : 0, t_out[i] = tmax[i];
		}
		else
		{
			RayHit h;
			wave_find_nearest( s, lane, O, D, tmax[i], h, L, walked, nIter );
			if (lane == 0) obj_out[i] = h.obj, t_out[i] = h.t;
		}
	}
}

// ---- carry: the late paths (Sinks) that are alive when a batch leaves the wavefront loop join the NEXT batch's queue ------
// k_carry_out splits the batch's last state set: late paths go to the carry buffer (while it has room), the others stay
// with the batch (its k_tail finishes them before its k_accumulate).  k_carry_in appends the carry buffer to the state
// queue that k_primary has just filled, as entries N, N+1, ...: camera-ray paths and carried paths then bounce together.
// A carried path keeps its slot (it needs it for its pixel only) and its rng state: its arithmetic is what it would have
// been in a tail of its own.
// Veterans: a carried path gets the slot of its pixel in the LAST sample plane a batch can have (k_carry_in; a late path's
// slot is read modulo W*H only), so a late path whose slot is at or above `veteran_from` when the batch leaves the wavefront
// loop has already ridden through a batch: it has survived 30+ bounces and is most likely trapped between specular
// surfaces for hundreds more.  It stays with the batch, whose tail gives it some hundred bounces at once beside the next
// batches' bulk, instead of 17 per batch as a rider: the fewer such paths are alive when the caller finally waits, the
// shorter that wait.  (The batch's own paths of that sample plane are taken for veterans too: 1 in S of the newly late
// paths, a routing matter only.  veteran_from = 0xFFFFFFFF: none.)
__global__ __launch_bounds__( 512 ) void k_carry_out( PathState in, const uint32_t* __restrict__ n_in, PathState rest, uint32_t* n_rest, PathState carry, uint32_t* n_carry, uint32_t cap,
	uint32_t veteran_from )
{
	__shared__ uint32_t lds_a[16], lds_b[16];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = *n_in;
	float4 ot = make_float4( 0, 0, 0, 0 ), dh = ot, ts = ot;
	uint32_t meta = 0;
	const bool valid = i < n;
	if (valid) ot = in.o_t[i], dh = in.d_h[i], ts = in.T_s[i], meta = in.meta[i];
	const bool late = valid && ( ( meta >> META_AGE_SHIFT ) & META_AGE_LATE ) == META_AGE_LATE && !( meta & META_SUSPENDED ) && ( meta & META_SLOT_MASK ) < veteran_from;
	uint32_t idx[1];
	{
		const bool push[1] = { late };
		uint32_t* const counters[1] = { n_carry };
		block_append<1>( push, counters, idx, lds_a );
	}
	const bool carried = late && idx[0] < cap; // the counter may run past cap: readers clamp
	if (carried) carry.o_t[idx[0]] = ot, carry.d_h[idx[0]] = dh, carry.T_s[idx[0]] = ts, carry.meta[idx[0]] = meta;
	{
		const bool push[1] = { valid && !carried };
		uint32_t* const counters[1] = { n_rest };
		block_append<1>( push, counters, idx, lds_b );
	}
	if (valid && !carried) rest.o_t[idx[0]] = ot, rest.d_h[idx[0]] = dh, rest.T_s[idx[0]] = ts, rest.meta[idx[0]] = meta;
}
__global__ __launch_bounds__( 512 ) void k_carry_in( DScene s, PathState carry, const uint32_t* __restrict__ n_carry, uint32_t cap, PathState st, const uint32_t* __restrict__ n_st, uint32_t* __restrict__ wq, DevCounters* ctr,
	uint32_t WH, uint32_t veteran_from )
{
	__shared__ uint32_t lds[16];
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t nc = *n_carry < cap ? *n_carry : cap, N = *n_st; // N changes only in k_carry_commit
	bool walk = false;
	if (j < nc)
	{
		const float4 ot = carry.o_t[j], dh = carry.d_h[j];
		const f3 O = mk3( ot ), D = mk3( dh );
		RayHit h;
		walk = begin_closest( s, O, D, h ); // the entry's ray has not been traced yet
		st.o_t[N + j] = make_float4( O.x, O.y, O.z, h.t );
		st.d_h[N + j] = make_float4( D.x, D.y, D.z, __int_as_float( h.obj ) );
		st.T_s[N + j] = carry.T_s[j];
		uint32_t meta = carry.meta[j];
		if (veteran_from != 0xFFFFFFFFu) meta = ( meta & ~META_SLOT_MASK ) | ( ( meta & META_SLOT_MASK ) % WH + veteran_from ); // (every carried path is late)
		st.meta[N + j] = meta;
	}
	const bool push[1] = { walk };
	uint32_t* const counters[1] = { &ctr->n_walk[0] };
	uint32_t idx[1];
	block_append<1>( push, counters, idx, lds );
	if (walk) wq[idx[0]] = N + j;
}
__global__ void k_carry_commit( uint32_t* n_st, uint32_t* n_carry, uint32_t cap )
{
	*n_st += *n_carry < cap ? *n_carry : cap;
	*n_carry = 0;
}
// accumulator[p] += the exact sum of the late contributions of pixel p, which starts again at zero
__global__ __launch_bounds__( BLOCK ) void k_fold_late( float4* __restrict__ acc, long long* __restrict__ late, float4* __restrict__ odd, int WH )
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= WH) return;
	const longlong2 s01 = *(const longlong2*)( late + 4 * (size_t)p );
	const long long s2 = late[4 * (size_t)p + 2];
	const float4 o = odd[p]; // contributions that did not fit the fixed-point sums (add_odd); almost always +0
	const bool anyOdd = ( __float_as_uint( o.x ) | __float_as_uint( o.y ) | __float_as_uint( o.z ) ) != 0;
	if (( s01.x | s01.y | s2 ) == 0 && !anyOdd) return;
	const float inv = 1.0f / ES_SCALE;
	float4 a = acc[p];
	a.x += (float)s01.x * inv, a.y += (float)s01.y * inv, a.z += (float)s2 * inv;
	if (anyOdd) a.x += o.x, a.y += o.y, a.z += o.z, odd[p] = make_float4( 0, 0, 0, 0 );
	acc[p] = a;
	late[4 * (size_t)p] = 0, late[4 * (size_t)p + 1] = 0, late[4 * (size_t)p + 2] = 0;
}

// end of a bounce: the queue set that was just consumed becomes the output set of the next bounce
// `report` (may be null): a word of pinned host memory that receives the number of paths + carried shadow rays the next
// bounce starts with -- written from here instead of a 4-byte copy command per bounce (a copy kernel and its launch gap)
__global__ void k_next_bounce( DevCounters* ctr, int consumed, uint32_t* report )
{
	ctr->n_state[consumed] = 0, ctr->n_walk[consumed] = 0, ctr->n_shadow[consumed] = 0, ctr->pool_extend = 0, ctr->pool_connect = 0, ctr->pool_tail = 0;
	ctr->n_rec[consumed ^ 1] = 0, ctr->n_srec[consumed ^ 1] = 0; // read by this bounce's walks, written again by the next bounce's
	const uint32_t pending = ctr->n_state[consumed ^ 1] + ctr->n_shadow[consumed ^ 1];
	ctr->pending = pending;
	if (report) { *report = pending; __threadfence_system(); }
}

// the same bookkeeping in two parts, for a bounce whose shadow walk runs on the side stream beside the next bounce's
// closest-hit walk (render_batch): what the next closest-hit walk needs, after k_shade; the rest, after the shadow walk
__global__ void k_next_bounce_pre( DevCounters* ctr, int consumed )
{
	ctr->pool_extend = 0, ctr->n_rec[consumed ^ 1] = 0;
}
__global__ void k_next_bounce_post( DevCounters* ctr, int consumed, uint32_t* report )
{
	ctr->n_state[consumed] = 0, ctr->n_walk[consumed] = 0, ctr->n_shadow[consumed] = 0, ctr->pool_connect = 0, ctr->pool_tail = 0;
	ctr->n_srec[consumed ^ 1] = 0;
	const uint32_t pending = ctr->n_state[consumed ^ 1] + ctr->n_shadow[consumed ^ 1];
	ctr->pending = pending;
	if (report) { *report = pending; __threadfence_system(); }
}

// ---- accumulate: accumulator[p] += Sample() per sample, in sample order (renderer.cpp:566-567) -------------
__global__ __launch_bounds__( BLOCK ) void k_accumulate( float4* __restrict__ acc, const float4* __restrict__ E, const long long* __restrict__ Es, int WH, int S )
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= WH) return;
	float4 a = acc[p];
	for (int k = 0; k < S; k++)
	{
		const size_t slot = (size_t)k * WH + p;
		const float4 e = E[slot];
		const longlong2 s01 = *(const longlong2*)( Es + 4 * slot );
		const long long s2 = Es[4 * slot + 2];
		const float inv = 1.0f / ES_SCALE;
		// Sample() of this path = float chain + exactly summed walked-shadow contributions, then accumulator += Sample()
		a.x += e.x + (float)s01.x * inv, a.y += e.y + (float)s01.y * inv, a.z += e.z + (float)s2 * inv, a.w += 0.0f;
	}
	acc[p] = a;
}

// ---- finalize: RGBF32_to_RGB8 (precomp.h:431-465, finalize.cl) ------------------------------------------------
__global__ __launch_bounds__( BLOCK ) void k_finalize( const float4* __restrict__ acc, uint32_t* __restrict__ out, int n, int frames )
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= n) return;
	float4 v = acc[p];
	if (frames > 0) { const float f = (float)frames; v.x = v.x / f, v.y = v.y / f, v.z = v.z / f; }
	// std::min(1.0f, x) = (x < 1.0f) ? x : 1.0f ; NaN -> 1.0f.  Whitted glass can leave slightly negative channels
	// (Fresnel terms above 1): the reference's (uint) cast of a negative float is the x86-64 one, a truncating
	// 64-bit conversion whose low 32 bits are kept, so -3.2 becomes 0xFFFFFFFD; restated here, not saturated to 0.
	const uint32_t r = (uint32_t)(long long)( 255.0f * ( v.x < 1.0f ? v.x : 1.0f ) );
	const uint32_t g = (uint32_t)(long long)( 255.0f * ( v.y < 1.0f ? v.y : 1.0f ) );
	const uint32_t b = (uint32_t)(long long)( 255.0f * ( v.z < 1.0f ? v.z : 1.0f ) );
	out[p] = ( r << 16 ) + ( g << 8 ) + b;
}

// ---- post-processing (postprocessing.h:16-99) ----------------------------------------------------------------
// gamma: per channel c -> (uint)(255 * pow(c/255, 1/gamma)); a function of the 8-bit input only, so the 256-entry
// table is computed on the host with the same powf the CPU path uses and applied here.
__global__ __launch_bounds__( BLOCK ) void k_gamma( uint32_t* __restrict__ px, int n, const uint32_t* __restrict__ lut )
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= n) return;
	const uint32_t c = px[p];
	px[p] = ( lut[( c >> 16 ) & 255] << 16 ) + ( lut[( c >> 8 ) & 255] << 8 ) + lut[c & 255];
}
__global__ __launch_bounds__( BLOCK ) void k_chroma( const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int W, int H, float intensity )
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= W * H) return;
	const int x = p % W, y = p / W;
	const float cx = (float)W / 2.0f, cy = (float)H / 2.0f;
	const float max_distance = sqrtf( cx * cx + cy * cy );
	const float g_off = intensity, b_off = intensity * 2;
	const float fx = (float)x, fy = (float)y;
	const float ddx = cx - fx, ddy = cy - fy;
	const float distance = sqrtf( ddx * ddx + ddy * ddy );
	const float ratio = distance / max_distance;
	const uint32_t r = ( src[p] & 0xFF0000 ) >> 16;
	// dir = normalize(cord - center); at the centre pixel this is 0 * inf = NaN in the reference and the int cast is
	// undefined: the offset is defined as 0 there.
	const float vx = fx - cx, vy = fy - cy;
	const float inv = 1.0f / sqrtf( vx * vx + vy * vy );
	float dirx = vx * inv, diry = vy * inv;
	if (!( dirx == dirx ) || !( diry == diry )) dirx = 0.0f, diry = 0.0f;
	float c2x = fx + (float)(int)( dirx * g_off * ratio ), c2y = fy + (float)(int)( diry * g_off * ratio );
	int nx = max( 0, min( (int)c2x, W - 1 ) ), ny = max( 0, min( (int)c2y, H - 1 ) );
	const uint32_t g = ( src[nx + ny * W] & 0x00FF00 ) >> 8;
	c2x = fx + (float)(int)( dirx * b_off * ratio ), c2y = fy + (float)(int)( diry * b_off * ratio );
	nx = max( 0, min( (int)c2x, W - 1 ) ), ny = max( 0, min( (int)c2y, H - 1 ) );
	const uint32_t b = src[nx + ny * W] & 0x0000FF;
	dst[p] = ( r << 16 ) + ( g << 8 ) + b;
}
__global__ __launch_bounds__( BLOCK ) void k_vignette( uint32_t* __restrict__ px, int W, int H, float intensity )
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= W * H) return;
	const int x = p % W, y = p / W;
	const float cx = (float)W / 2.0f, cy = (float)H / 2.0f;
	const float max_distance = sqrtf( cx * cx + cy * cy );
	const float ddx = cx - (float)x, ddy = cy - (float)y;
	const float distance = sqrtf( ddx * ddx + ddy * ddy );
	const float q = 1 - ( distance * intensity / max_distance );
	const float dimming = ( q < 0.0f ) ? 0.0f : q; // std::max(q, 0.0f)
	const uint32_t c = px[p];
	const uint32_t r = (uint32_t)( (float)( ( c & 0xFF0000 ) >> 16 ) * dimming );
	const uint32_t g = (uint32_t)( (float)( ( c & 0x00FF00 ) >> 8 ) * dimming );
	const uint32_t b = (uint32_t)( (float)( c & 0x0000FF ) * dimming );
	px[p] = ( r << 16 ) + ( g << 8 ) + b;
}

// ---- parity hook: caller rays -----------------------------------------------------------------------------------
__global__ __launch_bounds__( BLOCK ) void k_trace_rays( DScene s, int n, const float* __restrict__ O3, const float* __restrict__ D3, const float* __restrict__ tmax,
	int any_hit, int* __restrict__ obj_out, float* __restrict__ t_out )
{
	extern __shared__ uint32_t lds_stack[];
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	const bool live = i < n;
	const int k = live ? i : 0;
	const f3 O = mk3( O3[3 * k], O3[3 * k + 1], O3[3 * k + 2] ), D = mk3( D3[3 * k], D3[3 * k + 1], D3[3 * k + 2] );
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	if (live)
	{
		if (any_hit)
		{
			const bool occ = is_occluded<false>( s, O, D, tmax[k], lds_stack + threadIdx.x, BLOCK, tc );
			obj_out[k] = occ ? 1 : 0, t_out[k] = tmax[k];
		}
		else
		{
			RayHit h; h.t = tmax[k], h.obj = -1, h.pos = -1;
			find_nearest<false>( s, O, D, h, lds_stack + threadIdx.x, BLOCK, tc );
			obj_out[k] = h.obj, t_out[k] = h.t;
		}
	}
}
__global__ __launch_bounds__( BLOCK ) void k_copy_hits( PathState st, int n, int* __restrict__ obj_out, float* __restrict__ t_out )
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	obj_out[i] = __float_as_int( st.d_h[i].w ), t_out[i] = st.o_t[i].w;
}
// The same hook through the render path's own machinery (option hook_kernel = 1): caller rays become entries of a path queue
// (closest hit) or of a shadow queue (any hit) exactly as k_primary / k_shade would have queued them, and k_trace walks them.
// An unoccluded shadow ray adds its "contribution" (1, 0, 0) to Es[ray]; k_rays_any_result turns that back into a flag.
__global__ __launch_bounds__( SBLOCK ) void k_rays_begin( DScene s, int n, const float* __restrict__ O3, const float* __restrict__ D3, const float* __restrict__ tmax, int any_hit,
	PathState st, ShadowQueue sq, float4* __restrict__ Es, uint32_t* __restrict__ wq, int* __restrict__ obj_out, float* __restrict__ t_out, DevCounters* ctr )
{
	__shared__ uint32_t lds[2 * 16];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	bool walk = false, swalk = false;
	f3 O = mk3( 0, 0, 0 ), D = mk3( 0, 0, 1 );
	float tm = 0;
	if (i < n)
	{
		O = mk3( O3[3 * i], O3[3 * i + 1], O3[3 * i + 2] ), D = mk3( D3[3 * i], D3[3 * i + 1], D3[3 * i + 2] ), tm = tmax[i];
		if (any_hit)
		{
			bool occluded;
			begin_any( s, O, D, tm, occluded, swalk );
			obj_out[i] = occluded ? 1 : ( swalk ? 2 : 0 ), t_out[i] = tm;
			Es[2 * (size_t)i] = make_float4( 0, 0, 0, 0 ), Es[2 * (size_t)i + 1] = make_float4( 0, 0, 0, 0 );
		}
		else
		{
			RayHit h;
			walk = begin_closest( s, O, D, h, tm );
			st.o_t[i] = make_float4( O.x, O.y, O.z, h.t );
			st.d_h[i] = make_float4( D.x, D.y, D.z, __int_as_float( h.obj ) );
			st.T_s[i] = make_float4( 0, 0, 0, 0 );
			st.meta[i] = (uint32_t)i;
		}
	}
	const bool push[2] = { walk, swalk };
	uint32_t* const counters[2] = { &ctr->n_walk[0], &ctr->n_shadow[0] };
	uint32_t idx[2];
	block_append<2>( push, counters, idx, lds );
	if (walk) wq[idx[0]] = (uint32_t)i;
	if (swalk)
	{
		sq.o_t[idx[1]] = make_float4( O.x, O.y, O.z, tm );
		sq.d_s[idx[1]] = make_float4( D.x, D.y, D.z, __uint_as_float( (uint32_t)i ) );
		sq.c[idx[1]] = make_float4( 1, 0, 0, 0 );
	}
	if (i == 0) ctr->n_state[0] = (uint32_t)n;
}
__global__ __launch_bounds__( BLOCK ) void k_rays_any_result( const long long* __restrict__ Es, int n, int* __restrict__ obj_out )
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n && obj_out[i] == 2) obj_out[i] = Es[4 * (size_t)i] != 0 ? 0 : 1;
}

// ---- Whitted / debug views: Renderer::Trace without recursion (renderer.cpp:80-244) ------------------------
// Each pending call is a frame; a GLASS frame is visited up to three times (before the reflection child, before the
// refraction child, after both).  Frames live in a per-thread slice of a global buffer (depth <= max_depth).
struct WFrame
{
	f3 O, D;          // the ray of this call
	float t;          // ray.t after FindNearest (needed by the glass absorption and Fresnel terms)
	uint32_t inside;  // ray.inside
	int stage;        // 0 = not started, 1 = waiting for reflection, 2 = waiting for refraction, 3 = mirror child
	f3 aa;            // albedo * absorption (glass) or material.color (mirror)
	f3 refl;          // reflectionColor
	f3 N;
	float n1, n2, ior, ratio, inc, k;
};

__device__ __forceinline__ f3 direct_illumination( const DScene& s, f3 I, f3 N, uint32_t* stack, int stride ) // scene.h:490-515
{
	f3 acc = mk3( 0, 0, 0 );
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	for (int i = 0; i < 2; i++)
	{
		const f3 dir = mk3( s.wl_pos[i] ) - I;
		const f3 dirN = normalize3( dir );
		const float eps = 0.001f;
		const f3 so = I + dirN * eps;
		const float mag = magnitude3( dir );
		const float st = mag - eps * 2;
		if (!is_occluded<false>( s, so, dirN, st, stack, stride, tc ))
		{
			const float distanceEnergy = 1 / sqrf_( st );
			const float d = dot3( N, dirN );
			const float angularEnergy = ( d < 0.0f ) ? 0.0f : d; // std::max(d, 0.0f)
			acc = acc + mk3( s.wl_color[i] ) * s.wl_intensity * distanceEnergy * angularEnergy;
		}
	}
	return acc;
}

__global__ __launch_bounds__( BLOCK ) void k_whitted( DScene s, CamProp cam, int W, int H, float invW, float invH, int mode, int max_depth, int aa,
	WFrame* __restrict__ frames, float4* __restrict__ acc )
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t* stack = lds_stack + threadIdx.x;
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p >= W * H) return;
	const int x = p % W, y = p / W;
	WFrame* fr = frames + (size_t)p * ( max_depth + 1 );
	const float pat[8] = { -1.0f / 8.0f, 3.0f / 8.0f, 3.0f / 8.0f, 1.0f / 8.0f, -3.0f / 8.0f, -1.0f / 8.0f, 1.0f / 8.0f, -3.0f / 8.0f }; // renderer.h:31-36
	float4 total = make_float4( 0, 0, 0, 0 );
	const int nsamp = aa ? 4 : 1;
	TravCount tc; tc.pairs = 0, tc.prims = 0;
	for (int smp = 0; smp < nsamp; smp++)
	{
		f3 O, D;
		const float sx = aa ? (float)x + pat[2 * smp] : (float)x, sy = aa ? (float)y + pat[2 * smp + 1] : (float)y;
		primary_ray( cam, sx, sy, invW, invH, O, D );
		int depth = 0;
		fr[0].O = O, fr[0].D = D, fr[0].inside = 0, fr[0].stage = 0;
		f3 ret = mk3( 0, 0, 0 ); // value returned by the most recently finished call
		while (depth >= 0)
		{
			WFrame& f = fr[depth];
			if (f.stage == 0)
			{
				if (depth >= max_depth) { ret = mk3( 0, 0, 0 ), depth--; continue; } // renderer.cpp:135-136
				RayHit h; h.t = 1e34f, h.obj = -1, h.pos = -1;
				find_nearest<false>( s, f.O, f.D, h, stack, BLOCK, tc );
				if (h.obj == -1) { ret = mk3( 0, 0, 0 ), depth--; continue; }
				const f3 I = f.O + h.t * f.D;
				HitAttr a;
				hit_attributes( s, h.obj, I, f.D, a );
				const int type = a.mat->type;
				if (mode == RTMI_ALBEDO) { ret = a.albedo, depth--; continue; }
				if (mode == RTMI_NORMAL) { ret = ( a.N + mk3( 1, 1, 1 ) ) * 0.5f, depth--; continue; }
				if (mode == RTMI_DISTANCE) { ret = 0.1f * mk3( h.t, h.t, h.t ), depth--; continue; }
				if (type == RTMI_DIFFUSE || type == RTMI_LIGHT)
				{
					ret = a.albedo / 3.14159265358979323846264f * direct_illumination( s, I, a.N, stack, BLOCK );
					depth--;
				}
				else if (type == RTMI_MIRROR)
				{
					const f3 refl = f.D - ( 2 * dot3( f.D, a.N ) ) * a.N;
					f.aa = mk3( a.mat->color ), f.stage = 3;
					WFrame& c = fr[depth + 1];
					c.O = I + refl * 0.001f, c.D = refl, c.inside = 0, c.stage = 0;
					depth++;
				}
				else if (type == RTMI_GLASS) // ComputePixelColorGlass, renderer.cpp:80-128
				{
					const f3 refl = f.D - 2.0f * ( dot3( f.D, a.N ) ) * a.N;
					const float air = 1.0003f, ior = a.mat->ior;
					f3 absorption = mk3( 1, 1, 1 );
					if (f.inside)
					{
						const float4 ab = a.mat->absorption;
						absorption = mk3( expf( -ab.x * h.t ), expf( -ab.y * h.t ), expf( -ab.z * h.t ) );
						f.n1 = ior, f.n2 = air;
					}
					else f.n1 = air, f.n2 = ior;
					f.ior = ior, f.ratio = f.n1 / f.n2, f.aa = a.albedo * absorption, f.N = a.N, f.t = h.t;
					f.inc = dot3( a.N, -f.D );
					f.k = 1.0f - sqrf_( f.ratio ) * ( 1.0f - sqrf_( f.inc ) );
					f.stage = 1;
					WFrame& c = fr[depth + 1];
					c.O = I + refl * 0.001f, c.D = refl, c.inside = f.inside ? 1u : 0u, c.stage = 0;
					// keep the hit point for the refraction child
					f.refl = I; // temporarily holds I until the reflection colour replaces it
					depth++;
				}
				else { ret = mk3( 0, 0, 0 ), depth--; } // SUBSTANCE: the reference falls out of the switch (UB); defined as black
			}
			else if (f.stage == 3) { ret = f.aa * ret, depth--; } // material.color * Trace(...)
			else if (f.stage == 1)
			{
				const f3 I = f.refl;
				const f3 reflectionColor = f.aa * ret;
				if (f.k < 0) { ret = reflectionColor, depth--; continue; }
				const f3 refr = f.ratio * f.D + f.N * ( f.ratio * f.inc - sqrtf( f.k ) );
				f.refl = reflectionColor, f.stage = 2;
				// outcoming angle needs refr again after the child returns: recompute then (same expression, same bits)
				WFrame& c = fr[depth + 1];
				c.O = I + refr * 0.001f, c.D = refr, c.inside = f.inside ? 0u : 1u, c.stage = 0;
				depth++;
			}
			else // stage 2: both children done
			{
				const f3 refractionColor = f.aa * ret;
				const f3 refr = f.ratio * f.D + f.N * ( f.ratio * f.inc - sqrtf( f.k ) );
				const float outc = dot3( -f.N, refr );
				const double left = sqrf_( ( f.n1 * f.inc - f.ior * outc ) / ( f.n1 * f.inc + f.n2 * outc ) );
				const double right = sqrf_( ( f.n1 * outc - f.ior * f.inc ) / ( f.n1 * outc + f.n2 * f.inc ) );
				const float Fr = (float)( 0.5 * ( left + right ) );
				ret = Fr * f.refl + ( 1 - Fr ) * refractionColor;
				depth--;
			}
		}
		total.x += ret.x, total.y += ret.y, total.z += ret.z;
	}
	if (aa) total.x /= 4.0f, total.y /= 4.0f, total.z /= 4.0f, total.w /= 4.0f;
	acc[p] = total;
}

// =================================================================================================================
// host side
// =================================================================================================================
// One render batch in flight.  The bulk bounces of a batch run on the handle's stream; one k_tail round for the few paths
// that are not late at the hand-over (with RTMI_CARRY=0: the whole tail) and its k_accumulate run on the slot's own stream,
// so that they overlap the bulk bounces of the following batches.  k_accumulate launches are chained by events in batch
// order (the accumulator sums samples in sample order, renderer.cpp:566-567).
#define MAX_SLOTS 16
struct BatchSlot
{
	int batch = 0;              // capacity in samples
	PathState st[2] = {};       // queue-ordered path state, ping-pong
	ShadowQueue sq[2] = {};     // ping-pong: the other one collects the shadow rays parked by k_trace<any>
	uint32_t* wq[2] = { 0, 0 }; // per state set: entries that walk the BVH
	uint32_t* rec[2] = { 0, 0 }, *srec[2] = { 0, 0 }; // suspend records of k_trace<closest> / <any>, ping-pong
	float4* E = 0;              // radiance estimate per slot index (sample_in_batch * W*H + pixel)
	long long* Es = 0;          // 4 x int64 per slot index: exact sum of walked shadow contributions
	uint32_t rec_cap = 0;
	int rec_stride = 0;
	DevCounters* ctr = 0;
	uint32_t* host_counts = 0;  // pinned, device-mapped ring of queue lengths written by k_next_bounce
	uint32_t* host_counts_dev = 0; // the same ring as the device sees it
	hipEvent_t count_ev[4] = { 0, 0, 0, 0 };
	hipStream_t tail_stream = 0;
	hipEvent_t ev_shade = 0, ev_any = 0; // side stream hand-shake (render_batch)
	PathState left = {};        // late paths that outlived the first round of this batch's k_tail: they join the next batch that uses the slot
	uint32_t* n_left = 0;
	hipEvent_t bulk_done = 0, acc_done = 0;
	bool busy = false;          // acc_done has been recorded and not yet waited for by a new user of the slot
};

// The accumulator sums the batches in sample order (renderer.cpp:566-567), whichever lane (below) renders them: every batch
// draws a ticket when it is dispatched and takes its turn at the point where it enqueues its k_accumulate -- wait for the
// previous batch's event, launch, record its own.  Two host threads at most; a batch that fails still passes its turn on.
struct AccChain
{
	std::mutex m;
	std::condition_variable cv;
	uint64_t next_ticket = 0, turn = 0;
	hipEvent_t last_acc = 0; // acc_done of the most recent batch: the next accumulate and every reader wait for it
};
struct Turn
{
	AccChain* ch; uint64_t ticket; bool entered = false, left = false;
	std::unique_lock<std::mutex> lk;
	Turn( AccChain* ch, uint64_t ticket ) : ch( ch ), ticket( ticket ), lk( ch->m, std::defer_lock ) {}
	void enter() { if (entered) return; lk.lock(); ch->cv.wait( lk, [&] { return ch->turn == ticket; } ); entered = true; }
	void leave() { if (left) return; enter(); ch->turn++; left = true; lk.unlock(); ch->cv.notify_all(); }
	~Turn() { leave(); }
};

struct rtmi_ctx
{
	int device = 0, W = 0, H = 0;
	// Second lane (option lanes = 2): a whole second pipeline -- streams, batch slots, carry buffer -- in a context of its own that
	// shares this one's scene, seeds, accumulator and late sums.  rtmi_render gives every other batch to it, enqueued by a second
	// host thread, so that two full batches are in flight: one's k_shade and small late bounces run beside the other's walks.
	rtmi_ctx* child = 0;
	bool is_child = false;
	int lanes = 1;
	uint32_t* child_seeds = 0; long long* child_late = 0; float4* child_odd = 0; // the child's own allocations (it uses the parent's while it lives)
	hipEvent_t ev_fork = 0, ev_join = 0;
	uint64_t batch_seq = 0;      // batches dispatched: even ones run here, odd ones on the child
	AccChain own_chain; AccChain* chain = &own_chain;
	hipStream_t own_stream = 0, stream = 0;
	std::string err;
	bool has_scene = false;
	DScene ds = {};
	std::vector<void*> scene_allocs;
	int stack_depth = 8;         // per-lane traversal stack over the pairs (k_whitted, k_trace_rays)
	int walk_depth = 8;          // same for k_trace and k_tail: the larger need of the pairs and of the 4-wide nodes; sizes the suspend records
	hipStream_t side = 0;        // a bounce's shadow walk runs here, beside the next bounce's closest-hit walk on the main stream (RTMI_OVERLAP=0: on the main stream)
	bool overlap_on = true;
	bool plane_pairs_on = true;  // option plane_pairs = 0: the generic plane loop even for the reference's room (isect_planes); takes effect at the next upload
	bool wide_on = true;         // RTMI_WIDE=0: k_trace walks the pairs instead of the 4-wide nodes with quantised boxes (rt_device.h: wide_step)
	int refill = 16;        // refill a wave of k_trace when at least this many lanes are idle
	int max_blocks = 1024;  // resident 256-thread blocks for the persistent kernels (set from LDS use at upload)
	uint32_t grid_closest[2] = { 0, 0 }, grid_any[2] = { 0, 0 }; // wgrid; RTMI_GRID_CLOSEST / RTMI_GRID_ANY "a,b" (paths) override, "0,0": always 4 blocks per CU
	int n_cus = 256;
	CamProp cam = {};
	uint32_t* seeds = 0;
	float4* accum_own = 0;
	float4* accum = 0;
	uint32_t* pixels = 0, *pixels2 = 0, *gamma_lut = 0;
	// path tracing: up to MAX_SLOTS render batches in flight (see BatchSlot)
	BatchSlot slots[MAX_SLOTS];
	int n_slots = 4, next_slot = 0;
	uint32_t* draining = 0, *draining_dev = 0; // pinned, device-mapped flag read by k_tail (see there)
	// late paths (Sinks): exact per-pixel sums, and the buffer in which the late survivors of a batch wait for the next one
	long long* late = 0;         // 4 x int64 per pixel
	float4* odd = 0;             // per pixel: contributions outside the fixed-point range (add_odd), folded with `late`
	PathState carry[2] = {};     // [0] collects (k_carry_out) and feeds (k_carry_in); [1] is the ping-pong partner of the final tail
	DevCounters* carry_ctr = 0;  // n_state[0] = paths in carry[0]
	uint32_t carry_cap = 1u << 23, left_cap = 1u << 16;
	bool carry_on = true;        // RTMI_CARRY=0: every batch finishes all of its paths in its own tail
	bool paths_carried = false;  // some batch since the last join handed paths on (carry buffer, slot lists): the next batch, or join_batches, takes them in
	uint32_t carry_min = 1u << 22; // option carry_min: a batch of no more camera paths than this finishes its own paths (see render_batch)
	bool carry_pending = false;  // paths may be waiting in carry[0] / late sums may be non-zero: join_batches finishes them
	int count_lag = 1;           // option count_lag: bounces after which a small batch's path counts are read (see render_batch)
	// option small_paths: a batch with no more paths than this leaves the wavefront loop at once.  Renderer::Tick at 1280x720
	// (0.92 M paths per frame): 2^17 5.92 ms, 2^18 5.32, 2^19 4.91, 2^20 4.72 -- below a million paths a wavefront bounce is
	// launch latency (0.28 ms each) and the tail kernel's 32 bounces in one launch are cheaper; the 20-step bench does not care
	uint32_t small_paths = 1u << 20;
	int carry_bounce = 16;       // with carry: the wavefront loop runs at least this many bounces (paths are late from vertex 7; the bounces up to the 16th are large enough to be worth running here)
	uint32_t tail_paths = 1u << 18; // finish a batch with k_tail once no more than this many paths are alive (0 = never)
	int tail_lanes = 16;         // fewest paths per wave in the tail rounds after the first (power of two <= 64)
	int sblock = 512;            // threads per block of k_primary / k_shade (RTMI_SBLOCK): blocks of 16 waves wait for a CU with 4 free
	                             // slots on every SIMD, which the long-lived tail waves of earlier batches rarely leave
	int tail_waves = 16;         // resident k_tail waves per CU
	int tail_block = 4;          // waves per k_tail workgroup (RTMI_TAIL_BLOCK)
	uint32_t tail_dense = 4096, tail_mid = 0; // k_tail: 64 paths per wave above tail_dense live paths, 16 above tail_mid, else tail_lanes
	bool log_bounces = false;   // option log_bounces: print the path count per bounce (diagnostics)
	int budget = 32;            // tail steps after which a still-walking ray is parked (0 = never)
	int batch_opt = 0;          // option batch: samples per render batch (0 = as many as 2^RTMI_SLOT_BITS paths allow)
	int fit_samples = 0;        // > 0: the largest batch whose queues were found to fit in device memory (rtmi_render)
	int blocks_per_cu_opt = 0;  // option blocks_per_cu: resident walk blocks per CU (0 = min(4, what the LDS stack allows))
	bool tail_paths_set = false;
	bool check_launches = false; // option check_launches: synchronise and check after every bounce (localises a fault; slow)
	int hook_kernel = 0;        // option hook_kernel: rtmi_trace_rays runs 0 = k_trace_rays (one ray per lane over the reference's pairs),
	                            // 1 = the render path's own queues and persistent walk (k_trace: 4-wide nodes where built),
	                            // 2 = the end-of-render walker, one ray per wave (k_rays_wave = k_tail_wave's device code)
	// option veteran_rounds: -1 (default) = carried paths are carried again; >= 0: they stay with the batch they rode, whose tail
	// runs this many extra rounds.  Measured (20 bench steps): the final wait falls from 357 ms to 208 ms (2 rounds) / 190 ms (3),
	// but the tails' waves beside the bulk cost more than that: 126.4 -> 140.5 ms per step.  A trapped path-bounce costs about
	// 1 ns as a rider of the wavefront kernels and 2-3 ns in a tail kernel, wherever that tail runs.
	int last_round_bounces = 1 << 16; // see launch_tail_rounds
	int veteran_rounds = -1;
	// option preview_reads: rtmi_finalize shows the image as far as it has got instead of finishing every path first -- the frame
	// loop's display path (Renderer::Tick = one sample and a finalize per frame).  The paths of a frame that are still bouncing
	// when it is shown (a few thousand of 2 M on the 1 M-triangle scene, inside the mirror and glass knots for thousands of
	// bounces) ride the next frames' batches; every other read, rtmi_flush and a camera change join as before.
	bool preview_reads = false;
	bool stop_late = true;      // option stop_late: a batch's stragglers leave its tail round as soon as they are late (k_tail) instead of after 32 bounces
	bool tail_team = true;      // option tail_team: draining rounds with no more than team_below live paths run k_tail_team (four lanes per path)
	bool tail_profile = false;  // option tail_profile: the tail kernels time their phases (rtmi_stats.tail_profile)
	uint32_t team_below = 1u << 16; // option team_below.  Measured at the end of an 8 x 64-spp render (108 k ... 6 k paths alive over the rounds),
	                            // before k_tail_wave took more than 4 k paths: k_tail alone 372 ms, teams below 16 k 349, below 64 k 296, below 128 k 281, below 256 k 282
	bool tail_wave = true;      // option tail_wave: ... and k_tail_wave (one path per wave) at or below wave_below live paths
	// option wave_below.  The late rounds (1 024 bounces and more) last as long as their longest path: bounces x latency of a bounce,
	// 9 us in k_tail_wave against 31 us in a loaded k_tail_team -- worth it even when the 4 096 resident waves take four paths each
	// one after the other.  End of a 20-step render (wave_below / team_below): 4 k / 128 k 425 ms (two lanes) 356 (one);
	// 8 k / 64 k 362, 367; 16 k / 64 k 363, 328; 16 k / 16 k 372; 32 k / 128 k 411; 64 k / 128 k 461
	uint32_t wave_below = 1u << 14;
	// device scratch of the parity hooks and rtmi_postprocess, grown on demand and kept (no allocation per call)
	void* hook_buf = 0; size_t hook_cap = 0;
	uint32_t gamma_host[256]; float gamma_cached = 0; // host image of gamma_lut (lives as long as the copy needs it)
	void* noalloc_scratch = 0;  // this device's address of the file-scope scratch array (rtmi_trace_rays_noalloc)
	bool drain_open = false; double drain_t0 = 0;
	WFrame* frames = 0;
	int frames_depth = 0;
	// stats / profiling
	bool prof_events = false, count_nodes = false;
	rtmi_stats stats = {};
	struct TimedEv { hipEvent_t a, b; int cls; };
	std::vector<TimedEv> timed;
	std::vector<hipEvent_t> ev_pool;
	hipEvent_t prof_base = 0;    // recorded when event timing is switched on: launch intervals are taken relative to it
	std::vector<std::pair<float, float>> walk_intervals; // the same for the walk windows
	std::vector<std::pair<float, float>> ext_intervals; // [start, end) of the closest-hit walk launches since rtmi_reset_stats, ms after prof_base
};

// ---- rtmi_trace_rays_noalloc: staging in a file-scope device array instead of allocated memory --------------------------
// 36 bytes per ray (origin, direction, tmax in; object id, distance out).  Every GPU the library runs on holds its own copy
// of the array, at its own address, until the process ends.  In host code the NAME is only a handle for
// hipGetSymbolAddress: the address kernels and copies use is looked up once per device (scratch_address, under the
// device being current) and kept.
#define RTMI_NOALLOC_MAX_RAYS ( 1 << 18 )
static __device__ __align__( 256 ) unsigned char scratch[(size_t)RTMI_NOALLOC_MAX_RAYS * 36];
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[64] = {};
static hipError_t scratch_address( int device, void** out ) // the caller has made `device` current
{
	std::lock_guard<std::mutex> lock( g_scratch_mutex );
	*out = 0;
	if (device < 0 || device >= 64) return hipErrorInvalidDevice;
	if (!g_scratch_ptr[device])
	{
		void* p = 0;
		const hipError_t e = hipGetSymbolAddress( &p, HIP_SYMBOL( scratch ) );
		if (e != hipSuccess) return e;
		g_scratch_ptr[device] = p;
	}
	*out = g_scratch_ptr[device];
	return hipSuccess;
}

#define HIPCHK( call )                                                                                                         \
	do {                                                                                                                       \
		hipError_t e_ = ( call );                                                                                              \
		if (e_ != hipSuccess) { c->err = std::string( #call ) + ": " + hipGetErrorString( e_ ); return RTMI_ERR_HIP; }        \
	} while (0)

static int fail( rtmi_ctx* c, int code, const std::string& msg ) { if (c) c->err = msg; return code; }
static inline int nblocks( size_t n ) { return (int)( ( n + BLOCK - 1 ) / BLOCK ); }

template <typename T> static int dev_alloc( rtmi_ctx* c, T** p, size_t count, std::vector<void*>* track = 0 )
{
	*p = 0;
	if (count == 0) count = 1;
	void* q = 0;
	hipError_t e = hipMalloc( &q, count * sizeof( T ) );
	if (e != hipSuccess) { c->err = std::string( "hipMalloc: " ) + hipGetErrorString( e ); return RTMI_ERR_NOMEM; }
	*p = (T*)q;
	if (track) track->push_back( q );
	return RTMI_OK;
}
template <typename T> static int dev_upload( rtmi_ctx* c, T** p, const T* src, size_t count, std::vector<void*>* track )
{
	int r = dev_alloc( c, p, count, track );
	if (r) return r;
	if (count) HIPCHK( hipMemcpy( *p, src, count * sizeof( T ), hipMemcpyHostToDevice ) );
	return RTMI_OK;
}

static hipEvent_t get_event( rtmi_ctx* c ) // null when the runtime has no event to give: the measurement is skipped, not the work
{
	if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
	hipEvent_t e = 0;
	if (hipEventCreate( &e ) != hipSuccess) { c->err = "hipEventCreate failed: kernel timing incomplete"; return 0; }
	return e;
}
struct ScopedTimer // records a pair of events around a launch when profiling is on
{
	rtmi_ctx* c; int cls; hipEvent_t a = 0, b = 0;
	hipStream_t st;
	ScopedTimer( rtmi_ctx* c, int cls, hipStream_t on ) : c( c ), cls( cls ), st( on )
	{
		if (!c->prof_events) return;
		a = get_event( c ), b = get_event( c );
		if (a && b) (void)hipEventRecord( a, st );
		else { if (a) c->ev_pool.push_back( a ); if (b) c->ev_pool.push_back( b ); a = b = 0; }
	}
	~ScopedTimer() { if (a && b) { (void)hipEventRecord( b, st ); c->timed.push_back( { a, b, cls } ); } }
};
static void resolve_timers( rtmi_ctx* c )
{
	for (auto& t : c->timed)
	{
		float ms = 0;
		if (hipEventSynchronize( t.b ) == hipSuccess && hipEventElapsedTime( &ms, t.a, t.b ) == hipSuccess)
		{
			if (t.cls == 0)
			{
				c->stats.ms_extend += ms, c->stats.n_extend_launches++;
				float t0 = 0;
				if (c->prof_base && hipEventElapsedTime( &t0, c->prof_base, t.a ) == hipSuccess) c->ext_intervals.push_back( { t0, t0 + ms } );
			}
			else if (t.cls == 1) c->stats.ms_shade += ms, c->stats.n_shade_launches++;
			else if (t.cls == 2) c->stats.ms_connect += ms, c->stats.n_connect_launches++;
			else if (t.cls == 4)
			{
				c->stats.ms_walk += ms, c->stats.n_walk_windows++;
				float t0 = 0;
				if (c->prof_base && hipEventElapsedTime( &t0, c->prof_base, t.a ) == hipSuccess) c->walk_intervals.push_back( { t0, t0 + ms } );
			}
			else c->stats.ms_other += ms;
		}
		c->ev_pool.push_back( t.a ), c->ev_pool.push_back( t.b );
	}
	c->timed.clear();
}

static void free_scene( rtmi_ctx* c )
{
	for (void* p : c->scene_allocs) (void)hipFree( p );
	c->scene_allocs.clear();
	c->has_scene = false;
}
static void free_batch( BatchSlot& b )
{
	void* ps[] = { b.st[0].o_t, b.st[0].d_h, b.st[0].T_s, b.st[0].meta, b.st[1].o_t, b.st[1].d_h, b.st[1].T_s, b.st[1].meta,
		b.sq[0].o_t, b.sq[0].d_s, b.sq[0].c, b.sq[1].o_t, b.sq[1].d_s, b.sq[1].c, b.E, b.Es, b.wq[0], b.wq[1], b.rec[0], b.rec[1], b.srec[0], b.srec[1] };
	for (void* p : ps) if (p) (void)hipFree( p );
	b.st[0] = b.st[1] = PathState{}, b.sq[0] = b.sq[1] = ShadowQueue{}, b.E = 0, b.wq[0] = b.wq[1] = 0, b.batch = 0;
	b.rec[0] = b.rec[1] = b.srec[0] = b.srec[1] = 0, b.rec_cap = 0, b.Es = 0;
}
static void free_batches( rtmi_ctx* c ) { for (auto& b : c->slots) free_batch( b ); } // callers have synchronised the device
// Queue state per path in flight: 2 x 52 B path state, 16 B E, 32 B Es, 2 x 48 B shadow ray, 2 x 4 B walk index = 256 B, plus
// suspend records for one path in eight (4 x (6 + stack depth) words each).
static int ensure_batch( rtmi_ctx* c, BatchSlot& b, int S )
{
	if (b.batch >= S) return RTMI_OK;
	if (b.busy) { (void)hipEventSynchronize( b.acc_done ); b.busy = false; }
	free_batch( b );
	const size_t n0 = (size_t)S * c->W * c->H, n = n0 + c->carry_cap + (size_t)MAX_SLOTS * c->left_cap; // the queues also hold the paths carried in from earlier batches
	const size_t cap = n / 8 + 1024;
	b.rec_stride = c->walk_depth + REC_HEADER;
	int r = RTMI_OK;
	for (int k = 0; k < 2 && !r; k++)
		( r = dev_alloc( c, &b.st[k].o_t, n ) ) || ( r = dev_alloc( c, &b.st[k].d_h, n ) ) || ( r = dev_alloc( c, &b.st[k].T_s, n ) ) || ( r = dev_alloc( c, &b.st[k].meta, n ) ) ||
			( r = dev_alloc( c, &b.wq[k], n ) ) || ( r = dev_alloc( c, &b.sq[k].o_t, n + cap ) ) || ( r = dev_alloc( c, &b.sq[k].d_s, n + cap ) ) || ( r = dev_alloc( c, &b.sq[k].c, n + cap ) ) ||
			( r = dev_alloc( c, &b.rec[k], cap * b.rec_stride ) ) || ( r = dev_alloc( c, &b.srec[k], cap * b.rec_stride ) );
	if (!r) ( r = dev_alloc( c, &b.E, n0 ) ) || ( r = dev_alloc( c, &b.Es, 4 * n0 ) );
	if (r) { free_batch( b ); return r; }
	b.batch = S, b.rec_cap = (uint32_t)cap;
	return RTMI_OK;
}
// Everything enqueued by earlier rtmi_render calls (tails and accumulates on the slot streams) happens-before what is
// enqueued on the handle's stream after this call.
static Sinks make_sinks( rtmi_ctx* c, const BatchSlot& b ) { Sinks k; k.E = b.E, k.Es = b.Es, k.late = c->late, k.odd = c->odd, k.WH = (uint32_t)( c->W * c->H ); return k; }

// k_tail rounds of 32, 64, 128, ... bounces over the state set st[cur] (length ctr->n_state[cur], at most `upper`), enqueued
// blind on `ts` (a round over an empty queue returns at once); survivors are compacted between rounds, the last round has no
// limit.  An even number of rounds: the set that was current at the start is current, and empty, at the end.
// `drain`: these are the rounds the caller waits for (join_batches): the rounds with few live paths go to k_tail_wave /
// k_tail_team.  Which kernel a round belongs to is decided on the device (live count, draining flag): all are launched and the
// ones whose round it is not return at once -- every kernel pulls its paths from a pool, so any grid size is correct and an
// unexpectedly large round is only slower.
static int launch_tail_rounds( rtmi_ctx* c, hipStream_t ts, PathState* st, int cur, DevCounters* ctr, const Sinks& sinks, uint32_t upper, int first_round = 0, int last_round = 11, bool drain = false,
	int stop_late = 0 )
{
	const uint32_t max_waves = (uint32_t)( c->n_cus * c->tail_waves ), drain_waves = (uint32_t)c->n_cus * 16u;
	for (int round = first_round, len = 32 << first_round; round <= last_round; round++, len *= 2)
	{
		// The last round is where a path ENDS if it is still alive: the reference's Sample has no bounce limit (renderer.cpp:251) and
		// some paths never end -- total internal reflection inside the glass of albedo 1 whose absorption the `inside` quirk skips
		// (sample 220 of the default scene at 1280x720 holds one: the CPU restatement does not return from it either).  A kernel
		// that never finishes takes the GPU with it, so after the 65 504 bounces of the rounds before, the last round gives a path
		// `last_round_bounces` more (option; 0 = no limit, the reference's behaviour) and then drops it: it has added whatever it
		// was going to add many thousands of vertices ago.  The longest path seen to end by itself had 13 600 vertices.
		const int limit = round == 11 ? ( c->last_round_bounces > 0 ? c->last_round_bounces : 0x7FFFFFFF ) : len;
		const bool coop = drain && c->ds.wide && round > 0;
		const uint32_t wave_below = coop && c->tail_wave ? c->wave_below : 0u;
		const uint32_t team_below = std::max( wave_below, coop && c->tail_team ? c->team_below : 0u ); // k_tail yields at or below this
		// waves: enough for the round's upper bound at the density the kernel will choose at most (it reads the live count)
		const uint32_t waves = std::max( 1u, std::min( max_waves, ( upper + c->tail_lanes - 1 ) / c->tail_lanes ) );
		const uint32_t wpb = std::max( 1u, std::min( (uint32_t)c->tail_block, 65536u / ( (uint32_t)c->walk_depth * 256u ) ) ), blocks = ( waves + wpb - 1 ) / wpb; // <= 64 KiB of LDS per block
		const uint32_t lds_words = (uint32_t)c->walk_depth * 64u * wpb;
		// a draining round with few paths left belongs to k_tail_team (the device knows the live count and whether the host waits:
		// both kernels are launched, the one whose round it is not returns at once)
		hipLaunchKernelGGL( k_tail_decide, dim3( 1 ), dim3( 1 ), 0, ts, ctr, &ctr->n_state[cur], c->draining_dev, team_below, wave_below );
		if (c->tail_profile)
			hipLaunchKernelGGL( k_tail<true>, dim3( blocks ), dim3( 64 * wpb ), lds_words * sizeof( uint32_t ), ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit,
				round == 0 ? 0u : c->tail_dense, round == 0 ? 0u : c->tail_mid, c->tail_lanes, lds_words, c->draining_dev, team_below, ctr, stop_late );
		else
			hipLaunchKernelGGL( k_tail<false>, dim3( blocks ), dim3( 64 * wpb ), lds_words * sizeof( uint32_t ), ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit,
				round == 0 ? 0u : c->tail_dense, round == 0 ? 0u : c->tail_mid, c->tail_lanes, lds_words, c->draining_dev, team_below, ctr, stop_late );
		if (team_below > wave_below)
		{
			// (the draining kernels have the chip to themselves: 16 waves per CU whatever tail_waves says about tails beside a bulk)
			const uint32_t twaves = std::max( 1u, std::min( drain_waves, std::min( upper, team_below ) ) ), tblocks = ( twaves + 3 ) / 4;
			const size_t tlds = (size_t)c->walk_depth * 64u * sizeof( uint32_t ); // 4 waves x 16 teams, one column each
			if (c->tail_profile)
				hipLaunchKernelGGL( k_tail_team<true>, dim3( tblocks ), dim3( 256 ), tlds, ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit, c->draining_dev, team_below, wave_below, ctr );
			else
				hipLaunchKernelGGL( k_tail_team<false>, dim3( tblocks ), dim3( 256 ), tlds, ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit, c->draining_dev, team_below, wave_below, ctr );
		}
		if (wave_below)
		{
			// one wave per path, as many waves as the round can have paths (a wave that finds the pool empty leaves at once)
			// (the rounds of 4096 bounces and more are for a handful of paths: fewer waves, cheaper to dispatch when they find nothing)
			const uint32_t wwaves = std::max( 1u, std::min( round >= 7 ? 1024u : drain_waves, std::min( upper, wave_below ) ) ), wblocks = ( wwaves + WAVE_WG - 1 ) / WAVE_WG;
			if (c->tail_profile)
				hipLaunchKernelGGL( k_tail_wave<true>, dim3( wblocks ), dim3( 64 * WAVE_WG ), 0, ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit, c->draining_dev, wave_below, ctr );
			else
				hipLaunchKernelGGL( k_tail_wave<false>, dim3( wblocks ), dim3( 64 * WAVE_WG ), 0, ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit, c->draining_dev, wave_below, ctr );
		}
		if (c->log_bounces && drain) // diagnostics: the survivors of every draining round (synchronises)
		{
			uint32_t left = 0;
			(void)hipStreamSynchronize( ts );
			(void)hipMemcpy( &left, &ctr->n_state[cur ^ 1], 4, hipMemcpyDeviceToHost );
			fprintf( stderr, "rtmi: drain round %d (%d bounces): %u paths survive\n", round, limit, left );
		}
		hipLaunchKernelGGL( k_next_bounce, dim3( 1 ), dim3( 1 ), 0, ts, ctr, cur, (uint32_t*)0 );
		cur ^= 1;
	}
	return cur;
}

static int render_batch( rtmi_ctx* c, BatchSlot& b, int sample_begin, int S, int stride, uint64_t ticket );

static double now_ms() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
// host time from the call that starts finishing the paths in flight to the return of the blocking call that needed them
static void drain_closed( rtmi_ctx* c ) { if (c->drain_open) c->stats.ms_drain += now_ms() - c->drain_t0, c->stats.n_drains++, c->drain_open = false; }

// the camera-less batch of a lane whose batches handed paths on (see join_batches)
static int join_lane( rtmi_ctx* c, uint64_t ticket )
{
	(void)hipSetDevice( c->device );
	BatchSlot& b = c->slots[c->next_slot];
	c->next_slot = ( c->next_slot + 1 ) % c->n_slots;
	int r = render_batch( c, b, 0, 0, 1, ticket );
	if (!r) c->paths_carried = false;
	return r;
}
// `preview`: the caller wants the image as far as it has got (rtmi_finalize under option preview_reads): the accumulates in
// flight and the late sums so far, but the carried paths stay where they are and ride the next batch.
static int join_batches( rtmi_ctx* c, bool preview = false )
{
	rtmi_ctx* k = c->child;
	if (c->draining) *(volatile uint32_t*)c->draining = 1; // the caller is about to wait for the tails
	if (k && k->draining) *(volatile uint32_t*)k->draining = 1;
	const bool pending = c->carry_pending || ( k && k->carry_pending );
	if (pending && !c->drain_open) c->drain_open = true, c->drain_t0 = now_ms();
	if (pending && c->has_scene && preview)
	{
		if (c->chain->last_acc) HIPCHK( hipStreamWaitEvent( c->stream, c->chain->last_acc, 0 ) );
		ScopedTimer t( c, 3, c->stream );
		hipLaunchKernelGGL( k_fold_late, dim3( nblocks( (size_t)c->W * c->H ) ), dim3( BLOCK ), 0, c->stream, c->accum, c->late, c->odd, c->W * c->H );
		HIPCHK( hipGetLastError() );
		return RTMI_OK; // (carry_pending stays: the next full join finishes the carried paths)
	}
	if (pending && c->has_scene)
	{
		// The late paths that the last batches handed on have no batch to ride with: they get a batch of their own, without
		// camera rays (S = 0): wavefront bounces while they are many, then k_tail.  One per lane, enqueued side by side.
		const bool mine = c->carry_on && c->paths_carried && c->slots[c->next_slot].E, kids = k && k->carry_on && k->paths_carried && k->slots[k->next_slot].E;
		int rk = RTMI_OK, r = RTMI_OK;
		uint64_t t0;
		{ std::lock_guard<std::mutex> lock( c->chain->m ); t0 = c->chain->next_ticket; c->chain->next_ticket += ( mine ? 1 : 0 ) + ( kids ? 1 : 0 ); }
		std::thread th;
		bool threaded = false;
		if (kids && mine) { try { th = std::thread( [&] { rk = join_lane( k, t0 ); } ); threaded = true; } catch (...) {} } // (no thread to be had: one after the other, in ticket order)
		if (kids && !threaded) rk = join_lane( k, t0 );
		if (mine) r = join_lane( c, t0 + ( kids ? 1 : 0 ) );
		if (th.joinable()) th.join();
		if (rk) { c->err = k->err; return rk; }
		if (r) return r;
		if (c->chain->last_acc) HIPCHK( hipStreamWaitEvent( c->stream, c->chain->last_acc, 0 ) );
		// fold the late sums of all batches since the last read into the accumulator
		ScopedTimer t( c, 3, c->stream );
		hipLaunchKernelGGL( k_fold_late, dim3( nblocks( (size_t)c->W * c->H ) ), dim3( BLOCK ), 0, c->stream, c->accum, c->late, c->odd, c->W * c->H );
		HIPCHK( hipGetLastError() );
		c->carry_pending = false;
		if (k) k->carry_pending = false;
		return RTMI_OK;
	}
	if (c->chain->last_acc) HIPCHK( hipStreamWaitEvent( c->stream, c->chain->last_acc, 0 ) );
	return RTMI_OK;
}

// host-blocking: the handle's stream and every slot stream are idle afterwards
static int sync_all( rtmi_ctx* c )
{
	HIPCHK( hipStreamSynchronize( c->stream ) );
	if (c->side) HIPCHK( hipStreamSynchronize( c->side ) );
	for (int k = 0; k < MAX_SLOTS; k++) if (c->slots[k].tail_stream) { HIPCHK( hipStreamSynchronize( c->slots[k].tail_stream ) ); c->slots[k].busy = false; }
	if (c->child) { int r = sync_all( c->child ); if (r) { c->err = c->child->err; return r; } }
	if (!c->is_child) c->chain->last_acc = 0, drain_closed( c );
	return RTMI_OK;
}

// ---- options: one table for rtmi_set_option / rtmi_get_option and for the RTMI_<KEY> environment defaults ----------------
static void free_carry( rtmi_ctx* c )
{
	for (int k = 0; k < 2; k++)
	{
		void* ps[] = { c->carry[k].o_t, c->carry[k].d_h, c->carry[k].T_s, c->carry[k].meta };
		for (void* q : ps) if (q) (void)hipFree( q );
		c->carry[k] = PathState{};
	}
}
static int alloc_carry( rtmi_ctx* c )
{
	free_carry( c );
	int r = 0;
	for (int k = 0; k < 2 && !r; k++)
		r = dev_alloc( c, &c->carry[k].o_t, c->carry_cap ) || dev_alloc( c, &c->carry[k].d_h, c->carry_cap ) || dev_alloc( c, &c->carry[k].T_s, c->carry_cap ) || dev_alloc( c, &c->carry[k].meta, c->carry_cap );
	if (!r && hipMemset( c->carry_ctr, 0, sizeof( DevCounters ) ) != hipSuccess) r = RTMI_ERR_HIP;
	return r ? RTMI_ERR_NOMEM : RTMI_OK;
}
static void free_left( BatchSlot& b )
{
	void* lp[] = { b.left.o_t, b.left.d_h, b.left.T_s, b.left.meta };
	for (void* q : lp) if (q) (void)hipFree( q );
	b.left = PathState{};
}
static int alloc_left( rtmi_ctx* c, BatchSlot& b )
{
	free_left( b );
	if (dev_alloc( c, &b.left.o_t, c->left_cap ) || dev_alloc( c, &b.left.d_h, c->left_cap ) || dev_alloc( c, &b.left.T_s, c->left_cap ) || dev_alloc( c, &b.left.meta, c->left_cap )) return RTMI_ERR_NOMEM;
	return hipMemset( b.n_left, 0, 4 ) == hipSuccess ? RTMI_OK : RTMI_ERR_HIP;
}
// streams, events, counters and the slot's list of left-over late paths; idempotent
static int init_slot( rtmi_ctx* c, BatchSlot& b )
{
	if (b.ctr) return RTMI_OK;
	if (dev_alloc( c, &b.ctr, 1 ) || hipHostMalloc( (void**)&b.host_counts, 4 * sizeof( uint32_t ), hipHostMallocMapped ) != hipSuccess ||
		hipHostGetDevicePointer( (void**)&b.host_counts_dev, b.host_counts, 0 ) != hipSuccess) return RTMI_ERR_NOMEM;
	// (a low-priority stream would suit the tails, but a run with hipStreamCreateWithPriority stopped making progress here)
	bool ok = hipStreamCreateWithFlags( &b.tail_stream, hipStreamNonBlocking ) == hipSuccess;
	for (int i = 0; i < 4; i++) ok = ok && hipEventCreateWithFlags( &b.count_ev[i], hipEventDisableTiming ) == hipSuccess;
	ok = ok && hipEventCreateWithFlags( &b.bulk_done, hipEventDisableTiming ) == hipSuccess && hipEventCreateWithFlags( &b.acc_done, hipEventDisableTiming ) == hipSuccess;
	ok = ok && hipEventCreateWithFlags( &b.ev_shade, hipEventDisableTiming ) == hipSuccess && hipEventCreateWithFlags( &b.ev_any, hipEventDisableTiming ) == hipSuccess;
	if (!ok) return RTMI_ERR_HIP;
	(void)hipMemset( b.ctr, 0, sizeof( DevCounters ) );
	if (dev_alloc( c, &b.n_left, 1 )) return RTMI_ERR_NOMEM;
	return alloc_left( c, b );
}
static void set_max_blocks( rtmi_ctx* c )
{
	// blocks that stay resident per CU: 160 KiB LDS / per-block stack, at most 4 (16 waves per CU).  The walk is bound by
	// the memory system's random-access throughput, not by latency: measured on the 1M-triangle scene, 4 blocks per CU
	// run the walks 8 % faster than 6 (fewer rays in flight share each XCD's 4 MiB L2), 3 as fast as 4, 2 as slow as 6.
	const size_t per_block = (size_t)c->walk_depth * BLOCK * sizeof( uint32_t );
	int per_cu = (int)std::max<size_t>( 1, std::min<size_t>( 4, ( 160 * 1024 ) / per_block ) );
	if (c->blocks_per_cu_opt >= 1) per_cu = (int)std::min<size_t>( c->blocks_per_cu_opt, std::max<size_t>( 1, std::min<size_t>( 8, ( 160 * 1024 ) / per_block ) ) );
	c->max_blocks = c->n_cus * per_cu;
}

static int set_lanes( rtmi_ctx* c );
struct OptDesc { const char* key; long long lo, hi; };
static const OptDesc g_opts[] = {
	{ "log_bounces", 0, 1 }, { "overlap", 0, 1 }, { "wide", 0, 1 }, { "plane_pairs", 0, 1 }, { "carry", 0, 1 }, { "tail_lanes", 1, 64 }, { "tail_waves", 1, 32 }, { "sblock", 256, 1024 },
	{ "tail_block", 1, 16 }, { "tail_dense", 0, 1ll << 32 }, { "tail_mid", 0, 1ll << 32 }, { "slots", 1, MAX_SLOTS }, { "tail_paths", 0, 1ll << 32 },
	{ "carry_cap", 1, 1ll << 24 }, { "left_cap", 1, 1ll << 20 }, { "carry_bounce", 0, 1 << 20 }, { "step_budget", 0, 1 << 20 }, { "refill", 1, 64 },
	{ "batch", 0, 1 << 20 }, { "blocks_per_cu", 0, 8 }, { "grid_closest_hi", 0, 1ll << 32 }, { "grid_closest_lo", 0, 1ll << 32 }, { "grid_any_hi", 0, 1ll << 32 },
	{ "grid_any_lo", 0, 1ll << 32 }, { "check_launches", 0, 1 }, { "hook_kernel", 0, 2 }, { "tail_team", 0, 1 }, { "tail_profile", 0, 1 }, { "team_below", 0, 1 << 18 },
	{ "tail_wave", 0, 1 }, { "wave_below", 0, 1 << 18 }, { "veteran_rounds", -1, 10 }, { "carry_min", 0, 1 << RTMI_SLOT_BITS }, { "lanes", 1, 2 }, { "last_round_bounces", 0, 1 << 30 }, { "stop_late", 0, 1 }, { "small_paths", 1024, 1 << 22 }, { "count_lag", 1, 2 }, { "preview_reads", 0, 1 },
};
// `live`: the handle is in use (buffers exist, work may be in flight): wait, then resize what the option sizes
static int apply_option( rtmi_ctx* c, const char* key, long long v, bool live )
{
	const OptDesc* d = 0;
	for (const OptDesc& o : g_opts) if (!strcmp( o.key, key )) d = &o;
	if (!d) return fail( c, RTMI_ERR_INVALID, std::string( "unknown option " ) + key );
	if (v < d->lo || v > d->hi) return fail( c, RTMI_ERR_INVALID, std::string( "value out of range for option " ) + key );
	auto is = [&]( const char* k ) { return !strcmp( key, k ); };
	if (is( "log_bounces" )) c->log_bounces = v != 0;
	else if (is( "overlap" )) c->overlap_on = v != 0 && c->side != 0;
	else if (is( "wide" )) c->wide_on = v != 0;
	else if (is( "plane_pairs" )) c->plane_pairs_on = v != 0;
	else if (is( "carry" )) { c->carry_on = v != 0; if (!c->tail_paths_set) c->tail_paths = c->carry_on ? 1u << 22 : c->small_paths; }
	else if (is( "tail_lanes" )) { if (v & ( v - 1 )) return fail( c, RTMI_ERR_INVALID, "tail_lanes must be a power of two" ); c->tail_lanes = (int)v; }
	else if (is( "tail_waves" )) c->tail_waves = (int)v;
	else if (is( "sblock" )) { if (v != 256 && v != 512 && v != 1024) return fail( c, RTMI_ERR_INVALID, "sblock must be 256, 512 or 1024" ); c->sblock = (int)v; }
	else if (is( "tail_block" )) c->tail_block = (int)v;
	else if (is( "tail_dense" )) c->tail_dense = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll );
	else if (is( "tail_mid" )) c->tail_mid = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll );
	else if (is( "tail_paths" )) c->tail_paths = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll ), c->tail_paths_set = true;
	else if (is( "carry_bounce" )) c->carry_bounce = (int)v;
	else if (is( "step_budget" )) c->budget = (int)v;
	else if (is( "refill" )) c->refill = (int)v;
	else if (is( "batch" )) c->batch_opt = (int)v;
	else if (is( "grid_closest_hi" )) c->grid_closest[0] = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll );
	else if (is( "grid_closest_lo" )) c->grid_closest[1] = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll );
	else if (is( "grid_any_hi" )) c->grid_any[0] = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll );
	else if (is( "grid_any_lo" )) c->grid_any[1] = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll );
	else if (is( "check_launches" )) c->check_launches = v != 0;
	else if (is( "hook_kernel" )) c->hook_kernel = (int)v;
	else if (is( "tail_team" )) c->tail_team = v != 0;
	else if (is( "tail_profile" )) c->tail_profile = v != 0;
	else if (is( "team_below" )) c->team_below = (uint32_t)v;
	else if (is( "tail_wave" )) c->tail_wave = v != 0;
	else if (is( "wave_below" )) c->wave_below = (uint32_t)v;
	else if (is( "veteran_rounds" )) c->veteran_rounds = (int)v;
	else if (is( "carry_min" )) c->carry_min = (uint32_t)v;
	else if (is( "small_paths" )) c->small_paths = (uint32_t)v;
	else if (is( "count_lag" )) c->count_lag = (int)v;
	else if (is( "stop_late" )) c->stop_late = v != 0;
	else if (is( "preview_reads" )) c->preview_reads = v != 0;
	else if (is( "last_round_bounces" )) c->last_round_bounces = (int)v;
	else if (is( "lanes" )) { c->lanes = (int)v; if (live) { int r = set_lanes( c ); if (r) return r; } }
	else if (is( "blocks_per_cu" )) { c->blocks_per_cu_opt = (int)v; if (live) set_max_blocks( c ); }
	else if (is( "slots" ))
	{
		c->n_slots = (int)v, c->next_slot = 0, c->fit_samples = 0;
		if (live) for (int k = 0; k < c->n_slots; k++) { int r = init_slot( c, c->slots[k] ); if (r) return r; }
	}
	else if (is( "carry_cap" ) || is( "left_cap" ))
	{
		( is( "carry_cap" ) ? c->carry_cap : c->left_cap ) = (uint32_t)v;
		c->fit_samples = 0;
		if (live)
		{
			free_batches( c ); // the queues of a batch are sized for the paths carried in
			if (c->child) free_batches( c->child ); // (its capacities follow at its next use: sync_child)
			int r = is( "carry_cap" ) ? alloc_carry( c ) : RTMI_OK;
			if (!r && is( "left_cap" )) for (auto& b : c->slots) if (b.ctr && !r) r = alloc_left( c, b );
			if (r) return r;
		}
	}
	return RTMI_OK;
}
static long long read_option( rtmi_ctx* c, const char* key, bool& found )
{
	auto is = [&]( const char* k ) { return !strcmp( key, k ); };
	found = true;
	if (is( "log_bounces" )) return c->log_bounces; if (is( "overlap" )) return c->overlap_on; if (is( "wide" )) return c->wide_on; if (is( "plane_pairs" )) return c->plane_pairs_on; if (is( "carry" )) return c->carry_on;
	if (is( "tail_lanes" )) return c->tail_lanes; if (is( "tail_waves" )) return c->tail_waves; if (is( "sblock" )) return c->sblock; if (is( "tail_block" )) return c->tail_block;
	if (is( "tail_dense" )) return c->tail_dense; if (is( "tail_mid" )) return c->tail_mid; if (is( "slots" )) return c->n_slots; if (is( "tail_paths" )) return c->tail_paths;
	if (is( "carry_cap" )) return c->carry_cap; if (is( "left_cap" )) return c->left_cap; if (is( "carry_bounce" )) return c->carry_bounce; if (is( "step_budget" )) return c->budget;
	if (is( "refill" )) return c->refill; if (is( "batch" )) return c->batch_opt; if (is( "blocks_per_cu" )) return c->blocks_per_cu_opt;
	if (is( "grid_closest_hi" )) return c->grid_closest[0]; if (is( "grid_closest_lo" )) return c->grid_closest[1]; if (is( "grid_any_hi" )) return c->grid_any[0]; if (is( "grid_any_lo" )) return c->grid_any[1];
	if (is( "check_launches" )) return c->check_launches; if (is( "hook_kernel" )) return c->hook_kernel; if (is( "tail_team" )) return c->tail_team; if (is( "tail_profile" )) return c->tail_profile;
	if (is( "team_below" )) return c->team_below;
	if (is( "tail_wave" )) return c->tail_wave; if (is( "wave_below" )) return c->wave_below; if (is( "veteran_rounds" )) return c->veteran_rounds; if (is( "carry_min" )) return c->carry_min; if (is( "lanes" )) return c->lanes; if (is( "last_round_bounces" )) return c->last_round_bounces; if (is( "stop_late" )) return c->stop_late; if (is( "small_paths" )) return c->small_paths; if (is( "count_lag" )) return c->count_lag; if (is( "preview_reads" )) return c->preview_reads;
	found = false;
	return 0;
}

// ---- the second lane -----------------------------------------------------------------------------------------------------
// Everything a batch reads from its context, copied from the parent before the child is used (the caller may have changed
// options, camera, scene or accumulator since); buffers that the options size are re-made when the sizes differ.
static int sync_child( rtmi_ctx* c )
{
	rtmi_ctx* k = c->child;
	if (!k) return RTMI_OK;
	k->ds = c->ds, k->has_scene = c->has_scene, k->stack_depth = c->stack_depth, k->walk_depth = c->walk_depth;
	k->overlap_on = c->overlap_on && k->side != 0, k->wide_on = c->wide_on, k->plane_pairs_on = c->plane_pairs_on, k->refill = c->refill, k->max_blocks = c->max_blocks;
	for (int i = 0; i < 2; i++) k->grid_closest[i] = c->grid_closest[i], k->grid_any[i] = c->grid_any[i];
	k->cam = c->cam, k->seeds = c->seeds, k->accum = c->accum, k->late = c->late, k->odd = c->odd;
	k->carry_on = c->carry_on, k->carry_min = c->carry_min, k->small_paths = c->small_paths, k->count_lag = c->count_lag, k->carry_bounce = c->carry_bounce, k->tail_paths = c->tail_paths;
	k->tail_lanes = c->tail_lanes, k->sblock = c->sblock, k->tail_waves = c->tail_waves, k->tail_block = c->tail_block, k->tail_dense = c->tail_dense, k->tail_mid = c->tail_mid;
	k->log_bounces = c->log_bounces, k->budget = c->budget, k->batch_opt = c->batch_opt, k->blocks_per_cu_opt = c->blocks_per_cu_opt, k->check_launches = c->check_launches;
	k->veteran_rounds = c->veteran_rounds, k->tail_team = c->tail_team, k->tail_profile = c->tail_profile, k->team_below = c->team_below, k->tail_wave = c->tail_wave, k->wave_below = c->wave_below;
	k->prof_events = c->prof_events, k->count_nodes = c->count_nodes, k->prof_base = c->prof_base, k->stop_late = c->stop_late, k->last_round_bounces = c->last_round_bounces, k->preview_reads = c->preview_reads;
	int r = RTMI_OK;
	if (k->carry_cap != c->carry_cap) { r = apply_option( k, "carry_cap", c->carry_cap, true ); if (r) return r; }
	if (k->left_cap != c->left_cap) { r = apply_option( k, "left_cap", c->left_cap, true ); if (r) return r; }
	if (k->n_slots != c->n_slots) { r = apply_option( k, "slots", c->n_slots, true ); if (r) return r; }
	return RTMI_OK;
}
// lanes = 2: make the child; lanes = 1: retire it.  The caller has joined and synchronised.
static int set_lanes( rtmi_ctx* c )
{
	if (c->is_child) return RTMI_OK;
	if (c->lanes >= 2 && !c->child)
	{
		rtmi_ctx* k = 0;
		int r = rtmi_create( &k, c->device, c->W, c->H );
		if (r) return fail( c, r, "option lanes: the second lane could not be created" );
		k->is_child = true, k->lanes = 1, k->chain = c->chain;
		k->child_seeds = k->seeds, k->child_late = k->late, k->child_odd = k->odd; // its own, unused while it shares the parent's
		c->child = k;
		if (!c->ev_fork && ( hipEventCreateWithFlags( &c->ev_fork, hipEventDisableTiming ) != hipSuccess || hipEventCreateWithFlags( &c->ev_join, hipEventDisableTiming ) != hipSuccess ))
			return fail( c, RTMI_ERR_HIP, "hipEventCreate" );
	}
	else if (c->lanes < 2 && c->child)
	{
		rtmi_ctx* k = c->child;
		k->seeds = k->child_seeds, k->late = k->child_late, k->odd = k->child_odd, k->accum = k->accum_own, k->ds = DScene{}, k->has_scene = false;
		c->child = 0;
		rtmi_destroy( k );
	}
	return RTMI_OK;
}
static void merge_stats( rtmi_stats& a, rtmi_stats& b ) // a += b, b = 0
{
	a.primary_samples += b.primary_samples, a.extend_rays += b.extend_rays, a.shadow_rays += b.shadow_rays, a.extend_pairs += b.extend_pairs, a.extend_prims += b.extend_prims;
	a.shadow_pairs += b.shadow_pairs, a.shadow_prims += b.shadow_prims, a.bounce_launches += b.bounce_launches;
	a.ms_extend += b.ms_extend, a.ms_shade += b.ms_shade, a.ms_connect += b.ms_connect, a.ms_other += b.ms_other, a.ms_walk += b.ms_walk;
	a.n_extend_launches += b.n_extend_launches, a.n_shade_launches += b.n_shade_launches, a.n_connect_launches += b.n_connect_launches, a.n_walk_windows += b.n_walk_windows;
	a.extend_walkers += b.extend_walkers, a.shadow_walkers += b.shadow_walkers, a.suspended += b.suspended;
	a.max_extend_steps = std::max( a.max_extend_steps, b.max_extend_steps ), a.max_shadow_steps = std::max( a.max_shadow_steps, b.max_shadow_steps );
	for (int i = 0; i < 12; i++) a.tail_profile[i] += b.tail_profile[i];
	b = rtmi_stats{};
}

extern "C" {

const char* rtmi_version( void ) { return "rtmi 0.2 (gfx950)"; }
const char* rtmi_last_error( rtmi_ctx* c ) { return c ? c->err.c_str() : "null context"; }

int rtmi_create( rtmi_ctx** out, int device_id, int width, int height )
{
	if (!out || width <= 0 || height <= 0 || (long long)width * height > ( 1ll << RTMI_SLOT_BITS )) return RTMI_ERR_INVALID; // a batch is at least one sample of every pixel, at most 2^RTMI_SLOT_BITS paths (META_SLOT_MASK)
	*out = 0;
	rtmi_ctx* c = new rtmi_ctx;
	c->device = device_id, c->W = width, c->H = height;
	auto bail = [&]( int code ) { delete c; return code; };
	int ndev = 0;
	if (hipGetDeviceCount( &ndev ) != hipSuccess || ndev <= 0) { fprintf( stderr, "rtmi_create: no HIP device available\n" ); return bail( RTMI_ERR_HIP ); }
	if (device_id < 0 || device_id >= ndev) return bail( RTMI_ERR_INVALID );
	if (hipSetDevice( device_id ) != hipSuccess) return bail( RTMI_ERR_HIP );
	if (hipStreamCreateWithFlags( &c->own_stream, hipStreamNonBlocking ) != hipSuccess) return bail( RTMI_ERR_HIP );
	c->stream = c->own_stream;
	if (scratch_address( device_id, &c->noalloc_scratch ) != hipSuccess) { (void)hipStreamDestroy( c->own_stream ); return bail( RTMI_ERR_HIP ); }
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties( &prop, device_id ) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
		if (hipStreamCreateWithFlags( &c->side, hipStreamNonBlocking ) != hipSuccess) c->overlap_on = false, c->side = 0;
		const uint32_t lanes = (uint32_t)c->n_cus * 1024u; // 4 blocks of 256 per CU
		c->grid_closest[0] = 61u * lanes, c->grid_closest[1] = 19u * lanes, c->grid_any[0] = 76u * lanes, c->grid_any[1] = 30u * lanes;
		c->tail_paths = 1u << 22;
		// batches in flight: each uses its own stream, and streams beyond the runtime's hardware queues (GPU_MAX_HW_QUEUES,
		// 4 unless the process raises it before HIP starts) serialise behind each other
		// defaults from the environment (RTMI_<KEY>, upper case); a bad value is ignored with a note, not fatal
		for (const OptDesc& o : g_opts)
		{
			std::string name = "RTMI_";
			for (const char* k = o.key; *k; k++) name += (char)toupper( *k );
			const char* e = getenv( name.c_str() );
			if (!e) continue;
			const long long v = *e ? atoll( e ) : 1;
			if (apply_option( c, o.key, v, false ) != RTMI_OK) fprintf( stderr, "rtmi_create: ignoring %s=%s (%s)\n", name.c_str(), e, c->err.c_str() ), c->err.clear();
		}
		// earlier spellings of three of them
		if (getenv( "RTMI_GRID_CLOSEST" )) sscanf( getenv( "RTMI_GRID_CLOSEST" ), "%u,%u", &c->grid_closest[0], &c->grid_closest[1] );
		if (getenv( "RTMI_GRID_ANY" )) sscanf( getenv( "RTMI_GRID_ANY" ), "%u,%u", &c->grid_any[0], &c->grid_any[1] );
		// Two lanes pay only when their streams do not share hardware queues: 2 x (main, side, 4 tails) + the caller's.  HIP gives a
		// process GPU_MAX_HW_QUEUES of them (4 unless the process raised it before the runtime started): measured at 20 bench
		// steps, 2 lanes on 16 queues +8.5 % rays/s, on 4 queues -2 %.
		if (!getenv( "RTMI_LANES" )) { const char* q = getenv( "GPU_MAX_HW_QUEUES" ); c->lanes = q && atoi( q ) >= 12 ? 2 : 1; }
		// beside another lane's kernels 3 resident walk blocks per CU and 8 tail waves per CU do better than 4 and 16
		// (20 bench steps: 10 358 -> 10 623 Mrays/s; alone, 4 and 16 are the optimum)
		if (c->lanes == 2) { if (!getenv( "RTMI_BLOCKS_PER_CU" )) c->blocks_per_cu_opt = 3; if (!getenv( "RTMI_TAIL_WAVES" )) c->tail_waves = 8; }
		if (!getenv( "RTMI_SLOTS" ))
		{
			const char* q = getenv( "GPU_MAX_HW_QUEUES" );
			c->n_slots = c->carry_on || !q || atoi( q ) < 7 ? 4 : atoi( q ) >= 13 ? 12 : 6; // whole tails (carry = 0) last 0.5 s: more batches in flight
		}
	}
	const size_t n = (size_t)width * height;
	if (dev_alloc( c, &c->seeds, n ) || dev_alloc( c, &c->accum_own, n ) || dev_alloc( c, &c->pixels, n ) || dev_alloc( c, &c->pixels2, n ) ||
		dev_alloc( c, &c->gamma_lut, 256 )) { rtmi_destroy( c ); return RTMI_ERR_NOMEM; }
	c->accum = c->accum_own;
	if (hipHostMalloc( (void**)&c->draining, 64, hipHostMallocMapped ) != hipSuccess || hipHostGetDevicePointer( (void**)&c->draining_dev, c->draining, 0 ) != hipSuccess) { rtmi_destroy( c ); return RTMI_ERR_NOMEM; }
	*c->draining = 0;
	{
		int r = dev_alloc( c, &c->late, 4 * n ) || dev_alloc( c, &c->odd, n ) || dev_alloc( c, &c->carry_ctr, 1 );
		if (!r) r = alloc_carry( c );
		if (r) { rtmi_destroy( c ); return RTMI_ERR_NOMEM; }
		(void)hipMemset( c->late, 0, 4 * n * sizeof( long long ) );
		(void)hipMemset( c->odd, 0, n * sizeof( float4 ) );
	}
	for (int k = 0; k < c->n_slots; k++) { int r = init_slot( c, c->slots[k] ); if (r) { rtmi_destroy( c ); return r; } }
	(void)hipMemset( c->accum_own, 0, n * sizeof( float4 ) );
	(void)hipMemset( c->seeds, 0, n * sizeof( uint32_t ) );
	// default camera, camera.h:20-23
	const float aspect = (float)width / (float)height;
	c->cam.pos = make_float4( 0, 0, -2, 0 ), c->cam.tl = make_float4( -aspect, 1, 0, 0 ), c->cam.tr = make_float4( aspect, 1, 0, 0 ), c->cam.bl = make_float4( -aspect, -1, 0, 0 );
	*out = c;
	return RTMI_OK; // (the second lane, if asked for, is made at the first render that can use it: rtmi_render)
}

int rtmi_destroy( rtmi_ctx* c )
{
	if (!c) return RTMI_ERR_INVALID;
	(void)hipSetDevice( c->device );
	(void)hipDeviceSynchronize();
	if (c->child) { c->lanes = 1; (void)set_lanes( c ); }
	if (c->prof_base && !c->is_child) (void)hipEventDestroy( c->prof_base );
	if (c->ev_fork) (void)hipEventDestroy( c->ev_fork );
	if (c->ev_join) (void)hipEventDestroy( c->ev_join );
	resolve_timers( c );
	free_scene( c ), free_batches( c ), free_carry( c );
	void* ps[] = { c->seeds, c->accum_own, c->pixels, c->pixels2, c->gamma_lut, c->frames, c->late, c->odd, c->carry_ctr, c->hook_buf };
	for (void* p : ps) if (p) (void)hipFree( p );
	for (auto& b : c->slots)
	{
		if (b.ctr) (void)hipFree( b.ctr );
		free_left( b );
		if (b.n_left) (void)hipFree( b.n_left );
		if (b.host_counts) (void)hipHostFree( b.host_counts );
		for (int i = 0; i < 4; i++) if (b.count_ev[i]) (void)hipEventDestroy( b.count_ev[i] );
		if (b.bulk_done) (void)hipEventDestroy( b.bulk_done );
		if (b.acc_done) (void)hipEventDestroy( b.acc_done );
		if (b.ev_shade) (void)hipEventDestroy( b.ev_shade );
		if (b.ev_any) (void)hipEventDestroy( b.ev_any );
		if (b.tail_stream) (void)hipStreamDestroy( b.tail_stream );
	}
	if (c->draining) (void)hipHostFree( c->draining );
	for (auto e : c->ev_pool) (void)hipEventDestroy( e );
	if (c->own_stream) (void)hipStreamDestroy( c->own_stream );
	if (c->side) (void)hipStreamDestroy( c->side );
	delete c;
	return RTMI_OK;
}

int rtmi_set_option( rtmi_ctx* c, const char* key, long long value )
{
	if (!c || !key) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c ); if (r) return r; } // what is in flight finishes under the settings it was started with
	{ int r = sync_all( c ); if (r) return r; }
	return apply_option( c, key, value, true );
}
int rtmi_get_option( rtmi_ctx* c, const char* key, long long* value_out )
{
	if (!c || !key || !value_out) return RTMI_ERR_INVALID;
	bool found;
	*value_out = read_option( c, key, found );
	return found ? RTMI_OK : fail( c, RTMI_ERR_INVALID, std::string( "unknown option " ) + key );
}

// the value is the stream: NULL is HIP's null stream (a torch process that never leaves its default stream passes 0 here)
int rtmi_set_stream( rtmi_ctx* c, void* s )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c ); if (r) return r; }
	int r = sync_all( c );
	if (r) return r;
	c->stream = (hipStream_t)s;
	return RTMI_OK;
}
int rtmi_use_own_stream( rtmi_ctx* c )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c ); if (r) return r; }
	int r = sync_all( c );
	if (r) return r;
	c->stream = c->own_stream;
	return RTMI_OK;
}

int rtmi_synchronize( rtmi_ctx* c )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	int r = sync_all( c );
	if (r) return r;
	resolve_timers( c );
	return RTMI_OK;
}

int rtmi_flush( rtmi_ctx* c )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	return join_batches( c );
}

// ---- 4-wide nodes with quantised boxes over the pair array (rt_device.h: wide_step) --------------------------------
// Collapse: the node of pair p starts with p's two children; while it has room, the interior child with the largest
// surface area is replaced by its own two children.  Each plane of a child box becomes one byte q with
// fma(q, 2^e, origin) <= exact lower bound (>= exact upper bound), checked here with the fma the kernel executes.
// Also writes the exact box of every leaf at leafbox[2 * first].  Returns the deepest stack a traversal can need
// (sum over a root-to-leaf chain of children - 1), or -1.
static int build_wide_nodes( const std::vector<float4>& pairs, uint32_t root_pair, int nbvh, std::vector<float4>& wide, std::vector<float4>& leafbox )
{
	struct Child { float mn[3], mx[3]; uint32_t ref; };
	struct Work { uint32_t pair, node; int above; };
	auto child_of = [&]( uint32_t pair, int side ) {
		const float4 a = pairs[4 * (size_t)pair + 2 * side], b = pairs[4 * (size_t)pair + 2 * side + 1];
		Child ch; ch.mn[0] = a.x, ch.mn[1] = a.y, ch.mn[2] = a.z, ch.mx[0] = a.w, ch.mx[1] = b.x, ch.mx[2] = b.y;
		memcpy( &ch.ref, &b.z, 4 );
		return ch; };
	auto area = []( const Child& ch ) {
		const double dx = (double)ch.mx[0] - ch.mn[0], dy = (double)ch.mx[1] - ch.mn[1], dz = (double)ch.mx[2] - ch.mn[2];
		return dx * dy + dy * dz + dz * dx; };
	leafbox.assign( (size_t)std::max( nbvh, 1 ) * 2, make_float4( 0, 0, 0, 0 ) );
	wide.assign( 4, make_float4( 0, 0, 0, 0 ) );
	std::vector<Work> work;
	work.push_back( { root_pair, 0u, 0 } );
	int deepest = 0;
	while (!work.empty())
	{
		const Work w = work.back();
		work.pop_back();
		Child ch[4];
		int n = 2;
		ch[0] = child_of( w.pair, 0 ), ch[1] = child_of( w.pair, 1 );
		while (n < 4)
		{
			int best = -1;
			double bestArea = -1;
			for (int i = 0; i < n; i++) if (!( ch[i].ref & RT_LEAF ) && area( ch[i] ) > bestArea) best = i, bestArea = area( ch[i] );
			if (best < 0) break;
			const uint32_t p = ch[best].ref;
			ch[best] = child_of( p, 0 ), ch[n++] = child_of( p, 1 );
		}
		const int above = w.above + n - 1;
		deepest = std::max( deepest, above );
		float o[3];
		uint32_t ebits = 0, lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
		for (int ax = 0; ax < 3; ax++)
		{
			float mn = ch[0].mn[ax], mx = ch[0].mx[ax];
			for (int i = 1; i < n; i++) mn = std::min( mn, ch[i].mn[ax] ), mx = std::max( mx, ch[i].mx[ax] );
			if (!( mn <= mx ) || !std::isfinite( mn ) || !std::isfinite( mx )) return -1;
			o[ax] = mn;
			int e = 1; // biased exponent of the step 2^(e-127); smallest with 255 steps >= extent
			const double extent = (double)mx - (double)mn;
			while (e < 254 && 255.0 * std::ldexp( 1.0, e - 127 ) < extent) e++;
			for (;; e++)
			{
				if (e > 254) return -1;
				const float step = std::ldexp( 1.0f, e - 127 );
				bool ok = true;
				uint32_t l = 0, h = 0;
				for (int i = 0; i < 4 && ok; i++)
				{
					int ql = 255, qh = 0; // an empty slot: inverted box
					if (i < n)
					{
						ql = (int)std::floor( ( (double)ch[i].mn[ax] - mn ) / step ), qh = (int)std::ceil( ( (double)ch[i].mx[ax] - mn ) / step );
						ql = std::max( 0, std::min( 255, ql ) ), qh = std::max( 0, qh );
						while (ql > 0 && std::fmaf( (float)ql, step, mn ) > ch[i].mn[ax]) ql--;
						while (qh <= 255 && std::fmaf( (float)std::min( qh, 255 ), step, mn ) < ch[i].mx[ax]) qh++;
						if (qh > 255 || std::fmaf( (float)ql, step, mn ) > ch[i].mn[ax]) ok = false;
					}
					l |= (uint32_t)ql << ( 8 * i ), h |= (uint32_t)( qh & 255 ) << ( 8 * i );
				}
				if (ok) { lo[ax] = l, hi[ax] = h; break; }
			}
			ebits |= (uint32_t)e << ( 8 * ax );
		}
		// an empty slot: inverted box (never passes unless rounding collapses it) and the first leaf position as a one-primitive
		// leaf, so that even then the walk only tests a primitive it would confirm against that leaf's exact box anyway
		uint32_t ref[4] = { RT_LEAF, RT_LEAF, RT_LEAF, RT_LEAF };
		for (int i = 0; i < n; i++)
		{
			if (ch[i].ref & RT_LEAF)
			{
				ref[i] = ch[i].ref;
				const size_t first = ch[i].ref & RT_LEAF_FIRST_MASK;
				leafbox[2 * first] = make_float4( ch[i].mn[0], ch[i].mn[1], ch[i].mn[2], ch[i].mx[0] );
				leafbox[2 * first + 1] = make_float4( ch[i].mx[1], ch[i].mx[2], 0, 0 );
			}
			else
			{
				ref[i] = (uint32_t)( wide.size() / 4 );
				wide.resize( wide.size() + 4, make_float4( 0, 0, 0, 0 ) );
				work.push_back( { ch[i].ref, ref[i], above } );
			}
		}
		auto f = []( uint32_t u ) { float v; memcpy( &v, &u, 4 ); return v; };
		float4* q = &wide[4 * (size_t)w.node];
		q[0] = make_float4( o[0], o[1], o[2], f( ebits ) );
		q[1] = make_float4( f( ref[0] ), f( ref[1] ), f( ref[2] ), f( ref[3] ) );
		q[2] = make_float4( f( lo[0] ), f( lo[1] ), f( lo[2] ), f( hi[0] ) );
		q[3] = make_float4( f( hi[1] ), f( hi[2] ), 0, 0 );
	}
	return deepest;
}

// Validates the reference-layout node array and repacks it: pair k = nodes[3+2k], nodes[4+2k] (the reference
// allocates children in adjacent pairs starting at node 3), interior leftFirst -> pair index, leaf -> encoded range.
int rtmi_upload_scene( rtmi_ctx* c, const rtmi_scene_desc* d )
{
	if (!c || !d) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c ); if (r) return r; } // paths in flight finish in the scene they were started in
	{ int r = sync_all( c ); if (r) return r; }
	if (d->n_quads < 0 || d->n_spheres < 0 || d->n_cubes < 0 || d->n_planes < 0 || d->n_triangles < 0) return fail( c, RTMI_ERR_INVALID, "negative primitive count" );
	const int nbvh = d->n_spheres + d->n_triangles;
	const long long total = (long long)d->n_quads + d->n_spheres + d->n_cubes + d->n_planes + d->n_triangles;
	if (total > ( 1ll << 30 )) return fail( c, RTMI_ERR_INVALID, "too many primitives" );
	if (nbvh > (int)RT_LEAF_FIRST_MASK) return fail( c, RTMI_ERR_BVH, "more than 2^25-1 BVH primitives" );
	if (total > 0 && ( !d->primitives || !d->albedo || !d->material_type || !d->ior || !d->absorption )) return fail( c, RTMI_ERR_INVALID, "null per-primitive array" );
	if (( d->n_quads && ( !d->quad_inv_T || !d->quad_size ) ) || ( d->n_spheres && ( !d->sphere_info || !d->sphere_invr ) ) ||
		( d->n_cubes && ( !d->cube_M || !d->cube_inv_M || !d->cube_b ) ) || ( d->n_triangles && !d->tri_verts )) return fail( c, RTMI_ERR_INVALID, "null geometry array" );
	if (nbvh > 0 && ( !d->nodes || !d->prim_idx || d->n_nodes < 3 )) return fail( c, RTMI_ERR_BVH, "BVH arrays missing" );
	if (d->texture && ( d->tex_w != 128 || d->tex_h != 64 )) return fail( c, RTMI_ERR_INVALID, "texture must be 128x64 (primitive.h:179-181)" );

	DScene ds = {};
	std::vector<void*> allocs;
	auto bail = [&]( int code ) { for (void* p : allocs) (void)hipFree( p ); return code; };
	int r;
	int maxDepth = 0, wideNeed = 0;
	ds.n_quads = d->n_quads, ds.n_spheres = d->n_spheres, ds.n_cubes = d->n_cubes, ds.n_planes = d->n_planes, ds.n_tris = d->n_triangles, ds.n_bvh_prims = nbvh;
	if (nbvh > 0)
	{
		const int nn = d->n_nodes;
		if (( nn - 3 ) % 2 != 0) return fail( c, RTMI_ERR_BVH, "n_nodes must be 3 + 2*pairs" );
		const int npairs = ( nn - 3 ) / 2;
		std::vector<float4> pairs( (size_t)( npairs > 0 ? npairs : 1 ) * 4, make_float4( 0, 0, 0, 0 ) );
		// A leaf reference holds count - 1 in 6 bits.  The reference's builder stops splitting whenever no split lowers the SAH
		// cost (bvh.h:344), so coincident or degenerate geometry gives leaves of any size, which its traversal walks like any
		// other (bvh.h:513-523).  Such a leaf becomes a small balanced subtree of synthetic pairs appended behind the real ones:
		// every synthetic child carries the LEAF'S OWN box and the chunks cover its primitives in order, 64 at a time, so box
		// tests, primitive order and the (t, position) rule are what they were.
		std::function<uint32_t( int, int, const rtmi_bvh_node& )> chunked = [&]( int first, int count, const rtmi_bvh_node& n ) -> uint32_t {
			if (count <= RT_LEAF_MAX_COUNT) return RT_LEAF | ( (uint32_t)( count - 1 ) << RT_LEAF_FIRST_BITS ) | (uint32_t)first;
			const int chunks = ( count + RT_LEAF_MAX_COUNT - 1 ) / RT_LEAF_MAX_COUNT, lc = ( chunks / 2 ) * RT_LEAF_MAX_COUNT;
			const size_t idx = pairs.size() / 4;
			pairs.resize( pairs.size() + 4, make_float4( 0, 0, 0, 0 ) );
			const uint32_t rl = chunked( first, lc, n ), rr = chunked( first + lc, count - lc, n );
			const int cl = lc, cr = count - lc;
			float4 q1 = make_float4( n.aabbMaxy, n.aabbMaxz, 0, 0 ), q3 = q1;
			memcpy( &q1.z, &rl, 4 ), memcpy( &q1.w, &cl, 4 ), memcpy( &q3.z, &rr, 4 ), memcpy( &q3.w, &cr, 4 );
			pairs[4 * idx] = pairs[4 * idx + 2] = make_float4( n.aabbMinx, n.aabbMiny, n.aabbMinz, n.aabbMaxx );
			pairs[4 * idx + 1] = q1, pairs[4 * idx + 3] = q3;
			return (uint32_t)idx; };
		auto encode = [&]( const rtmi_bvh_node& n, uint32_t& ref ) -> bool {
			if (n.primitiveCount > 0)
			{
				if (n.leftFirst < 0 || (long long)n.leftFirst + n.primitiveCount > nbvh) return false;
				ref = chunked( n.leftFirst, n.primitiveCount, n );
				return true;
			}
			if (n.primitiveCount < 0 || n.leftFirst < 3 || n.leftFirst + 1 >= nn || ( ( n.leftFirst - 3 ) & 1 )) return false;
			ref = (uint32_t)( ( n.leftFirst - 3 ) / 2 );
			return true; };
		const rtmi_bvh_node& root = d->nodes[2];
		if (!encode( root, ds.root_ref )) return fail( c, RTMI_ERR_BVH, "bad root node (index or primitive range out of bounds)" );
		ds.root_min[0] = root.aabbMinx, ds.root_min[1] = root.aabbMiny, ds.root_min[2] = root.aabbMinz;
		ds.root_max[0] = root.aabbMaxx, ds.root_max[1] = root.aabbMaxy, ds.root_max[2] = root.aabbMaxz;
		for (int k = 0; k < npairs; k++)
		{
			const rtmi_bvh_node& L = d->nodes[3 + 2 * k], &R = d->nodes[4 + 2 * k];
			uint32_t rl, rr;
			if (!encode( L, rl ) || !encode( R, rr )) return fail( c, RTMI_ERR_BVH, "bad node (index or primitive range out of bounds)" );
			pairs[4 * (size_t)k + 0] = make_float4( L.aabbMinx, L.aabbMiny, L.aabbMinz, L.aabbMaxx );
			float4 q1 = make_float4( L.aabbMaxy, L.aabbMaxz, 0, 0 ); memcpy( &q1.z, &rl, 4 ); memcpy( &q1.w, &L.primitiveCount, 4 );
			pairs[4 * (size_t)k + 1] = q1;
			pairs[4 * (size_t)k + 2] = make_float4( R.aabbMinx, R.aabbMiny, R.aabbMinz, R.aabbMaxx );
			float4 q3 = make_float4( R.aabbMaxy, R.aabbMaxz, 0, 0 ); memcpy( &q3.z, &rr, 4 ); memcpy( &q3.w, &R.primitiveCount, 4 );
			pairs[4 * (size_t)k + 3] = q3;
		}
		// walk the tree: every pair reachable exactly once (no cycles / sharing), depth bound for the traversal stack
		{
			const size_t allPairs = pairs.size() / 4; // the reference's pairs and the synthetic ones
			std::vector<uint8_t> seen( allPairs, 0 );
			std::vector<std::pair<uint32_t, int>> stk;
			if (!( ds.root_ref & RT_LEAF )) stk.push_back( { ds.root_ref, 1 } );
			size_t visited = 0;
			while (!stk.empty())
			{
				auto [p, dep] = stk.back();
				stk.pop_back();
				if (p >= allPairs || seen[p]) return fail( c, RTMI_ERR_BVH, "node array is not a tree" );
				seen[p] = 1, visited++;
				maxDepth = std::max( maxDepth, dep );
				uint32_t rl, rr;
				memcpy( &rl, &pairs[4 * (size_t)p + 1].z, 4 ), memcpy( &rr, &pairs[4 * (size_t)p + 3].z, 4 );
				if (!( rl & RT_LEAF )) stk.push_back( { rl, dep + 1 } );
				if (!( rr & RT_LEAF )) stk.push_back( { rr, dep + 1 } );
			}
			if (maxDepth > RTMI_MAX_BVH_DEPTH) return fail( c, RTMI_ERR_BVH, "BVH deeper than RTMI_MAX_BVH_DEPTH" );
		}
		// leaf-ordered primitive records
		std::vector<float4> prims( (size_t)nbvh * 3 );
		const int triBase = d->n_quads + d->n_spheres + d->n_cubes + d->n_planes;
		for (int k = 0; k < nbvh; k++)
		{
			const int pi = d->prim_idx[k];
			if (pi < 0 || pi >= nbvh) return fail( c, RTMI_ERR_BVH, "prim_idx out of range" );
			float4 a, b, cc = make_float4( 0, 0, 0, 0 );
			if (pi < d->n_spheres)
			{
				const float* si = d->sphere_info + 4 * (size_t)pi;
				a = make_float4( si[0], si[1], si[2], __builtin_bit_cast( float, 1 ) );
				b = make_float4( si[3], 0, 0, __builtin_bit_cast( float, d->n_quads + pi ) );
			}
			else
			{
				const int ti = pi - d->n_spheres;
				const float* v = d->tri_verts + 12 * (size_t)ti;
				// edge1 = pos2 - pos1, edge2 = pos3 - pos1 (primitive.h:55-56), same two binary32 subtractions
				a = make_float4( v[0], v[1], v[2], __builtin_bit_cast( float, 0 ) );
				b = make_float4( v[4] - v[0], v[5] - v[1], v[6] - v[2], __builtin_bit_cast( float, triBase + ti ) );
				cc = make_float4( v[8] - v[0], v[9] - v[1], v[10] - v[2], 0 );
			}
			prims[3 * (size_t)k] = a, prims[3 * (size_t)k + 1] = b, prims[3 * (size_t)k + 2] = cc;
		}
		float4* dp = 0, *dq = 0;
		if (( r = dev_upload( c, &dp, pairs.data(), pairs.size(), &allocs ) ) || ( r = dev_upload( c, &dq, prims.data(), prims.size(), &allocs ) )) return bail( r );
		ds.pairs = dp, ds.prims = dq;
		if (c->wide_on && !( ds.root_ref & RT_LEAF ))
		{
			std::vector<float4> wide, leafbox;
			const int need = build_wide_nodes( pairs, ds.root_ref, nbvh, wide, leafbox );
			if (need >= 0 && need <= 56) // (the per-lane stack lives in LDS; a tree that would need more keeps the pairs)
			{
				float4* dw = 0, *dl = 0;
				if (( r = dev_upload( c, &dw, wide.data(), wide.size(), &allocs ) ) || ( r = dev_upload( c, &dl, leafbox.data(), leafbox.size(), &allocs ) )) return bail( r );
				ds.wide = dw, ds.leafbox = dl, ds.wide_root = 0;
				wideNeed = need;
				if (c->log_bounces) fprintf( stderr, "rtmi: %zu 4-wide nodes over %d pairs, stack need %d (pairs: %d)\n", wide.size() / 4, npairs, need, maxDepth );
			}
		}
	}
	else ds.root_ref = RT_DONE;
	// analytic primitives
	{
		std::vector<float4> q( (size_t)std::max( d->n_quads, 1 ) * 4 ), cu( (size_t)std::max( d->n_cubes, 1 ) * 8 ), pl( std::max( d->n_planes, 1 ) );
		for (int i = 0; i < d->n_quads; i++)
		{
			memcpy( &q[4 * (size_t)i], d->quad_inv_T + 16 * (size_t)i, 48 );
			q[4 * (size_t)i + 3] = make_float4( d->quad_size[i], 0, 0, 0 );
		}
		for (int i = 0; i < d->n_cubes; i++)
		{
			memcpy( &cu[8 * (size_t)i], d->cube_inv_M + 16 * (size_t)i, 48 );
			memcpy( &cu[8 * (size_t)i + 3], d->cube_b + 8 * (size_t)i, 32 );
			memcpy( &cu[8 * (size_t)i + 5], d->cube_M + 16 * (size_t)i, 48 );
		}
		const int planeBase = d->n_quads + d->n_spheres + d->n_cubes;
		for (int i = 0; i < d->n_planes; i++) memcpy( &pl[i], d->primitives + 4 * (size_t)( planeBase + i ), 16 );
		float4* a = 0, *b = 0, *p = 0;
		if (( r = dev_upload( c, &a, q.data(), q.size(), &allocs ) ) || ( r = dev_upload( c, &b, cu.data(), cu.size(), &allocs ) ) || ( r = dev_upload( c, &p, pl.data(), pl.size(), &allocs ) )) return bail( r );
		ds.quads = a, ds.cubes = b, ds.planes = p;
		for (int i = 0; i < 4; i++) ds.quad0[i] = q[i];
		// the reference's room (scene.h:70-75): +x, -x, +y, -y, +z, -z.  Then one division serves each pair (isect_plane_pair)
		static const float axes[6][3] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 0, 0, -1 } };
		bool paired = c->plane_pairs_on && d->n_planes == 6;
		for (int i = 0; i < 6 && paired; i++) paired = pl[i].x == axes[i][0] && pl[i].y == axes[i][1] && pl[i].z == axes[i][2] && std::isfinite( pl[i].w );
		ds.plane_pairs = paired ? 1 : 0;
		for (int i = 0; i < 6; i++) ds.pair_d[i] = paired ? pl[i].w : 0.0f;
	}
	// shading data: per-primitive materials deduplicated into a table (the reference assigns one Material per object / OBJ file)
	{
		std::vector<DMaterial> table;
		// (the device copy of `primitives` carries the material index in its w: hit_attributes reads xyz only, and the planes' d
		// lives in ds.planes)
		std::vector<float4> prim_host( (size_t)std::max<long long>( total, 1 ), make_float4( 0, 0, 0, 0 ) );
		for (long long i = 0; i < total; i++)
		{
			DMaterial m = {};
			memcpy( &m.color, d->albedo + 4 * i, 16 ), memcpy( &m.absorption, d->absorption + 4 * i, 16 );
			m.ior = d->ior[i], m.type = d->material_type[i];
			if (m.type < 0 || m.type > 4) return bail( fail( c, RTMI_ERR_INVALID, "material_type out of range" ) );
			int found = -1;
			for (int k = (int)table.size() - 1; k >= 0; k--) if (memcmp( &table[k], &m, sizeof( DMaterial ) ) == 0) { found = k; break; }
			if (found < 0)
			{
				if (table.size() >= 255) return bail( fail( c, RTMI_ERR_INVALID, "more than 255 distinct materials" ) );
				found = (int)table.size(), table.push_back( m );
			}
			memcpy( &prim_host[i], d->primitives + 4 * i, 12 );
			prim_host[i].w = __builtin_bit_cast( float, (uint32_t)found );
		}
		if (table.empty()) table.push_back( DMaterial{} );
		float4* prim = 0; DMaterial* mt = 0; float* invr = 0; uint32_t* tex = 0;
		if (( r = dev_upload( c, &prim, prim_host.data(), prim_host.size(), &allocs ) ) ||
			( r = dev_upload( c, &mt, table.data(), table.size(), &allocs ) ) || ( r = dev_upload( c, &invr, d->sphere_invr, (size_t)d->n_spheres, &allocs ) )) return bail( r );
		if (d->texture) { if (( r = dev_upload( c, &tex, d->texture, (size_t)d->tex_w * d->tex_h, &allocs ) )) return bail( r ); }
		ds.primitives = prim, ds.materials = mt, ds.sphere_invr = invr, ds.texture = tex, ds.tex_w = d->tex_w, ds.tex_h = d->tex_h;
	}
	memcpy( &ds.light_c1, d->light_corners, 16 ), memcpy( &ds.light_c2, d->light_corners + 4, 16 ), memcpy( &ds.light_c3, d->light_corners + 8, 16 );
	memcpy( &ds.light_emission, d->light_emission, 16 );
	{
		// Scene::RandomPointOnLight normalises the two edge vectors of the light for every sample (scene.h:319, 326); they do
		// not depend on the sample.  Same binary32 operations in the same order as normalize3 in rt_device.h (this file is
		// compiled with -ffp-contract=off for the host too; x86-64 sqrtss / divss round correctly like the device's)
		auto unit = []( const float4& a, const float4& b ) {
			const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
			const float inv = 1.0f / sqrtf( x * x + y * y + z * z );
			return make_float4( x * inv, y * inv, z * inv, 0 ); };
		ds.light_u = unit( ds.light_c1, ds.light_c2 ), ds.light_w = unit( ds.light_c3, ds.light_c2 );
	}
	if (d->n_quads > 0) memcpy( &ds.light_N, d->primitives, 16 ); // quads[0].N, scene.h:120 / 323
	ds.light_A = d->light_A, ds.light_s = d->light_s;
	memcpy( ds.wl_pos, d->whitted_light_pos, 32 ), memcpy( ds.wl_color, d->whitted_light_color, 32 );
	ds.wl_intensity = d->whitted_intensity;

	free_scene( c );
	free_batches( c ); // the suspend records are sized by the stack depth
	if (c->child) free_batches( c->child );
	c->scene_allocs = allocs, c->ds = ds, c->has_scene = true;
	c->stack_depth = std::max( 4, maxDepth + 1 );
	c->walk_depth = std::max( c->stack_depth, wideNeed + 1 );
	set_max_blocks( c );
	return RTMI_OK;
}

int rtmi_set_camera( rtmi_ctx* c, const float cam[16] )
{
	if (!c || !cam) return RTMI_ERR_INVALID;
	memcpy( &c->cam, cam, 64 );
	return RTMI_OK;
}
int rtmi_set_seeds( rtmi_ctx* c, const uint32_t* seeds )
{
	if (!c || !seeds) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	HIPCHK( hipMemcpyAsync( c->seeds, seeds, (size_t)c->W * c->H * 4, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	return RTMI_OK;
}
int rtmi_reset_accumulator( rtmi_ctx* c )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	// The accumulates in flight come first (they write the accumulator).  The paths that wait in the carry buffers and slot lists,
	// and the late sums, belong to the image that is being thrown away: they are dropped, not finished -- a camera that moves
	// every frame (renderer.cpp:622-633) would otherwise pay for the longest paths of a frame nobody will see.
	if (c->chain->last_acc) HIPCHK( hipStreamWaitEvent( c->stream, c->chain->last_acc, 0 ) );
	rtmi_ctx* both[2] = { c, c->child };
	for (rtmi_ctx* l : both)
	{
		if (!l) continue;
		if (l->carry_ctr) HIPCHK( hipMemsetAsync( &l->carry_ctr->n_state[0], 0, sizeof( uint32_t ), c->stream ) );
		for (auto& b : l->slots) if (b.n_left) HIPCHK( hipMemsetAsync( b.n_left, 0, sizeof( uint32_t ), c->stream ) );
		l->paths_carried = false, l->carry_pending = false;
	}
	const size_t n = (size_t)c->W * c->H;
	HIPCHK( hipMemsetAsync( c->late, 0, 4 * n * sizeof( long long ), c->stream ) );
	HIPCHK( hipMemsetAsync( c->odd, 0, n * sizeof( float4 ), c->stream ) );
	HIPCHK( hipMemsetAsync( c->accum, 0, n * sizeof( float4 ), c->stream ) );
	return RTMI_OK;
}
int rtmi_set_accumulator( rtmi_ctx* c, void* p )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c ); if (r) return r; } // what is in flight belongs to the current accumulator
	{ int r = sync_all( c ); if (r) return r; }
	c->accum = p ? (float4*)p : c->accum_own;
	return RTMI_OK;
}

static size_t stack_bytes( rtmi_ctx* c ) { return (size_t)c->stack_depth * BLOCK * sizeof( uint32_t ); }
// device scratch of the parity hooks and rtmi_postprocess: kept in the handle, grown when a call needs more
static int hook_reserve( rtmi_ctx* c, size_t bytes )
{
	if (c->hook_cap >= bytes) return RTMI_OK;
	HIPCHK( hipStreamSynchronize( c->stream ) );
	if (c->hook_buf) (void)hipFree( c->hook_buf ), c->hook_buf = 0, c->hook_cap = 0;
	const size_t want = ( bytes + ( 1u << 20 ) - 1 ) & ~(size_t)( ( 1u << 20 ) - 1 );
	hipError_t e = hipMalloc( &c->hook_buf, want );
	if (e != hipSuccess) { c->hook_buf = 0; c->err = std::string( "hipMalloc: " ) + hipGetErrorString( e ); return RTMI_ERR_NOMEM; }
	c->hook_cap = want;
	return RTMI_OK;
}

static inline int nsblocks( rtmi_ctx* c, size_t n ) { return (int)( ( n + c->sblock - 1 ) / c->sblock ); }
// persistent launches: as many blocks as stay resident (LDS stack bound), never more than the rays need.
// the walks: fewer resident blocks per CU once a bounce is small (`upper` = the bounce's path count bound; the rays that
// walk are a sixth to a tenth of it).  A small launch is a chain of dependent fetches per wave, and each wave's chain runs
// faster with fewer waves beside it on its SIMD; a large one needs all the lanes it can get.  Measured per launch on the
// bench scene (sequential streams, us, 4 / 3 / 2 blocks per CU): closest-hit walk of bounce 2: 1683 / 1892 / 2539, of bounce
// 16: 268 / 246 / 256; shadow walk of bounce 8: 340 / 312 / 314, of bounce 16: 226 / 203 / 163.  With the shadow walk on the
// side stream (render_batch) two small walks also fit a CU together.  Thresholds in paths per resident lane, set in rtmi_create.
static inline dim3 wgrid( rtmi_ctx* c, bool any, uint32_t upper )
{
	const uint32_t* lim = any ? c->grid_any : c->grid_closest; // path counts below which a launch gets 3, then 2 blocks per CU
	const int per_cu = upper < lim[1] ? 2 : upper < lim[0] ? 3 : 4;
	const int blocks = std::min( c->max_blocks, c->n_cus * per_cu );
	return dim3( (unsigned)std::max( 1, std::min( nblocks( upper ), blocks ) ) );
}

static void launch_trace( rtmi_ctx* c, BatchSlot& b, bool any, int set, uint32_t upper, int budget, hipStream_t on = 0 )
{
	const size_t lds = (size_t)c->walk_depth * BLOCK * sizeof( uint32_t );
	hipStream_t s = on ? on : c->stream;
	DevCounters* ctr = b.ctr;
	TraceArgs a = {};
	a.st = b.st[set], a.sq = b.sq[set], a.sq_out = b.sq[set ^ 1], a.sinks = make_sinks( c, b ), a.wq = b.wq[set];
	a.count_ptr = any ? &ctr->n_shadow[set] : &ctr->n_walk[set], a.rays_ptr = &ctr->n_state[set];
	a.pool = any ? &ctr->pool_connect : &ctr->pool_extend;
	a.rec_in = any ? b.srec[set ^ 1] : b.rec[set ^ 1], a.rec_out = any ? b.srec[set] : b.rec[set];
	a.n_rec_out = any ? &ctr->n_srec[set] : &ctr->n_rec[set], a.n_rec_in = any ? &ctr->n_srec[set ^ 1] : &ctr->n_rec[set ^ 1];
	a.n_shadow_out = &ctr->n_shadow[set ^ 1];
	a.rec_cap = b.rec_cap, a.rec_stride = b.rec_stride, a.refill = c->refill, a.budget = budget;
	ScopedTimer t( c, any ? 2 : 0, s );
	if (any)
	{
		if (c->count_nodes) hipLaunchKernelGGL( ( k_trace<true, true> ), wgrid( c, true, upper ), dim3( BLOCK ), lds, s, c->ds, a, ctr );
		else hipLaunchKernelGGL( ( k_trace<true, false> ), wgrid( c, true, upper ), dim3( BLOCK ), lds, s, c->ds, a, ctr );
	}
	else
	{
		if (c->count_nodes) hipLaunchKernelGGL( ( k_trace<false, true> ), wgrid( c, false, upper ), dim3( BLOCK ), lds, s, c->ds, a, ctr );
		else hipLaunchKernelGGL( ( k_trace<false, false> ), wgrid( c, false, upper ), dim3( BLOCK ), lds, s, c->ds, a, ctr );
	}
}
static void launch_primary( rtmi_ctx* c, BatchSlot& b, uint32_t N, int sample_begin, int stride )
{
	ScopedTimer t( c, 1, c->stream );
	hipLaunchKernelGGL( k_primary, dim3( nsblocks( c, N ) ), dim3( c->sblock ), 0, c->stream, c->ds, b.st[0], b.E, (float4*)b.Es, N, c->cam, c->W, c->H, 1.0f / c->W, 1.0f / c->H, // camera.h:31-32
		c->seeds, sample_begin, stride, b.wq[0], b.ctr );
}

// One batch of S samples.  On the handle's stream: ray generation over all S*W*H slots, then queue-driven bounces until
// few paths are left.  On the slot's stream, after the bulk: k_tail runs the remaining paths to their ends, then
// k_accumulate (after the previous batch's).  The call returns with all of it enqueued.
static int render_batch( rtmi_ctx* c, BatchSlot& b, int sample_begin, int S, int stride, uint64_t ticket )
{
	Turn turn( c->chain, ticket ); // passes the turn on when this function returns, wherever it returns
	const int WH = c->W * c->H;
	const uint32_t N = (uint32_t)( (size_t)S * WH );
	hipStream_t s = c->stream;
	DevCounters* ctr = b.ctr;
	if (b.busy) { HIPCHK( hipStreamWaitEvent( s, b.acc_done, 0 ) ); b.busy = false; } // the slot's previous batch is done with these buffers
	HIPCHK( hipMemsetAsync( ctr, 0, 64, s ) ); // queue lengths and pools
	// `last`: nothing is handed on.  The batch of the carried paths alone (join_batches, S = 0) -- and a SMALL batch: handing late
	// paths on keeps the bounces of a 33 M-path batch large; a batch that starts below the hand-over size (one sample of a 720p
	// frame: Renderer::Tick) would only move its paths from one small launch to another, and when the caller reads right after it
	// (Tick does) they would get a second batch of their own.  It finishes its paths itself: wavefront bounces down to
	// small_paths, then the tail rounds, which run for latency if the host is already waiting when they start (k_tail_decide).
	const bool last = S == 0 || ( N <= c->carry_min && !( c->preview_reads && c->carry_on ) );
	if (S != 0) launch_primary( c, b, N, sample_begin, stride );
	c->stats.primary_samples += N;
	uint32_t upper = N; // upper bound of the current path queue length (lengths never grow)
	// veterans (k_carry_out): the slots of the last sample plane that fits a batch; needs two planes and the carry
	const uint32_t planes = (uint32_t)( ( 1ull << RTMI_SLOT_BITS ) / (unsigned)WH );
	const uint32_t veteran_from = c->carry_on && c->veteran_rounds >= 0 && planes >= 2 ? ( planes - 1 ) * (uint32_t)WH : 0xFFFFFFFFu;
	if (c->carry_on && c->paths_carried)
	{
		// the late paths that earlier batches handed on bounce along with this batch's (entries N, N+1, ...): the carry buffer
		// (the previous batch's), and what outlived the first tail round of the batch that used this slot before -- of all
		// slots when this is the last batch
		auto take = [&]( PathState src, uint32_t* n_src, uint32_t cap ) {
			hipLaunchKernelGGL( k_carry_in, dim3( ( cap + 511 ) / 512 ), dim3( 512 ), 0, s, c->ds, src, n_src, cap, b.st[0], &ctr->n_state[0], b.wq[0], ctr, (uint32_t)WH, veteran_from );
			hipLaunchKernelGGL( k_carry_commit, dim3( 1 ), dim3( 1 ), 0, s, &ctr->n_state[0], n_src, cap );
			upper += cap; };
		take( c->carry[0], &c->carry_ctr->n_state[0], c->carry_cap );
		for (int k = 0; k < c->n_slots; k++)
		{
			BatchSlot& o = c->slots[k];
			if (&o != &b && S != 0) continue;
			if (&o != &b && o.busy) { HIPCHK( hipStreamWaitEvent( s, o.acc_done, 0 ) ); o.busy = false; } // its tail stream has written its list
			take( o.left, o.n_left, c->left_cap );
		}
	}
	int cur = 0;        // state / shadow queue set of this bounce
	int clean = 0;      // consecutive bounces launched without parking
	bool tail = false, side_busy = false;
	hipEvent_t window = 0; // (profiling) open walk window: from the end of a k_shade to the point where both walks after it are done
	auto close_window = [&]() { if (window) { hipEvent_t e = get_event( c ); if (e) { (void)hipEventRecord( e, s ); c->timed.push_back( { window, e, 4 } ); } else c->ev_pool.push_back( window ); window = 0; } };
	const int min_bounce = c->carry_on && !last ? c->carry_bounce : 0; // with carry, a batch that is still large stays in the loop until its paths are late
	const uint32_t tail_paths = last ? c->small_paths : c->tail_paths;
	// a batch leaves the loop early once this few paths are left, whatever their age.  A small batch that hands its paths on
	// (preview_reads) must not before they are late: what is not late goes through this batch's own tail, whose list holds left_cap
	const uint32_t small = last || N > c->carry_min ? c->small_paths : std::min( c->small_paths, c->left_cap );
	for (int bounce = 0;; bounce++)
	{
		// Near the end of the bulk: stop parking rays (so that no suspended state is left), one bounce later hand the
		// remaining paths on: the late ones to the next batch (k_carry_out), the others to this batch's k_tail.
		// (a launch without parking lasts as long as its longest ray, 3x the parked one at this size: only the bounce before
		// the hand-over runs that way; `upper` is two bounces old and the path count falls by about 0.83 per bounce)
		const unsigned long long soon = (unsigned long long)( upper * 0.85 );
		const bool prep = tail_paths && ( ( soon <= tail_paths && bounce + 1 >= min_bounce ) || soon <= small );
		if (tail_paths && clean >= 1 && ( ( upper <= tail_paths && bounce >= min_bounce ) || upper <= small )) { tail = true; break; }
		const int budget = prep ? 0 : c->budget;
		clean = prep ? clean + 1 : 0;
		if (bounce == 0 && c->overlap_on && c->prof_events) { window = get_event( c ); if (window) (void)hipEventRecord( window, s ); } // the first walk has a window to itself
		launch_trace( c, b, false, cur, upper, budget );
		if (side_busy) { HIPCHK( hipStreamWaitEvent( s, b.ev_any, 0 ) ); side_busy = false; } // the previous bounce's shadow walk fills the queue k_shade appends to
		close_window();
		{
			ScopedTimer t( c, 1, c->stream );
			hipLaunchKernelGGL( k_shade, dim3( nsblocks( c, upper ) ), dim3( c->sblock ), 0, s, c->ds, b.st[cur], b.st[cur ^ 1], b.sq[cur], make_sinks( c, b ), &ctr->n_state[cur], &ctr->n_state[cur ^ 1],
				b.wq[cur ^ 1], &ctr->n_walk[cur ^ 1], &ctr->n_shadow[cur], ctr );
		}
		// what the next bounce starts with (paths + carried shadow rays) goes to pinned host memory; it is looked at two
		// iterations later
		const int slot = bounce & 3;
		if (c->overlap_on)
		{
			// this bounce's shadow walk and the next bounce's closest-hit walk need nothing from each other: the shadow walk goes
			// to the side stream, and as its waves run dry the blocks of the closest-hit walk take their place
			HIPCHK( hipEventRecord( b.ev_shade, s ) );
			HIPCHK( hipStreamWaitEvent( c->side, b.ev_shade, 0 ) );
			launch_trace( c, b, true, cur, upper, budget, c->side );
			hipLaunchKernelGGL( k_next_bounce_post, dim3( 1 ), dim3( 1 ), 0, c->side, ctr, cur, b.host_counts_dev + slot );
			HIPCHK( hipEventRecord( b.count_ev[slot], c->side ) );
			HIPCHK( hipEventRecord( b.ev_any, c->side ) );
			side_busy = true;
			hipLaunchKernelGGL( k_next_bounce_pre, dim3( 1 ), dim3( 1 ), 0, s, ctr, cur );
			if (c->prof_events) { window = get_event( c ); if (window) (void)hipEventRecord( window, s ); }
		}
		else
		{
			launch_trace( c, b, true, cur, upper, budget ); // this bounce's shadow rays (at most `upper` + the carried ones)
			hipLaunchKernelGGL( k_next_bounce, dim3( 1 ), dim3( 1 ), 0, s, ctr, cur, b.host_counts_dev + slot );
			HIPCHK( hipEventRecord( b.count_ev[slot], s ) );
		}
		c->stats.bounce_launches++;
		cur ^= 1;
		HIPCHK( hipGetLastError() ); // a launch that was refused (bad configuration) is reported for the bounce it belongs to
		if (c->check_launches) // diagnostics: a fault inside a kernel surfaces here, at its bounce, instead of at the next read
		{
			HIPCHK( hipStreamSynchronize( s ) );
			if (c->side) HIPCHK( hipStreamSynchronize( c->side ) );
		}
		// the path count comes back two bounces late, so that the host never waits for the bounce it has just enqueued; a small batch
		// with nothing beside it (a frame of the interactive loop) reads it one bounce late: its bounces are launch-latency bound,
		// and every one it runs beyond the switch to the tail is 0.3 ms of the frame
		const int lag = last && S != 0 ? c->count_lag : 2;
		if (bounce >= lag)
		{
			const int old = ( bounce - lag ) & 3;
			HIPCHK( hipEventSynchronize( b.count_ev[old] ) );
			upper = std::min( upper, b.host_counts[old] );
			if (c->log_bounces && ( bounce < 64 || ( bounce & ( bounce - 1 ) ) == 0 )) fprintf( stderr, "rtmi: bounce %d starts with %u paths\n", bounce + 1 - lag, b.host_counts[old] );
		}
		if (upper == 0) break;
	}
	if (side_busy) { HIPCHK( hipStreamWaitEvent( s, b.ev_any, 0 ) ); side_busy = false; }
	close_window();
	if (tail && c->carry_on && !last)
	{
		c->paths_carried = true;
		// late paths -> carry buffer (they join the next batch, or join_batches finishes them); the others -> the other state set
		hipLaunchKernelGGL( k_carry_out, dim3( ( upper + 511 ) / 512 ), dim3( 512 ), 0, s, b.st[cur], &ctr->n_state[cur], b.st[cur ^ 1], &ctr->n_state[cur ^ 1],
			c->carry[0], &c->carry_ctr->n_state[0], c->carry_cap, veteran_from );
		if (c->log_bounces)
		{
			uint32_t rest = 0, carried = 0;
			(void)hipStreamSynchronize( s );
			(void)hipMemcpy( &rest, &ctr->n_state[cur ^ 1], 4, hipMemcpyDeviceToHost ), (void)hipMemcpy( &carried, &c->carry_ctr->n_state[0], 4, hipMemcpyDeviceToHost );
			fprintf( stderr, "rtmi: hand-over after %d bounces: %u paths carried on (buffer %u), %u stay for this batch's tail\n", (int)c->stats.bounce_launches, carried, c->carry_cap, rest );
		}
		hipLaunchKernelGGL( k_next_bounce, dim3( 1 ), dim3( 1 ), 0, s, ctr, cur, (uint32_t*)0 );
		cur ^= 1;
	}
	c->carry_pending = true; // also without carry: late sums wait to be folded
	// the rest of the batch runs beside the next batches' bulk
	hipStream_t ts = b.tail_stream;
	HIPCHK( hipEventRecord( b.bulk_done, s ) );
	HIPCHK( hipStreamWaitEvent( ts, b.bulk_done, 0 ) );
	if (tail)
	{
		ScopedTimer t( c, 3, ts );
		if (c->carry_on && !last)
		{
			// what stays with the batch are the paths that were not late yet (parked walks make a path lag): after the 32 bounces
			// of the first round the survivors are, and wait in the slot's list for the next batch that uses the slot; only
			// what does not fit goes through the remaining rounds
			// (with veterans: `veteran_rounds` more rounds -- 64, 128, ... bounces -- before the survivors go to the list)
			const int first_rounds = veteran_from != 0xFFFFFFFFu ? std::min( c->veteran_rounds, 10 ) : 0;
			cur = launch_tail_rounds( c, ts, b.st, cur, ctr, make_sinks( c, b ), upper, 0, first_rounds, false, first_rounds == 0 && c->stop_late ? 1 : 0 );
			hipLaunchKernelGGL( k_carry_out, dim3( ( upper + 511 ) / 512 ), dim3( 512 ), 0, ts, b.st[cur], &ctr->n_state[cur], b.st[cur ^ 1], &ctr->n_state[cur ^ 1], b.left, b.n_left, c->left_cap, 0xFFFFFFFFu );
			hipLaunchKernelGGL( k_next_bounce, dim3( 1 ), dim3( 1 ), 0, ts, ctr, cur, (uint32_t*)0 );
			cur ^= 1;
			launch_tail_rounds( c, ts, b.st, cur, ctr, make_sinks( c, b ), 4096, first_rounds + 1, 11 ); // normally nothing: a small persistent grid
		}
		else launch_tail_rounds( c, ts, b.st, cur, ctr, make_sinks( c, b ), upper, 0, 11, last );
		c->stats.bounce_launches++;
	}
	turn.enter(); // accumulate in batch order: the previous batch (of either lane) has recorded its event
	if (c->chain->last_acc) HIPCHK( hipStreamWaitEvent( ts, c->chain->last_acc, 0 ) );
	if (S != 0)
	{
		ScopedTimer t( c, 3, ts );
		hipLaunchKernelGGL( k_accumulate, dim3( nblocks( WH ) ), dim3( BLOCK ), 0, ts, c->accum, b.E, b.Es, WH, S );
	}
	HIPCHK( hipEventRecord( b.acc_done, ts ) );
	b.busy = true, c->chain->last_acc = b.acc_done;
	turn.leave();
	HIPCHK( hipGetLastError() );
	return RTMI_OK;
}

static int fetch_counters( rtmi_ctx* c )
{
	int r = c->is_child ? RTMI_OK : sync_all( c ); // (the parent's sync_all covers the child)
	if (r) return r;
	if (c->child)
	{
		if (( r = fetch_counters( c->child ) )) { c->err = c->child->err; return r; }
		resolve_timers( c->child );
		merge_stats( c->stats, c->child->stats );
		c->ext_intervals.insert( c->ext_intervals.end(), c->child->ext_intervals.begin(), c->child->ext_intervals.end() );
		c->child->ext_intervals.clear();
		c->walk_intervals.insert( c->walk_intervals.end(), c->child->walk_intervals.begin(), c->child->walk_intervals.end() );
		c->child->walk_intervals.clear();
	}
	for (int k = 0; k <= MAX_SLOTS; k++)
	{
		DevCounters h;
		DevCounters* src = k < MAX_SLOTS ? c->slots[k].ctr : c->carry_ctr; // every slot that was ever set up (option slots may have shrunk)
		if (!src) continue;
		HIPCHK( hipMemcpy( &h, src, sizeof( h ), hipMemcpyDeviceToHost ) );
		c->stats.extend_rays += h.extend_rays, c->stats.shadow_rays += h.shadow_rays;
		c->stats.extend_pairs += h.extend_pairs, c->stats.extend_prims += h.extend_prims;
		c->stats.shadow_pairs += h.shadow_pairs, c->stats.shadow_prims += h.shadow_prims;
		c->stats.extend_walkers += h.extend_walkers, c->stats.shadow_walkers += h.shadow_walkers, c->stats.suspended += h.suspended;
		c->stats.max_extend_steps = std::max<uint64_t>( c->stats.max_extend_steps, h.max_steps[0] ), c->stats.max_shadow_steps = std::max<uint64_t>( c->stats.max_shadow_steps, h.max_steps[1] );
		for (int i = 0; i < 12; i++) c->stats.tail_profile[i] += h.tail_prof[i];
		// the ray counters only: the queue lengths of the carry buffer stay
		HIPCHK( hipMemset( (char*)src + offsetof( DevCounters, extend_rays ), 0, sizeof( DevCounters ) - offsetof( DevCounters, extend_rays ) ) );
	}
	return RTMI_OK;
}

// Device memory ran out while the batch slots were being resized (rtmi_render): finish every path that is in flight or waiting
// in a carry buffer with the queues that are left, then free all of them.  A slot whose resize failed holds nothing, and the
// camera-less batch that finishes the carried paths (join_batches) needs one: the other slots go first, then it gets the smallest.
static int drain_and_free( rtmi_ctx* c )
{
	(void)hipGetLastError(); // (the refused allocation is dealt with here, not by the next launch check)
	int r = sync_all( c );
	if (r) return r;
	rtmi_ctx* both[2] = { c, c->child };
	for (rtmi_ctx* l : both)
	{
		if (!l) continue;
		for (int i = 0; i < MAX_SLOTS; i++) if (i != l->next_slot) free_batch( l->slots[i] );
		if (l->carry_on && l->paths_carried && !l->slots[l->next_slot].E && ( r = ensure_batch( l, l->slots[l->next_slot], 1 ) )) { if (l != c) c->err = l->err; return r; }
	}
	if (( r = join_batches( c ) ) || ( r = sync_all( c ) )) return r;
	free_batches( c );
	if (c->child) free_batches( c->child );
	return RTMI_OK;
}

int rtmi_render( rtmi_ctx* c, int mode, int sample_begin, int sample_count, int sample_stride, int max_depth, int anti_aliasing )
{
	if (!c) return RTMI_ERR_INVALID;
	if (!c->has_scene) return fail( c, RTMI_ERR_NO_SCENE, "rtmi_render before rtmi_upload_scene" );
	HIPCHK( hipSetDevice( c->device ) );
	const int WH = c->W * c->H;
	if (mode == RTMI_PATHTRACING)
	{
		if (sample_count <= 0 || sample_stride <= 0 || sample_begin < 0) return fail( c, RTMI_ERR_INVALID, "bad sample range" );
		// samples in flight per batch: enough paths to fill the chip in the long bounce tail, bounded memory
		int S = (int)std::max<long long>( 1, std::min<long long>( sample_count, ( 1ll << RTMI_SLOT_BITS ) / WH ) );
		if (c->batch_opt > 0) S = (int)std::max<long long>( 1, std::min<long long>( std::min( sample_count, c->batch_opt ), ( 1ll << RTMI_SLOT_BITS ) / WH ) );
		// Size every slot now: an allocation in the middle of a pipelined sequence of batches would drain it.  The queues of a
		// full-size batch are 19 GB per slot; on a GPU that has less to give (other users, other handles) first the second lane
		// goes, then the batches get smaller until they fit -- the image does not depend on either, only the speed does.
		if (c->fit_samples > 0) S = std::min( S, c->fit_samples );
		int n_batches = 0;
		rtmi_ctx* k = 0;
		for (;;)
		{
			int r = RTMI_OK;
			for (int i = 0; i < c->n_slots && !r; i++) r = ensure_batch( c, c->slots[i], S );
			if (r)
			{
				if (r != RTMI_ERR_NOMEM || S == 1) return r;
				const std::string why = c->err;
				if (( r = drain_and_free( c ) )) return r;
				S = ( S + 1 ) / 2, c->fit_samples = S;
				fprintf( stderr, "rtmi_render: the queues of %d-sample batches do not fit in device memory (%s): trying batches of %d\n", 2 * S, why.c_str(), S );
				c->err.clear();
				continue;
			}
			// more bulk work is coming: the tails in flight run for throughput again.  Not so for a call of one small batch (a frame of
			// the interactive loop): there is no bulk for its tail to run beside, and the read that follows finds the host still
			// enqueuing when the first tail rounds decide (k_tail_decide) -- they run for latency from the start
			*(volatile uint32_t*)c->draining = sample_count <= S && (unsigned long long)S * WH <= c->carry_min ? 1u : 0u;
			n_batches = ( sample_count + S - 1 ) / S;
			// the second lane: made when first needed (more than one batch of more than carry_min paths), brought in line every time
			if (c->lanes >= 2 && !c->child && n_batches >= 2 && (unsigned long long)S * WH > c->carry_min) { if (( r = set_lanes( c ) )) return r; }
			k = c->child && ( n_batches >= 2 || ( c->batch_seq & 1 ) ) && (unsigned long long)S * WH > c->carry_min ? c->child : 0;
			if (!k) break;
			if (( r = sync_child( c ) )) { c->err = k->err.empty() ? c->err : k->err; return r; }
			*(volatile uint32_t*)k->draining = 0;
			for (int i = 0; i < k->n_slots && !r; i++) r = ensure_batch( k, k->slots[i], S );
			if (!r) break;
			// no memory for a second set of queues: finish what is in flight, give the lane up and go on with one
			fprintf( stderr, "rtmi_render: the second lane's queues do not fit in device memory (%s): rendering on one lane\n", k->err.c_str() );
			if (( r = drain_and_free( c ) )) return r;
			c->lanes = 1, k = 0;
			(void)set_lanes( c );
		}
		if (k)
		{
			// the child's stream follows what the caller has enqueued on the handle's stream so far
			HIPCHK( hipEventRecord( c->ev_fork, c->stream ) );
			HIPCHK( hipStreamWaitEvent( k->stream, c->ev_fork, 0 ) );
		}
		uint64_t t0;
		{ std::lock_guard<std::mutex> lock( c->chain->m ); t0 = c->chain->next_ticket; c->chain->next_ticket += (uint64_t)n_batches; }
		// batch j of this call: samples [j*S, ...), ticket t0 + j (the order of the accumulates); lane = parity of the running
		// batch number, so that consecutive calls of one batch each alternate too
		auto run_lane = [&]( rtmi_ctx* l, int parity ) -> int {
			(void)hipSetDevice( l->device );
			for (int j = 0; j < n_batches; j++)
			{
				const bool mine = !k || (int)( ( c->batch_seq + (uint64_t)j ) & 1 ) == parity;
				if (!mine) continue;
				const int done = j * S, n = std::min( S, sample_count - done );
				BatchSlot& b = l->slots[l->next_slot]; // batches rotate over the slots so that one's tail overlaps the next one's bulk
				l->next_slot = ( l->next_slot + 1 ) % l->n_slots;
				int r = render_batch( l, b, sample_begin + done * sample_stride, n, sample_stride, t0 + (uint64_t)j );
				if (r)
				{
					// the batches of this lane that will not be rendered still have to pass their turns on
					for (int jj = j + 1; jj < n_batches; jj++) if (!k || (int)( ( c->batch_seq + (uint64_t)jj ) & 1 ) == parity) { Turn skip( c->chain, t0 + (uint64_t)jj ); }
					return r;
				}
			}
			return RTMI_OK; };
		int r = RTMI_OK, rk = RTMI_OK;
		if (k)
		{
			std::thread th;
			bool threaded = false;
			try { th = std::thread( [&] { rk = run_lane( k, 1 ); } ); threaded = true; } catch (...) {}
			if (threaded) { r = run_lane( c, 0 ); th.join(); }
			else
			{
				// no thread to be had: both lanes from this one, batch by batch in ticket order (a batch takes its turn before it returns)
				for (int j = 0; j < n_batches && !r && !rk; j++)
				{
					const bool childs = ( ( c->batch_seq + (uint64_t)j ) & 1 ) != 0;
					rtmi_ctx* l = childs ? k : c;
					(void)hipSetDevice( l->device );
					const int done = j * S, n = std::min( S, sample_count - done );
					BatchSlot& b = l->slots[l->next_slot];
					l->next_slot = ( l->next_slot + 1 ) % l->n_slots;
					int rr = render_batch( l, b, sample_begin + done * sample_stride, n, sample_stride, t0 + (uint64_t)j );
					if (rr) { ( childs ? rk : r ) = rr; for (int jj = j + 1; jj < n_batches; jj++) { Turn skip( c->chain, t0 + (uint64_t)jj ); } }
				}
			}
			if (rk && !r) c->err = k->err, r = rk;
		}
		else r = run_lane( c, 0 );
		c->batch_seq += (uint64_t)n_batches;
		// counters are folded into the stats lazily (rtmi_get_stats), no sync here
		return r;
	}
	if (mode < 0 || mode > RTMI_PATHTRACING) return fail( c, RTMI_ERR_INVALID, "unknown mode" );
	if (max_depth < 1 || max_depth > 64) return fail( c, RTMI_ERR_INVALID, "max_depth must be 1..64" );
	{ int r = join_batches( c ); if (r) return r; } // this mode overwrites the accumulator
	if (c->frames_depth < max_depth)
	{
		if (c->frames) (void)hipFree( c->frames ), c->frames = 0;
		int r = dev_alloc( c, &c->frames, (size_t)WH * ( max_depth + 1 ) );
		if (r) return r;
		c->frames_depth = max_depth;
	}
	{
		ScopedTimer t( c, 3, c->stream );
		hipLaunchKernelGGL( k_whitted, dim3( nblocks( WH ) ), dim3( BLOCK ), stack_bytes( c ), c->stream, c->ds, c->cam, c->W, c->H, 1.0f / c->W, 1.0f / c->H, mode, max_depth,
			anti_aliasing, c->frames, c->accum );
	}
	HIPCHK( hipGetLastError() );
	return RTMI_OK;
}

int rtmi_trace_primary( rtmi_ctx* c, int32_t* obj_out, float* t_out )
{
	if (!c || !obj_out || !t_out) return RTMI_ERR_INVALID;
	if (!c->has_scene) return fail( c, RTMI_ERR_NO_SCENE, "rtmi_trace_primary before rtmi_upload_scene" );
	HIPCHK( hipSetDevice( c->device ) );
	BatchSlot& b = c->slots[0];
	int r = ensure_batch( c, b, 1 );
	if (r) return r;
	if (b.busy) { HIPCHK( hipStreamWaitEvent( c->stream, b.acc_done, 0 ) ); b.busy = false; }
	const int WH = c->W * c->H;
	if (( r = hook_reserve( c, (size_t)WH * 8 ) )) return r;
	int* dobj = (int*)c->hook_buf; float* dt = (float*)c->hook_buf + WH;
	hipError_t e0 = hipMemsetAsync( b.ctr, 0, 64, c->stream );
	launch_primary( c, b, (uint32_t)WH, 0, 1 );
	launch_trace( c, b, false, 0, (uint32_t)WH, 0 ); // no step budget: every ray finishes in this launch
	hipLaunchKernelGGL( k_copy_hits, dim3( nblocks( WH ) ), dim3( BLOCK ), 0, c->stream, b.st[0], WH, dobj, dt );
	hipError_t e1 = hipMemcpyAsync( obj_out, dobj, (size_t)WH * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e2 = hipMemcpyAsync( t_out, dt, (size_t)WH * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e3 = hipStreamSynchronize( c->stream );
	HIPCHK( e0 ); HIPCHK( e1 ); HIPCHK( e2 ); HIPCHK( e3 );
	HIPCHK( hipGetLastError() );
	return RTMI_OK;
}

// the kernels of rtmi_trace_rays over rays that are already on the device (shared with rtmi_trace_rays_noalloc)
static int launch_trace_rays( rtmi_ctx* c, int n, const float* dO, const float* dD, const float* dT, int any_hit, int* dobj, float* dt )
{
	if (c->hook_kernel == 0)
	{
		hipLaunchKernelGGL( k_trace_rays, dim3( nblocks( n ) ), dim3( BLOCK ), stack_bytes( c ), c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
		HIPCHK( hipGetLastError() );
		return RTMI_OK;
	}
	if (c->hook_kernel == 2) // the end-of-render walker: one ray per wave (k_tail_wave's device code)
	{
		const int blocks = std::max( 1, std::min( ( n + 3 ) / 4, c->n_cus * 4 ) );
		hipLaunchKernelGGL( k_rays_wave, dim3( blocks ), dim3( 256 ), 0, c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
		HIPCHK( hipGetLastError() );
		return RTMI_OK;
	}
	const size_t WH = (size_t)c->W * c->H;
	if ((size_t)n > ( 1u << 25 )) return fail( c, RTMI_ERR_INVALID, "hook_kernel = 1: at most 2^25 rays per call" );
	BatchSlot& b = c->slots[0];
	int r = ensure_batch( c, b, (int)( ( (size_t)n + WH - 1 ) / WH ) );
	if (r) return r;
	if (b.busy) { HIPCHK( hipStreamWaitEvent( c->stream, b.acc_done, 0 ) ); b.busy = false; }
	HIPCHK( hipMemsetAsync( b.ctr, 0, 64, c->stream ) );
	hipLaunchKernelGGL( k_rays_begin, dim3( nsblocks( c, (size_t)n ) ), dim3( c->sblock ), 0, c->stream, c->ds, n, dO, dD, dT, any_hit, b.st[0], b.sq[0], (float4*)b.Es, b.wq[0], dobj, dt, b.ctr );
	launch_trace( c, b, any_hit != 0, 0, (uint32_t)n, 0 ); // no step budget: every ray finishes in this launch
	if (any_hit) hipLaunchKernelGGL( k_rays_any_result, dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, b.Es, n, dobj );
	else hipLaunchKernelGGL( k_copy_hits, dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, b.st[0], n, dobj, dt );
	HIPCHK( hipGetLastError() );
	return RTMI_OK;
}

int rtmi_trace_rays( rtmi_ctx* c, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int32_t* obj_out, float* t_out )
{
	if (!c || n < 0 || ( n && ( !O3 || !D3 || !tmax || !obj_out || !t_out ) )) return RTMI_ERR_INVALID;
	if (!c->has_scene) return fail( c, RTMI_ERR_NO_SCENE, "rtmi_trace_rays before rtmi_upload_scene" );
	if (n == 0) return RTMI_OK;
	HIPCHK( hipSetDevice( c->device ) );
	int r = hook_reserve( c, (size_t)n * 9 * 4 );
	if (r) return r;
	float* dO = (float*)c->hook_buf, *dD = dO + 3 * (size_t)n, *dT = dD + 3 * (size_t)n, *dt = dT + n;
	int* dobj = (int*)( dt + n );
	HIPCHK( hipMemcpyAsync( dO, O3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dD, D3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dT, tmax, (size_t)n * 4, hipMemcpyHostToDevice, c->stream ) );
	if (( r = launch_trace_rays( c, n, dO, dD, dT, any_hit, dobj, dt ) )) return r;
	hipError_t e1 = hipMemcpyAsync( obj_out, dobj, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e2 = hipMemcpyAsync( t_out, dt, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e3 = hipStreamSynchronize( c->stream );
	HIPCHK( e1 ); HIPCHK( e2 ); HIPCHK( e3 );
	return RTMI_OK;
}

// rtmi_trace_rays without any allocation: the rays are staged in this device's copy of `scratch` (through the address kept at
// rtmi_create, never the array's name).  Nothing in the array has to start at zero: the copies fill the inputs of rays
// 0 .. n-1 and the kernel writes both outputs of every one of them, so what an earlier, larger call left behind is never read.
// Calls that overlap on one GPU (two handles, two host threads) share the array and race.
int rtmi_trace_rays_noalloc( rtmi_ctx* c, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int32_t* obj_out, float* t_out )
{
	if (!c || n < 0 || ( n && ( !O3 || !D3 || !tmax || !obj_out || !t_out ) )) return RTMI_ERR_INVALID;
	if (!c->has_scene) return fail( c, RTMI_ERR_NO_SCENE, "rtmi_trace_rays_noalloc before rtmi_upload_scene" );
	if (n > RTMI_NOALLOC_MAX_RAYS) return fail( c, RTMI_ERR_INVALID, "rtmi_trace_rays_noalloc: at most " + std::to_string( RTMI_NOALLOC_MAX_RAYS ) + " rays per call (use rtmi_trace_rays)" );
	if (n == 0) return RTMI_OK;
	HIPCHK( hipSetDevice( c->device ) );
	float* dO = (float*)c->noalloc_scratch, *dD = dO + 3 * (size_t)RTMI_NOALLOC_MAX_RAYS, *dT = dD + 3 * (size_t)RTMI_NOALLOC_MAX_RAYS, *dt = dT + RTMI_NOALLOC_MAX_RAYS;
	int* dobj = (int*)( dt + RTMI_NOALLOC_MAX_RAYS );
	HIPCHK( hipMemcpyAsync( dO, O3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dD, D3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dT, tmax, (size_t)n * 4, hipMemcpyHostToDevice, c->stream ) );
	// the same kernels as rtmi_trace_rays; option hook_kernel = 1 (the render path's queues) needs a render batch's buffers,
	// which may have to be allocated, so it is served by kernel 0 here
	if (c->hook_kernel == 2)
		hipLaunchKernelGGL( k_rays_wave, dim3( std::max( 1, std::min( ( n + 3 ) / 4, c->n_cus * 4 ) ) ), dim3( 256 ), 0, c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
	else
		hipLaunchKernelGGL( k_trace_rays, dim3( nblocks( n ) ), dim3( BLOCK ), stack_bytes( c ), c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
	HIPCHK( hipGetLastError() );
	hipError_t e1 = hipMemcpyAsync( obj_out, dobj, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e2 = hipMemcpyAsync( t_out, dt, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e3 = hipStreamSynchronize( c->stream );
	HIPCHK( e1 ); HIPCHK( e2 ); HIPCHK( e3 );
	return RTMI_OK;
}

static int run_postfx( rtmi_ctx* c, uint32_t* px, uint32_t* tmp, int W, int H, const rtmi_postfx* fx )
{
	const int n = W * H;
	if (fx->gamma_on)
	{
		// GammaCorrection, postprocessing.h:78-99
		if (c->gamma_cached != fx->gamma) // the table lives in the handle: no host wait for a stack copy, no re-upload per frame
		{
			const float gc = 1.0f / fx->gamma;
			HIPCHK( hipStreamSynchronize( c->stream ) ); // a copy of the previous table may still be reading gamma_host
			for (int i = 0; i < 256; i++) c->gamma_host[i] = (uint32_t)( 255.0f * powf( (float)i / 255.0f, gc ) ) & 255u;
			HIPCHK( hipMemcpyAsync( c->gamma_lut, c->gamma_host, sizeof( c->gamma_host ), hipMemcpyHostToDevice, c->stream ) );
			c->gamma_cached = fx->gamma;
		}
		hipLaunchKernelGGL( k_gamma, dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, px, n, c->gamma_lut );
	}
	if (fx->chroma_on)
	{
		hipLaunchKernelGGL( k_chroma, dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, px, tmp, W, H, fx->chroma_intensity );
		HIPCHK( hipMemcpyAsync( px, tmp, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream ) ); // postprocessing.h:75
	}
	if (fx->vignette_on) hipLaunchKernelGGL( k_vignette, dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, px, W, H, fx->vignette_intensity );
	HIPCHK( hipGetLastError() );
	return RTMI_OK;
}

int rtmi_finalize( rtmi_ctx* c, uint32_t* out, int frames, const rtmi_postfx* fx )
{
	if (!c || !out) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c, c->preview_reads ); if (r) return r; }
	const int n = c->W * c->H;
	{
		ScopedTimer t( c, 3, c->stream );
		hipLaunchKernelGGL( k_finalize, dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, c->accum, c->pixels, n, frames );
	}
	if (fx) { int r = run_postfx( c, c->pixels, c->pixels2, c->W, c->H, fx ); if (r) return r; }
	HIPCHK( hipMemcpyAsync( out, c->pixels, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	drain_closed( c );
	return RTMI_OK;
}

int rtmi_postprocess( rtmi_ctx* c, uint32_t* px, int W, int H, const rtmi_postfx* fx )
{
	if (!c || !px || !fx || W <= 0 || H <= 0) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	const size_t n = (size_t)W * H;
	int r = hook_reserve( c, n * 8 );
	if (r) return r;
	uint32_t* a = (uint32_t*)c->hook_buf, *b = a + n;
	hipError_t e = hipMemcpyAsync( a, px, n * 4, hipMemcpyHostToDevice, c->stream );
	if (e == hipSuccess) r = run_postfx( c, a, b, W, H, fx );
	if (e == hipSuccess && !r) e = hipMemcpyAsync( px, a, n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e2 = hipStreamSynchronize( c->stream );
	if (r) return r;
	HIPCHK( e ); HIPCHK( e2 );
	return RTMI_OK;
}

int rtmi_read_accumulator( rtmi_ctx* c, float* out )
{
	if (!c || !out) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	{ int r = join_batches( c ); if (r) return r; }
	HIPCHK( hipMemcpyAsync( out, c->accum, (size_t)c->W * c->H * 16, hipMemcpyDeviceToHost, c->stream ) );
	HIPCHK( hipStreamSynchronize( c->stream ) );
	drain_closed( c );
	return RTMI_OK;
}

// ---- multi-GPU inside the boundary: one RCCL reduce of the float4 accumulators ---------------------------------
// librccl is bound at first use with dlopen/dlsym so that librtmi.so neither needs it at load time nor brings a second copy
// into a process that already has one (torch ships its own librccl.so.1: the same SONAME resolves to the loaded copy).
typedef int ( *rccl_reduce_fn )( const void*, void*, size_t, int, int, int, void*, hipStream_t );
typedef int ( *rccl_allreduce_fn )( const void*, void*, size_t, int, int, void*, hipStream_t );
typedef const char* ( *rccl_errstr_fn )( int );
static struct { std::once_flag once; rccl_reduce_fn reduce = 0; rccl_allreduce_fn allreduce = 0; rccl_errstr_fn errstr = 0; std::string how; } g_rccl;
static void bind_rccl()
{
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	void* h = 0;
	for (const char* n : names) if (!h) h = dlopen( n, RTLD_NOW | RTLD_NOLOAD ); // a copy the process already loaded
	for (const char* n : names) if (!h) h = dlopen( n, RTLD_NOW | RTLD_LOCAL );
	if (!h) { g_rccl.how = std::string( "cannot load librccl.so.1: " ) + ( dlerror() ? dlerror() : "?" ); return; }
	g_rccl.reduce = (rccl_reduce_fn)dlsym( h, "ncclReduce" ), g_rccl.allreduce = (rccl_allreduce_fn)dlsym( h, "ncclAllReduce" );
	g_rccl.errstr = (rccl_errstr_fn)dlsym( h, "ncclGetErrorString" );
	if (!g_rccl.reduce || !g_rccl.allreduce) g_rccl.how = "librccl has no ncclReduce / ncclAllReduce", g_rccl.reduce = 0, g_rccl.allreduce = 0;
}
int rtmi_reduce_accumulator( rtmi_ctx* c, void* comm, int root )
{
	if (!c || !comm) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	std::call_once( g_rccl.once, bind_rccl );
	if (!g_rccl.reduce) return fail( c, RTMI_ERR_HIP, g_rccl.how );
	{ int r = join_batches( c ); if (r) return r; } // whole samples only, and the collective is ordered after them on the handle's stream
	const size_t count = (size_t)c->W * c->H * 4;
	const int ncclFloat = 7, ncclSum = 0; // rccl.h:448, 466
	const int e = root >= 0 ? g_rccl.reduce( c->accum, c->accum, count, ncclFloat, ncclSum, root, comm, c->stream )
		: g_rccl.allreduce( c->accum, c->accum, count, ncclFloat, ncclSum, comm, c->stream );
	if (e != 0) return fail( c, RTMI_ERR_HIP, std::string( root >= 0 ? "ncclReduce: " : "ncclAllReduce: " ) + ( g_rccl.errstr ? g_rccl.errstr( e ) : "error" ) );
	return RTMI_OK;
}

int rtmi_set_profiling( rtmi_ctx* c, int ev, int cn )
{
	if (!c) return RTMI_ERR_INVALID;
	if (ev && !c->prof_events)
	{
		HIPCHK( hipSetDevice( c->device ) );
		if (!c->prof_base && hipEventCreate( &c->prof_base ) != hipSuccess) c->prof_base = 0;
		if (c->prof_base) { HIPCHK( hipEventRecord( c->prof_base, c->stream ) ); HIPCHK( hipEventSynchronize( c->prof_base ) ); }
	}
	c->prof_events = ev != 0, c->count_nodes = cn != 0;
	return RTMI_OK;
}
// the length of the union of the closest-hit walk launches' intervals (they overlap when two lanes render side by side)
static double union_ms( std::vector<std::pair<float, float>> v )
{
	std::sort( v.begin(), v.end() );
	double total = 0; float hi = -1e30f;
	for (auto& i : v)
	{
		if (i.first > hi) total += i.second - i.first, hi = i.second;
		else if (i.second > hi) total += i.second - hi, hi = i.second;
	}
	return total;
}
int rtmi_get_stats( rtmi_ctx* c, rtmi_stats* out )
{
	if (!c || !out) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	int r = fetch_counters( c );
	if (r) return r;
	resolve_timers( c );
	c->stats.ms_extend_union = union_ms( c->ext_intervals ), c->stats.ms_walk_union = union_ms( c->walk_intervals );
	*out = c->stats;
	return RTMI_OK;
}
int rtmi_reset_stats( rtmi_ctx* c )
{
	if (!c) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	int r = fetch_counters( c );
	if (r) return r;
	resolve_timers( c );
	c->stats = rtmi_stats{};
	c->ext_intervals.clear(), c->walk_intervals.clear();
	return RTMI_OK;
}

} // extern "C"
