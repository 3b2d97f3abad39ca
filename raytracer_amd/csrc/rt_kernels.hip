// rt_kernels.hip -- the gfx950 kernels of librtmi.so and one launcher per kernel (rt_shared.h); the C ABI of include/rtmi.h and
// everything else that runs on the host is rtmi_host.cpp.
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
#include "../../include/rtmi.h" // the enums rtmi_mode / rtmi_material
#include <mutex>



// ---- queue append.  One device-scope atomic on a single word retires at ~88 per microsecond on this chip, so a
// per-wave atomic (half a million waves per bounce) is the bottleneck of a wavefront path tracer; here a whole
// 1024-thread block (16 waves) reserves its entries with ONE atomic per queue: ballot + mbcnt prefix inside the wave,
// a 16-entry scan in LDS across waves. ----------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void block_append( const bool ( &push )[NQ], uint32_t* const ( &counters )[NQ], uint32_t ( &idx )[NQ], uint32_t* lds /* NQ * 16 (+ 2 * NQ with `span`) */,
	uint32_t ( *span )[2] = nullptr /* [NQ][2]: the block's first entry in each queue and how many it appends */ )
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
		if (span) lds[NQ * 16 + 2 * q] = base, lds[NQ * 16 + 2 * q + 1] = total;
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < NQ; q++) idx[q] = lds[q * 16 + wave] + rank[q]; // (valid in every lane: the entries reserved by the lanes before it)
	if (span)
	{
#pragma unroll
		for (int q = 0; q < NQ; q++) span[q][0] = lds[NQ * 16 + 2 * q], span[q][1] = lds[NQ * 16 + 2 * q + 1];
	}
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
// `qray` (may be null): set when the ray is one that option qbvh sends through the reference's QBVH (an infinite reciprocal direction
// component in a scene that carries the QBVH): the queue entry is flagged, k_trace skips it, k_qbvh_fix walks it.
__device__ __forceinline__ bool begin_closest( const DScene& s, f3 O, f3 D, RayHit& h, float t0 = 1e34f, bool* qray = nullptr )
{
	h.t = t0, h.obj = -1, h.pos = -1;
	if (qray) *qray = false;
	isect_analytic( s, O, D, h );
	if (s.n_bvh_prims == 0) return false;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (inexact && s.qnodes) { if (qray) *qray = true; return true; } // IntersectQBVH never tests the primitive root's own box (bvh.h:603)
	return ( inexact ? bvh_enter<true>( s, O, rD, h.t ) : bvh_enter<false>( s, O, rD, h.t ) ) != RT_DONE;
}
// Same for Scene::IsOccluded (scene.h:369-379): quad and cube with the shrinking ray.t, then the planes with early out.
__device__ __forceinline__ void begin_any( const DScene& s, f3 O, f3 D, float tmaxRay, bool& occluded, bool& walk, bool* qray = nullptr )
{
	RayHit h; h.t = tmaxRay, h.obj = -1, h.pos = -1;
	if (qray) *qray = false;
	if (s.n_quads > 0) isect_quad<true>( s, 0, O, D, h ); // (<true>: quotient_reaches, rt_device.h -- the divisions that cannot matter are left out)
	if (s.n_cubes > 0) isect_cube( s, 0, O, D, h );
	if (h.obj == -1) isect_planes<true>( s, O, D, h );
	occluded = h.obj != -1, walk = false;
	if (occluded || s.n_bvh_prims == 0) return;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (inexact && s.qnodes) { if (qray) *qray = true; walk = true; return; }
	walk = ( inexact ? bvh_enter<true>( s, O, rD, h.t ) : bvh_enter<false>( s, O, rD, h.t ) ) != RT_DONE;
}

// ---- ray generation (bounce 0): camera ray, per-(pixel, sample) rng state, analytic part of FindNearest --------------
__global__ __launch_bounds__( SBLOCK ) void k_primary( DScene s, PathState st, float4* __restrict__ E, float4* __restrict__ Es, uint32_t N, CamProp cam, int W, int H, float invW, float invH,
	const uint32_t* __restrict__ seeds, int sample_begin, int sample_stride, uint32_t* __restrict__ wq, DevCounters* ctr )
{
	__shared__ uint32_t lds[16 + 2];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	bool walk = false, qray = false;
	f3 O = mk3( 0, 0, 0 ), D = mk3( 0, 0, 1 );
	RayHit h; h.t = 1e34f, h.obj = -1, h.pos = -1;
	uint32_t seed = 0;
	if (i < N)
	{
		const uint32_t WH = (uint32_t)( W * H ), p = i % WH, sidx = (uint32_t)sample_begin + ( i / WH ) * (uint32_t)sample_stride;
		primary_ray( cam, (float)(int)( p % (uint32_t)W ), (float)(int)( p / (uint32_t)W ), invW, invH, O, D );
		walk = begin_closest( s, O, D, h, 1e34f, &qray );
		seed = sample_seed( seeds[p], sidx );
		E[i] = make_float4( 0, 0, 0, 0 );
		Es[2 * (size_t)i] = make_float4( 0, 0, 0, 0 ), Es[2 * (size_t)i + 1] = make_float4( 0, 0, 0, 0 ); // 4 x int64 zero
	}
	const bool push[1] = { walk };
	uint32_t* const counters[1] = { &ctr->n_walk[0] };
	uint32_t idx[1], span[1][2];
	block_append<1>( push, counters, idx, lds, span );
	if (i < N)
	{
		// the block's paths stay in the block's run of entries, the ones that walk the BVH last (as k_shade orders its survivors):
		// slot i is entry j
		const uint32_t first = blockIdx.x * blockDim.x, valid = N - first < blockDim.x ? N - first : blockDim.x;
		const uint32_t walkersBefore = idx[0] - span[0][0], nonWalkers = valid - span[0][1];
		const uint32_t j = first + ( walk ? nonWalkers + walkersBefore : threadIdx.x - walkersBefore );
		st.o_t[j] = make_float4( O.x, O.y, O.z, h.t );
		st.d_h[j] = make_float4( D.x, D.y, D.z, __int_as_float( h.obj ) );
		st.T_s[j] = make_float4( 1, 1, 1, __uint_as_float( seed ) );
		st.meta[j] = i | ( FLAG_SPECULAR << 30 );
		if (walk) wq[idx[0]] = j | ( qray ? WQ_INEXACT : 0u );
	}
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
	const bool use_wide = ( !COUNT || a.count_wide ) && s.wide != nullptr;
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
	TravCount tc; tc.pairs = 0, tc.prims = 0, tc.wide = 0, tc.confirms = 0;
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
						bool qray; // option qbvh: a ray that k_qbvh_fix walks, not this kernel
						if (ANY)
						{
							entry = my;
							const float4 ot = a.sq.o_t[entry], ds = a.sq.d_s[entry];
							O = mk3( ot ), D = mk3( ds ), h.t = ot.w, h.obj = -1, slot = __float_as_uint( ds.w );
							resumed = ( slot & RESUME_BIT ) != 0, slot &= ~RESUME_BIT, qray = ( slot & SQ_INEXACT ) != 0;
							if (resumed) rec = __float_as_uint( a.sq.c[entry].w );
						}
						else
						{
							const uint32_t e = a.wq[my];
							entry = e & ~( RESUME_BIT | WQ_INEXACT ), resumed = ( e & RESUME_BIT ) != 0, qray = ( e & WQ_INEXACT ) != 0;
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
						if (qray) cur = RT_DONE, inexact = false, occluded = true; // retires with the next round, untouched: no hit written, nothing added
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
				if (atNode && !inexact) { if (COUNT) tc.wide++, steps++; cur = wide_step<ANY>( s, cur, O, rD, h.t, stack, BLOCK, sp ); }
				if (__ballot( atNode && inexact ) != 0) { if (atNode && inexact) { if (COUNT) tc.pairs++, steps++; cur = interior_step<ANY, true>( s, cur, O, rD, h.t, stack, BLOCK, sp ); } }
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
				if (COUNT) tc.confirms++;
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
		wave_add64( tc.wide, &ctr->wide_nodes[ANY ? 1 : 0] ), wave_add64( tc.confirms, &ctr->confirms[ANY ? 1 : 0] );
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

// ---- option qbvh: the rays that k_trace leaves alone ------------------------------------------------------------------------------
// A ray with an infinite reciprocal direction component meets 0 * inf = NaN in the slab tests, and there the reference's default
// traversal (IntersectQBVH: SSE min / max, no test of the primitive root's box, bvh.h:455-638) decides differently from its binary
// one.  Their queue entries are flagged by the kernel that creates the ray (begin_closest / begin_any), k_trace retires them
// untouched, and this kernel -- launched behind every k_trace of a handle whose scene carries the QBVH, over the same queue -- walks
// the reference's own nodes for them, whole, one ray per lane.  They are few (the centre row and column of the camera rays: a zero
// direction component needs an exact zero), so the kernel is a scan of the queue's flags; and keeping traverse_qbvh out of k_trace
// keeps that kernel what it is without the option (inside it, the closest-hit walk was 37 % slower for rays that never take the branch).
template <bool ANY>
__global__ __launch_bounds__( BLOCK ) void k_qbvh_fix( DScene s, TraceArgs a )
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t* stack = lds_stack + threadIdx.x;
	const uint32_t n = *a.count_ptr;
	for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
	{
		RayHit h; h.obj = -1, h.pos = -1;
		if (ANY)
		{
			const uint32_t slot = __float_as_uint( a.sq.d_s[i].w );
			if (!( slot & SQ_INEXACT )) continue;
			const float4 ot = a.sq.o_t[i], ds = a.sq.d_s[i];
			const f3 O = mk3( ot ), D = mk3( ds );
			h.t = ot.w;
			if (!traverse_qbvh<true>( s, O, D, mk3( 1 / D.x, 1 / D.y, 1 / D.z ), h, stack, BLOCK ))
				add_walked( a.sinks, slot & META_SLOT_MASK, ( slot & SQ_LATE ) != 0, a.sq.c[i] );
		}
		else
		{
			const uint32_t e = a.wq[i];
			if (!( e & WQ_INEXACT )) continue;
			const uint32_t entry = e & ~( RESUME_BIT | WQ_INEXACT );
			const float4 ot = a.st.o_t[entry], dh = a.st.d_h[entry];
			const f3 O = mk3( ot ), D = mk3( dh );
			h.t = ot.w, h.obj = __float_as_int( dh.w ); // the analytic hit so far
			traverse_qbvh<false>( s, O, D, mk3( 1 / D.x, 1 / D.y, 1 / D.z ), h, stack, BLOCK );
			if (h.pos >= 0) a.st.o_t[entry].w = h.t, a.st.d_h[entry].w = __int_as_float( h.obj );
		}
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
			// Same bits with fewer instructions: x * 2 is exact, so fma( x, 2, -1 ) rounds what x * 2.0f - 1.0f rounds; and with a correctly
			// rounded square root sqrtf( s ) > 1.0f exactly when s > 1 + 2^-23 (sqrt( 1 + 2^-23 ) = 1 + 2^-24 - ... rounds to 1; checked for
			// every float in [0.25, 4)); normalize( R ) divides by the square root of the same sum
			f3 R;
			float len2;
			do
			{
				const float rx = __builtin_fmaf( rng_float( rng ), 2.0f, -1.0f );
				const float ry = __builtin_fmaf( rng_float( rng ), 2.0f, -1.0f );
				const float rz = __builtin_fmaf( rng_float( rng ), 2.0f, -1.0f );
				R = mk3( rx, ry, rz );
				len2 = rx * rx + ry * ry + rz * rz; // magnitude( R )^2, precomp.h:848
			} while (len2 > 1.00000011920928955078125f);
			R = R * ( 1.0f / sqrtf( len2 ) ); // normalize( R )
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
	__shared__ uint32_t lds[3 * 16 + 6 + 1];
	// This kernel is bound by instruction issue, the walks that run beside it (the other lane's, this lane's shadow walk) by memory
	// latency: its waves go first on the SIMDs they share.  Two-lane bench: k_shade 1 514 -> 1 360 ms per 20 steps, the walks
	// 2 150 -> 2 460, +0.7 % rays/s in all (four A/B pairs on two boxes); a priority for the closest-hit walk on top of it: nothing.
	__builtin_amdgcn_s_setprio( 1 );
	if (threadIdx.x == 0) lds[54] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n = *n_in;
	bool walk_next = false, push_shadow = false, parked = false, qray = false, qshadow = false;
	uint32_t slot = 0, flags = 0, rng = 0, age = 0;
	bool late = false;
	f3 T;
	Vertex v; v.push_next = false, v.shadow_any = false, v.shT = 0;
	RayHit nh; nh.t = 1e34f, nh.obj = -1, nh.pos = -1;
	if (i < n)
	{
		const float4 ot = ld_stream( &in.o_t[i] ), dh = ld_stream( &in.d_h[i] ), ts = ld_stream( &in.T_s[i] );
		const uint32_t meta = ld_stream( &in.meta[i] );
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
			if (v.push_next) walk_next = begin_closest( s, v.newO, v.newD, nh, 1e34f, &qray );
			if (v.shadow_any)
			{
				bool occluded;
				begin_any( s, v.shO, v.shD, v.shT, occluded, push_shadow, &qshadow );
				if (!occluded && !push_shadow) add_chain( sinks, slot, late, v.contrib ); // unoccluded without a walk: E += contribution now (renderer.cpp:343)
			}
		}
	}
	{
		const unsigned long long m = __ballot( v.shadow_any );
		if (( threadIdx.x & 63 ) == 0 && m) atomicAdd( &lds[54], (uint32_t)__popcll( m ) );
	}
	const bool push[3] = { v.push_next, walk_next, push_shadow };
	uint32_t* const counters[3] = { n_out, n_walk_out, n_shadow };
	uint32_t idx[3];
	uint32_t span[3][2];
	block_append<3>( push, counters, idx, lds, span );
	if (v.push_next)
	{
		// Within the block's run of entries the paths whose next ray walks the BVH come LAST: every hit on a mesh -- the mirror and
		// glass branches of shade_vertex, the triangle branch of hit_attributes -- belongs to a walker, so the next bounce's waves
		// over the front of the run take none of those branches, and k_trace finds its rays' state side by side.  (Order inside a
		// queue has no influence on any path's arithmetic.)
		const uint32_t before = idx[0] - span[0][0], walkersBefore = idx[1] - span[1][0], nonWalkers = span[0][1] - span[1][1];
		const uint32_t j = span[0][0] + ( walk_next ? nonWalkers + walkersBefore : before - walkersBefore );
		st_stream( &out.o_t[j], make_float4( v.newO.x, v.newO.y, v.newO.z, nh.t ) );
		st_stream( &out.d_h[j], make_float4( v.newD.x, v.newD.y, v.newD.z, __int_as_float( nh.obj ) ) );
		st_stream( &out.T_s[j], make_float4( T.x, T.y, T.z, __uint_as_float( rng ) ) );
		st_stream( &out.meta[j], slot | ( age << META_AGE_SHIFT ) | ( flags << 30 ) | ( parked ? META_SUSPENDED : 0u ) );
		if (walk_next) st_stream( &wq_out[idx[1]], j | ( parked ? RESUME_BIT : 0u ) | ( qray ? WQ_INEXACT : 0u ) );
	}
	if (push_shadow)
	{
		const uint32_t si = idx[2];
		st_stream( &sq.o_t[si], make_float4( v.shO.x, v.shO.y, v.shO.z, v.shT ) );
		st_stream( &sq.d_s[si], make_float4( v.shD.x, v.shD.y, v.shD.z, __uint_as_float( slot | ( late ? SQ_LATE : 0u ) | ( qshadow ? SQ_INEXACT : 0u ) ) ) );
		st_stream( &sq.c[si], make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
	}
	if (threadIdx.x == 0 && lds[54]) atomicAdd( &ctr->shadow_rays, (unsigned long long)lds[54] );
}

// ---- tail: once only a few thousand paths are alive a bounce is pure latency (three launches, each as long as one
// ray's walk), and paths trapped inside the mirror or glass object live for thousands of bounces (the reference has
// no bounce limit, renderer.cpp:251).  This kernel runs each remaining path for up to `max_bounces` bounces in one
// launch: one lane per path loops FindNearest -> shade_vertex -> IsOccluded.  Paths still alive are written compacted
// to the other state set for the next round, so that a round's waves are dense again (nearly every wave of the first
// round holds one long path, and a thousand almost-empty resident waves would starve the next batches' bulk kernels).
// Same device functions and the same per-path order of operations as the wavefront kernels: results do not depend on
// when the host switches over or on the round lengths. ------------------------------------------------------------------
template <bool QBVH>
__global__ __launch_bounds__( 1024 ) void k_tail( DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, int max_bounces, uint32_t dense_above, int min_lanes, uint32_t lds_words, uint32_t team_below, DevCounters* ctr, int stop_late )
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
	int per_wave;
	if (hurry)
	{
		const uint32_t waves = gridDim.x * ( blockDim.x >> 6 );
		const uint32_t need = ( n + waves - 1 ) / waves;
		per_wave = 1;
		while ((uint32_t)per_wave < need && per_wave < 64) per_wave *= 2;
	}
	else per_wave = n > dense_above ? 64 : min_lanes;
	const uint32_t stride = 64u / (uint32_t)per_wave;
	// Workgroups of several waves: the waves of this kernel live for milliseconds, and a CU that hosts one serves the
	// blocks of the next batches' k_shade at about half speed (the old wave wins the issue arbitration; the block's barriers
	// wait for its slowest wave): 250 tail waves spread one to a CU slowed every k_shade launch by 1.5-2x.  Waves that the
	// round's live path count does not need leave at once, so the ones that stay are the first blocks of the grid: a few CUs.
	const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
	if (( blockIdx.x * waves_per_block + wave_in_block ) * (uint32_t)per_wave >= n) return; // wave-uniform
	uint32_t* stack = lds_stack + wave_in_block * ( lds_words / waves_per_block ) + lane / stride; // one column per active lane
	unsigned long long nExt = 0, nSh = 0, nExtWalk = 0, nShWalk = 0;
	TravCount tc = {};
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
				const bool walk = begin_closest( s, O, D, h );
				nExt++;
				if (walk)
				{
					nExtWalk++;
					const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
					const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
					if (inexact) walk_inexact<false, QBVH>( s, O, D, rD, h, stack, per_wave );
					else if (s.wide) traverse_wide<false>( s, O, D, rD, h, stack, per_wave );
					else traverse<false, false, false>( s, O, D, rD, h, stack, per_wave, tc );
				}
				Vertex v;
				const bool late = age == META_AGE_LATE;
				shade_vertex( s, O, D, h.t, h.obj, T, rng, flags, sinks, slot, late, v );
				age = age < META_AGE_LATE ? age + 1 : age;
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
						const bool occ = inexact ? walk_inexact<true, QBVH>( s, v.shO, v.shD, rD, sh, stack, per_wave )
							: s.wide ? traverse_wide<true>( s, v.shO, v.shD, rD, sh, stack, per_wave ) : traverse<true, false, false>( s, v.shO, v.shD, rD, sh, stack, per_wave, tc );
						if (!occ) add_walked( sinks, slot, late, make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
					}
				}
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
template <bool QBVH>
__global__ __launch_bounds__( 256, 4 ) void k_tail_team( DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, int max_bounces, DevCounters* ctr )
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
	uint32_t nSteps = 0, nPrims = 0, nStepsAny = 0, nPrimsAny = 0;
	TravCount tc = {};
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
				const bool walk = begin_closest( s, O, D, h );
				if (j == 0) nExt++;
				if (walk)
				{
					if (j == 0) nExtWalk++;
					const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
					const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
					// rays that the 4-wide nodes cannot take walk the pairs, the four lanes in step (same values to the same column)
					if (inexact) walk_inexact<false, QBVH>( s, O, D, rD, h, stack, stride );
					else if (s.wide) team_traverse_wide<false>( s, j, O, D, rD, h, stack, stride, nSteps, nPrims );
					else traverse<false, false, false>( s, O, D, rD, h, stack, stride, tc );
				}
				Vertex v;
				const bool late = age == META_AGE_LATE;
				shade_vertex( s, O, D, h.t, h.obj, T, rng, flags, sinks, slot, late, v, j == 0 );
				age = age < META_AGE_LATE ? age + 1 : age;
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
						const bool occ = inexact ? walk_inexact<true, QBVH>( s, v.shO, v.shD, rD, sh, stack, stride )
							: s.wide ? team_traverse_wide<true>( s, j, v.shO, v.shD, rD, sh, stack, stride, nStepsAny, nPrimsAny )
							: traverse<true, false, false>( s, v.shO, v.shD, rD, sh, stack, stride, tc );
						if (!occ && j == 0) add_walked( sinks, slot, late, make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
					}
				}
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
}

// ---- the very end of a render: one path per wave ---------------------------------------------------------------------------
// Below a few thousand live paths even a team leaves 60 lanes of its wave idle while its path's walk is a chain of ~30
// dependent steps.  Here the whole wave walks the one path breadth-first (rt_device.h: wave_traverse_wide), the analytic
// primitives are tested one per lane, and shading runs replicated as in k_tail_team; lane 0 writes.
template <bool QBVH>
__device__ __forceinline__ void wave_find_nearest( const DScene& s, int lane, f3 O, f3 D, float t0, RayHit& h, WaveLds& L, bool& walked, uint32_t& nIter )
{
	wave_analytic( s, lane, O, D, t0, h );
	walked = false;
	if (s.n_bvh_prims == 0) return;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (!( QBVH && inexact ) && ( inexact ? bvh_enter<true>( s, O, rD, h.t ) : bvh_enter<false>( s, O, rD, h.t ) ) == RT_DONE) return;
	walked = true;
	TravCount tc = {};
	// rays the 4-wide nodes cannot take, and rays whose frontier outgrows the LDS, walk as in k_tail: the 64 lanes in step,
	// one stack column (the same values to the same addresses)
	if (inexact) { walk_inexact<false, QBVH>( s, O, D, rD, h, L.nref, 1 ); return; }
	if (!s.wide) { traverse<false, false, false>( s, O, D, rD, h, L.nref, 1, tc ); return; }
	bool overflow;
	wave_traverse_wide<false>( s, lane, O, D, rD, h, L, overflow, nIter );
	if (overflow) traverse_wide<false>( s, O, D, rD, h, L.nref, 1 );
}
template <bool QBVH>
__device__ __forceinline__ bool wave_is_occluded( const DScene& s, int lane, f3 O, f3 D, float tmaxRay, WaveLds& L, bool& walked, uint32_t& nIter )
{
	walked = false;
	if (wave_analytic_any( s, lane, O, D, tmaxRay )) return true;
	if (s.n_bvh_prims == 0) return false;
	const f3 rD = mk3( 1 / D.x, 1 / D.y, 1 / D.z );
	const bool inexact = !( fabsf( rD.x ) < INFINITY && fabsf( rD.y ) < INFINITY && fabsf( rD.z ) < INFINITY );
	if (!( QBVH && inexact ) && ( inexact ? bvh_enter<true>( s, O, rD, tmaxRay ) : bvh_enter<false>( s, O, rD, tmaxRay ) ) == RT_DONE) return false;
	walked = true;
	RayHit sh; sh.t = tmaxRay, sh.obj = -1, sh.pos = -1;
	TravCount tc = {};
	if (inexact) return walk_inexact<true, QBVH>( s, O, D, rD, sh, L.nref, 1 );
	if (!s.wide) return traverse<true, false, false>( s, O, D, rD, sh, L.nref, 1, tc );
	bool overflow;
	const bool occ = wave_traverse_wide<true>( s, lane, O, D, rD, sh, L, overflow, nIter );
	return overflow ? traverse_wide<true>( s, O, D, rD, sh, L.nref, 1 ) : occ;
}

template <bool QBVH>
__global__ __launch_bounds__( 64 * WAVE_WG, 4 ) void k_tail_wave( DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* __restrict__ n_in,
	uint32_t* n_out, int max_bounces, DevCounters* ctr )
{
	__shared__ WaveLds lds[WAVE_WG];
	const uint32_t n = *n_in;
	if (( ctr->tail_mode & 3u ) != 2) return; // another tail kernel's round (k_tail_decide)
	const int lane = (int)( threadIdx.x & 63u );
	WaveLds& L = lds[threadIdx.x >> 6];
	if (( blockIdx.x * (uint32_t)WAVE_WG + ( threadIdx.x >> 6 ) ) >= n) return; // surplus waves leave at once
	unsigned long long nExt = 0, nSh = 0, nExtWalk = 0, nShWalk = 0;
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
			wave_find_nearest<QBVH>( s, lane, O, D, 1e34f, h, L, walked, nIter );
			nExt++, nExtWalk += walked ? 1 : 0;
			Vertex v;
			const bool late = age == META_AGE_LATE;
			shade_vertex( s, O, D, h.t, h.obj, T, rng, flags, sinks, slot, late, v, lane == 0 );
			age = age < META_AGE_LATE ? age + 1 : age;
			if (v.shadow_any)
			{
				nSh++;
				const bool occ = wave_is_occluded<QBVH>( s, lane, v.shO, v.shD, v.shT, L, walked, nIterAny );
				nShWalk += walked ? 1 : 0;
				if (!occ && lane == 0)
				{
					if (walked) add_walked( sinks, slot, late, make_float4( v.contrib.x, v.contrib.y, v.contrib.z, 0 ) );
					else add_chain( sinks, slot, late, v.contrib );
				}
			}
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
	}
}

// the parity hook through the same code (option hook_kernel = 2): wave w walks caller rays w, w + waves, ...
template <bool QBVH>
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
			const bool occ = wave_is_occluded<QBVH>( s, lane, O, D, tmax[i], L, walked, nIter );
			if (lane == 0) obj_out[i] = occ ? 1 : 0, t_out[i] = tmax[i];
		}
		else
		{
			RayHit h;
			wave_find_nearest<QBVH>( s, lane, O, D, tmax[i], h, L, walked, nIter );
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
__global__ __launch_bounds__( 512 ) void k_carry_out( PathState in, const uint32_t* __restrict__ n_in, PathState rest, uint32_t* n_rest, PathState carry, uint32_t* n_carry, uint32_t cap )
{
	__shared__ uint32_t lds_a[16], lds_b[16];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = *n_in;
	float4 ot = make_float4( 0, 0, 0, 0 ), dh = ot, ts = ot;
	uint32_t meta = 0;
	const bool valid = i < n;
	if (valid) ot = in.o_t[i], dh = in.d_h[i], ts = in.T_s[i], meta = in.meta[i];
	const bool late = valid && ( ( meta >> META_AGE_SHIFT ) & META_AGE_LATE ) == META_AGE_LATE && !( meta & META_SUSPENDED );
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
__global__ __launch_bounds__( 512 ) void k_carry_in( DScene s, PathState carry, const uint32_t* __restrict__ n_carry, uint32_t cap, PathState st, const uint32_t* __restrict__ n_st, uint32_t* __restrict__ wq, DevCounters* ctr )
{
	__shared__ uint32_t lds[16];
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t nc = *n_carry < cap ? *n_carry : cap, N = *n_st; // N changes only in k_carry_commit
	bool walk = false, qray = false;
	if (j < nc)
	{
		const float4 ot = carry.o_t[j], dh = carry.d_h[j];
		const f3 O = mk3( ot ), D = mk3( dh );
		RayHit h;
		walk = begin_closest( s, O, D, h, 1e34f, &qray ); // the entry's ray has not been traced yet
		st.o_t[N + j] = make_float4( O.x, O.y, O.z, h.t );
		st.d_h[N + j] = make_float4( D.x, D.y, D.z, __int_as_float( h.obj ) );
		st.T_s[N + j] = carry.T_s[j];
		st.meta[N + j] = carry.meta[j];
	}
	const bool push[1] = { walk };
	uint32_t* const counters[1] = { &ctr->n_walk[0] };
	uint32_t idx[1];
	block_append<1>( push, counters, idx, lds );
	if (walk) wq[idx[0]] = ( N + j ) | ( qray ? WQ_INEXACT : 0u );
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
// `drop`: this was the last tail round (launch_tail_rounds): what it left alive has used up its bounces and ends here, counted
__global__ void k_next_bounce( DevCounters* ctr, int consumed, uint32_t* report, int drop )
{
	if (drop) ctr->dropped += ctr->n_state[consumed ^ 1], ctr->n_state[consumed ^ 1] = 0;
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
template <bool QBVH>
__global__ __launch_bounds__( BLOCK ) void k_trace_rays( DScene s, int n, const float* __restrict__ O3, const float* __restrict__ D3, const float* __restrict__ tmax,
	int any_hit, int* __restrict__ obj_out, float* __restrict__ t_out )
{
	extern __shared__ uint32_t lds_stack[];
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	const bool live = i < n;
	const int k = live ? i : 0;
	const f3 O = mk3( O3[3 * k], O3[3 * k + 1], O3[3 * k + 2] ), D = mk3( D3[3 * k], D3[3 * k + 1], D3[3 * k + 2] );
	TravCount tc = {};
	if (live)
	{
		if (any_hit)
		{
			const bool occ = is_occluded<false, QBVH>( s, O, D, tmax[k], lds_stack + threadIdx.x, BLOCK, tc );
			obj_out[k] = occ ? 1 : 0, t_out[k] = tmax[k];
		}
		else
		{
			RayHit h; h.t = tmax[k], h.obj = -1, h.pos = -1;
			find_nearest<false, QBVH>( s, O, D, h, lds_stack + threadIdx.x, BLOCK, tc );
			obj_out[k] = h.obj, t_out[k] = h.t;
		}
	}
}
__global__ __launch_bounds__( BLOCK ) void k_copy_hits( PathState st, int n, int* __restrict__ obj_out, float* __restrict__ t_out )
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint32_t p = st.meta[i] & META_SLOT_MASK; // entry i holds the path of slot p (k_primary orders a block's entries; k_rays_begin does not)
	obj_out[p] = __float_as_int( st.d_h[i].w ), t_out[p] = st.o_t[i].w;
}
// The same hook through the render path's own machinery (option hook_kernel = 1): caller rays become entries of a path queue
// (closest hit) or of a shadow queue (any hit) exactly as k_primary / k_shade would have queued them, and k_trace walks them.
// An unoccluded shadow ray adds its "contribution" (1, 0, 0) to Es[ray]; k_rays_any_result turns that back into a flag.
__global__ __launch_bounds__( SBLOCK ) void k_rays_begin( DScene s, int n, const float* __restrict__ O3, const float* __restrict__ D3, const float* __restrict__ tmax, int any_hit,
	PathState st, ShadowQueue sq, float4* __restrict__ Es, uint32_t* __restrict__ wq, int* __restrict__ obj_out, float* __restrict__ t_out, DevCounters* ctr )
{
	__shared__ uint32_t lds[2 * 16];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	bool walk = false, swalk = false, qray = false;
	f3 O = mk3( 0, 0, 0 ), D = mk3( 0, 0, 1 );
	float tm = 0;
	if (i < n)
	{
		O = mk3( O3[3 * i], O3[3 * i + 1], O3[3 * i + 2] ), D = mk3( D3[3 * i], D3[3 * i + 1], D3[3 * i + 2] ), tm = tmax[i];
		if (any_hit)
		{
			bool occluded;
			begin_any( s, O, D, tm, occluded, swalk, &qray );
			obj_out[i] = occluded ? 1 : ( swalk ? 2 : 0 ), t_out[i] = tm;
			Es[2 * (size_t)i] = make_float4( 0, 0, 0, 0 ), Es[2 * (size_t)i + 1] = make_float4( 0, 0, 0, 0 );
		}
		else
		{
			RayHit h;
			walk = begin_closest( s, O, D, h, tm, &qray );
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
	if (walk) wq[idx[0]] = (uint32_t)i | ( qray ? WQ_INEXACT : 0u );
	if (swalk)
	{
		sq.o_t[idx[1]] = make_float4( O.x, O.y, O.z, tm );
		sq.d_s[idx[1]] = make_float4( D.x, D.y, D.z, __uint_as_float( (uint32_t)i | ( qray ? SQ_INEXACT : 0u ) ) );
		sq.c[idx[1]] = make_float4( 1, 0, 0, 0 );
	}
	if (i == 0) ctr->n_state[0] = (uint32_t)n;
}
__global__ __launch_bounds__( BLOCK ) void k_rays_any_result( const long long* __restrict__ Es, int n, int* __restrict__ obj_out )
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n && obj_out[i] == 2) obj_out[i] = Es[4 * (size_t)i] != 0 ? 0 : 1;
}

// ---- single-primitive and attribute hooks: the reference's per-object methods on the device ------------------------------------
// Triangle / Sphere / Plane / Quad / Cube::Intersect (primitive.h:50-72, 108-127, 154-158, 284-295, 210-236) for one primitive that
// the caller hands over by value, against n rays: the same device functions the walks use.  hit[i] = the ray was updated.
__global__ __launch_bounds__( BLOCK ) void k_isect_one( OnePrim p, int n, const float* __restrict__ O3, const float* __restrict__ D3, const float* __restrict__ tin,
	int* __restrict__ hit, float* __restrict__ tout )
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const f3 O = mk3( O3[3 * i], O3[3 * i + 1], O3[3 * i + 2] ), D = mk3( D3[3 * i], D3[3 * i + 1], D3[3 * i + 2] );
	RayHit h; h.t = tin[i], h.obj = -1, h.pos = -1;
	if (p.kind == 0)
	{
		// edge1 = pos2 - pos1, edge2 = pos3 - pos1 (primitive.h:55-56)
		const float4 a = make_float4( p.d[0].x, p.d[0].y, p.d[0].z, __int_as_float( 0 ) );
		const float4 b = make_float4( p.d[1].x - p.d[0].x, p.d[1].y - p.d[0].y, p.d[1].z - p.d[0].z, __int_as_float( 0 ) );
		const float4 c = make_float4( p.d[2].x - p.d[0].x, p.d[2].y - p.d[0].y, p.d[2].z - p.d[0].z, 0 );
		isect_record<false>( a, b, c, 0, O, D, h );
	}
	else if (p.kind == 1) isect_record<false>( make_float4( p.d[0].x, p.d[0].y, p.d[0].z, __int_as_float( 1 ) ), make_float4( p.d[0].w, 0, 0, __int_as_float( 0 ) ), make_float4( 0, 0, 0, 0 ), 0, O, D, h );
	else if (p.kind == 2) isect_plane_nd( p.d[0], 0, O, D, h );
	else if (p.kind == 3) { m34 M; M.r0 = p.d[0], M.r1 = p.d[1], M.r2 = p.d[2]; isect_quad_rows( M, p.d[3].x, 0, O, D, h ); }
	else { m34 M; M.r0 = p.d[0], M.r1 = p.d[1], M.r2 = p.d[2]; isect_cube_rows( M, p.d[3], p.d[4], 0, O, D, h ); }
	hit[i] = h.obj != -1 ? 1 : 0, tout[i] = h.t;
}
// Scene::GetNormal / GetAlbedo / GetMaterial (scene.h:380-478) for n hits of the uploaded scene: hit_attributes, as k_shade calls it
__global__ __launch_bounds__( BLOCK ) void k_hit_attribs( DScene s, int n, const int* __restrict__ obj, const float* __restrict__ I3, const float* __restrict__ wo3,
	float* __restrict__ N3, float* __restrict__ alb3, int* __restrict__ mat )
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	HitAttr a;
	hit_attributes( s, obj[i], mk3( I3[3 * i], I3[3 * i + 1], I3[3 * i + 2] ), mk3( wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2] ), a );
	N3[3 * i] = a.N.x, N3[3 * i + 1] = a.N.y, N3[3 * i + 2] = a.N.z;
	alb3[3 * i] = a.albedo.x, alb3[3 * i + 1] = a.albedo.y, alb3[3 * i + 2] = a.albedo.z;
	mat[i] = a.mat->type;
}

// ---- Whitted / debug views: Renderer::Trace without recursion (renderer.cpp:80-244) ------------------------
// Each pending call is a frame; a GLASS frame is visited up to three times (before the reflection child, before the
// refraction child, after both).  Frames live in a per-thread slice of a global buffer (depth <= max_depth).

template <bool QBVH>
__device__ __forceinline__ f3 direct_illumination( const DScene& s, f3 I, f3 N, uint32_t* stack, int stride ) // scene.h:490-515
{
	f3 acc = mk3( 0, 0, 0 );
	TravCount tc = {};
	for (int i = 0; i < 2; i++)
	{
		const f3 dir = mk3( s.wl_pos[i] ) - I;
		const f3 dirN = normalize3( dir );
		const float eps = 0.001f;
		const f3 so = I + dirN * eps;
		const float mag = magnitude3( dir );
		const float st = mag - eps * 2;
		if (!is_occluded<false, QBVH>( s, so, dirN, st, stack, stride, tc ))
		{
			const float distanceEnergy = 1 / sqrf_( st );
			const float d = dot3( N, dirN );
			const float angularEnergy = ( d < 0.0f ) ? 0.0f : d; // std::max(d, 0.0f)
			acc = acc + mk3( s.wl_color[i] ) * s.wl_intensity * distanceEnergy * angularEnergy;
		}
	}
	return acc;
}

template <bool QBVH>
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
	TravCount tc = {};
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
				find_nearest<false, QBVH>( s, f.O, f.D, h, stack, BLOCK, tc );
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
					ret = a.albedo / 3.14159265358979323846264f * direct_illumination<QBVH>( s, I, a.N, stack, BLOCK );
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


// ---- rtmi_trace_rays_noalloc: staging in a file-scope device array instead of allocated memory --------------------------
// 36 bytes per ray (origin, direction, tmax in; object id, distance out).  Every GPU the library runs on holds its own copy
// of the array, at its own address, until the process ends.  In host code the NAME is only a handle for
// hipGetSymbolAddress: the address kernels and copies use is looked up once per device (scratch_address, under the
// device being current) and kept.
static __device__ __align__( 256 ) unsigned char scratch[(size_t)RTMI_NOALLOC_MAX_RAYS * 36];
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[64] = {};
hipError_t rtk_scratch_address( int device, void** out ) // the caller has made `device` current
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

// ---- launchers: all the host side (rtmi_host.cpp) sees of the kernels above -------------------------------------------------
void rtk_primary( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState st, float4* E, float4* Es, uint32_t N, CamProp cam, int W, int H, float invW, float invH, const uint32_t* seeds, int sample_begin, int sample_stride, uint32_t* wq, DevCounters* ctr )
{
	hipLaunchKernelGGL( k_primary, grid, block, lds, stream, s, st, E, Es, N, cam, W, H, invW, invH, seeds, sample_begin, sample_stride, wq, ctr );
}
void rtk_trace( bool any, bool count, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, TraceArgs a, DevCounters* ctr )
{
	if (any && count) hipLaunchKernelGGL( ( k_trace<true, true> ), grid, block, lds, stream, s, a, ctr );
	else if (any && !count) hipLaunchKernelGGL( ( k_trace<true, false> ), grid, block, lds, stream, s, a, ctr );
	else if (!any && count) hipLaunchKernelGGL( ( k_trace<false, true> ), grid, block, lds, stream, s, a, ctr );
	else if (!any && !count) hipLaunchKernelGGL( ( k_trace<false, false> ), grid, block, lds, stream, s, a, ctr );
}
void rtk_qbvh_fix( bool any, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, TraceArgs a )
{
	if (any) hipLaunchKernelGGL( ( k_qbvh_fix<true> ), grid, block, lds, stream, s, a );
	else if (!any) hipLaunchKernelGGL( ( k_qbvh_fix<false> ), grid, block, lds, stream, s, a );
}
void rtk_shade( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, ShadowQueue sq, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, uint32_t* wq_out, uint32_t* n_walk_out, uint32_t* n_shadow, DevCounters* ctr )
{
	hipLaunchKernelGGL( k_shade, grid, block, lds, stream, s, in, out, sq, sinks, n_in, n_out, wq_out, n_walk_out, n_shadow, ctr );
}
void rtk_tail( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, int max_bounces, uint32_t dense_above, int min_lanes, uint32_t lds_words, uint32_t team_below, DevCounters* ctr, int stop_late )
{
	if (qbvh) hipLaunchKernelGGL( ( k_tail<true> ), grid, block, lds, stream, s, in, out, sinks, n_in, n_out, max_bounces, dense_above, min_lanes, lds_words, team_below, ctr, stop_late );
	else if (!qbvh) hipLaunchKernelGGL( ( k_tail<false> ), grid, block, lds, stream, s, in, out, sinks, n_in, n_out, max_bounces, dense_above, min_lanes, lds_words, team_below, ctr, stop_late );
}
void rtk_tail_decide( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, const uint32_t* n_in, const uint32_t* draining, uint32_t team_below, uint32_t wave_below )
{
	hipLaunchKernelGGL( k_tail_decide, grid, block, lds, stream, ctr, n_in, draining, team_below, wave_below );
}
void rtk_tail_team( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, int max_bounces, DevCounters* ctr )
{
	if (qbvh) hipLaunchKernelGGL( ( k_tail_team<true> ), grid, block, lds, stream, s, in, out, sinks, n_in, n_out, max_bounces, ctr );
	else if (!qbvh) hipLaunchKernelGGL( ( k_tail_team<false> ), grid, block, lds, stream, s, in, out, sinks, n_in, n_out, max_bounces, ctr );
}
void rtk_tail_wave( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, int max_bounces, DevCounters* ctr )
{
	if (qbvh) hipLaunchKernelGGL( ( k_tail_wave<true> ), grid, block, lds, stream, s, in, out, sinks, n_in, n_out, max_bounces, ctr );
	else if (!qbvh) hipLaunchKernelGGL( ( k_tail_wave<false> ), grid, block, lds, stream, s, in, out, sinks, n_in, n_out, max_bounces, ctr );
}
void rtk_rays_wave( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int* obj_out, float* t_out )
{
	if (qbvh) hipLaunchKernelGGL( ( k_rays_wave<true> ), grid, block, lds, stream, s, n, O3, D3, tmax, any_hit, obj_out, t_out );
	else if (!qbvh) hipLaunchKernelGGL( ( k_rays_wave<false> ), grid, block, lds, stream, s, n, O3, D3, tmax, any_hit, obj_out, t_out );
}
void rtk_carry_out( dim3 grid, dim3 block, size_t lds, hipStream_t stream, PathState in, const uint32_t* n_in, PathState rest, uint32_t* n_rest, PathState carry, uint32_t* n_carry, uint32_t cap )
{
	hipLaunchKernelGGL( k_carry_out, grid, block, lds, stream, in, n_in, rest, n_rest, carry, n_carry, cap );
}
void rtk_carry_in( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState carry, const uint32_t* n_carry, uint32_t cap, PathState st, const uint32_t* n_st, uint32_t* wq, DevCounters* ctr )
{
	hipLaunchKernelGGL( k_carry_in, grid, block, lds, stream, s, carry, n_carry, cap, st, n_st, wq, ctr );
}
void rtk_carry_commit( dim3 grid, dim3 block, size_t lds, hipStream_t stream, uint32_t* n_st, uint32_t* n_carry, uint32_t cap )
{
	hipLaunchKernelGGL( k_carry_commit, grid, block, lds, stream, n_st, n_carry, cap );
}
void rtk_fold_late( dim3 grid, dim3 block, size_t lds, hipStream_t stream, float4* acc, long long* late, float4* odd, int WH )
{
	hipLaunchKernelGGL( k_fold_late, grid, block, lds, stream, acc, late, odd, WH );
}
void rtk_next_bounce( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, int consumed, uint32_t* report, int drop )
{
	hipLaunchKernelGGL( k_next_bounce, grid, block, lds, stream, ctr, consumed, report, drop );
}
void rtk_next_bounce_pre( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, int consumed )
{
	hipLaunchKernelGGL( k_next_bounce_pre, grid, block, lds, stream, ctr, consumed );
}
void rtk_next_bounce_post( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, int consumed, uint32_t* report )
{
	hipLaunchKernelGGL( k_next_bounce_post, grid, block, lds, stream, ctr, consumed, report );
}
void rtk_accumulate( dim3 grid, dim3 block, size_t lds, hipStream_t stream, float4* acc, const float4* E, const long long* Es, int WH, int S )
{
	hipLaunchKernelGGL( k_accumulate, grid, block, lds, stream, acc, E, Es, WH, S );
}
void rtk_finalize( dim3 grid, dim3 block, size_t lds, hipStream_t stream, const float4* acc, uint32_t* out, int n, int frames )
{
	hipLaunchKernelGGL( k_finalize, grid, block, lds, stream, acc, out, n, frames );
}
void rtk_gamma( dim3 grid, dim3 block, size_t lds, hipStream_t stream, uint32_t* px, int n, const uint32_t* lut )
{
	hipLaunchKernelGGL( k_gamma, grid, block, lds, stream, px, n, lut );
}
void rtk_chroma( dim3 grid, dim3 block, size_t lds, hipStream_t stream, const uint32_t* src, uint32_t* dst, int W, int H, float intensity )
{
	hipLaunchKernelGGL( k_chroma, grid, block, lds, stream, src, dst, W, H, intensity );
}
void rtk_vignette( dim3 grid, dim3 block, size_t lds, hipStream_t stream, uint32_t* px, int W, int H, float intensity )
{
	hipLaunchKernelGGL( k_vignette, grid, block, lds, stream, px, W, H, intensity );
}
void rtk_trace_rays( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int* obj_out, float* t_out )
{
	if (qbvh) hipLaunchKernelGGL( ( k_trace_rays<true> ), grid, block, lds, stream, s, n, O3, D3, tmax, any_hit, obj_out, t_out );
	else if (!qbvh) hipLaunchKernelGGL( ( k_trace_rays<false> ), grid, block, lds, stream, s, n, O3, D3, tmax, any_hit, obj_out, t_out );
}
void rtk_copy_hits( dim3 grid, dim3 block, size_t lds, hipStream_t stream, PathState st, int n, int* obj_out, float* t_out )
{
	hipLaunchKernelGGL( k_copy_hits, grid, block, lds, stream, st, n, obj_out, t_out );
}
void rtk_rays_begin( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const float* O3, const float* D3, const float* tmax, int any_hit, PathState st, ShadowQueue sq, float4* Es, uint32_t* wq, int* obj_out, float* t_out, DevCounters* ctr )
{
	hipLaunchKernelGGL( k_rays_begin, grid, block, lds, stream, s, n, O3, D3, tmax, any_hit, st, sq, Es, wq, obj_out, t_out, ctr );
}
void rtk_rays_any_result( dim3 grid, dim3 block, size_t lds, hipStream_t stream, const long long* Es, int n, int* obj_out )
{
	hipLaunchKernelGGL( k_rays_any_result, grid, block, lds, stream, Es, n, obj_out );
}
void rtk_isect_one( dim3 grid, dim3 block, size_t lds, hipStream_t stream, OnePrim p, int n, const float* O3, const float* D3, const float* tin, int* hit, float* tout )
{
	hipLaunchKernelGGL( k_isect_one, grid, block, lds, stream, p, n, O3, D3, tin, hit, tout );
}
void rtk_hit_attribs( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const int* obj, const float* I3, const float* wo3, float* N3, float* alb3, int* mat )
{
	hipLaunchKernelGGL( k_hit_attribs, grid, block, lds, stream, s, n, obj, I3, wo3, N3, alb3, mat );
}
void rtk_whitted( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, CamProp cam, int W, int H, float invW, float invH, int mode, int max_depth, int aa, WFrame* frames, float4* acc )
{
	if (qbvh) hipLaunchKernelGGL( ( k_whitted<true> ), grid, block, lds, stream, s, cam, W, H, invW, invH, mode, max_depth, aa, frames, acc );
	else if (!qbvh) hipLaunchKernelGGL( ( k_whitted<false> ), grid, block, lds, stream, s, cam, W, H, invW, invH, mode, max_depth, aa, frames, acc );
}
