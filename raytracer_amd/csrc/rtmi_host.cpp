// rtmi_host.cpp -- librtmi.so: the C ABI of include/rtmi.h.  Host side only (plain C++): contexts, batch slots, streams and
// events, the two lanes and their tickets, the options, scene upload and repacking, RCCL binding.  The kernels and their
// launchers are rt_kernels.hip (rt_shared.h).
//
// Build: g++ -O2 -ffp-contract=off -c (this file), then hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared with
// rt_kernels.hip (see raytracer_amd/build.py)
#include "rt_shared.h"
#include "../../include/rtmi.h"
#include "../host/qbvh.h"
#include <string>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <condition_variable>
#include <dlfcn.h>
#include <chrono>

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
	hipEvent_t ev_q0 = 0, ev_q1 = 0;     // option qbvh: k_qbvh_fix of a closest-hit walk runs on the side stream beside it (launch_trace)
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
	bool qbvh_on = false;        // option qbvh = 1: results follow the reference's default traversal, IntersectQBVH (rtmi_upload_scene); takes effect at the next upload
	bool wide_on = true;         // RTMI_WIDE=0: k_trace walks the pairs instead of the 4-wide nodes with quantised boxes (rt_device.h: wide_step)
	int refill = 16;        // refill a wave of k_trace when at least this many lanes are idle (8 / 24 / 32: within 1 %)
	int max_blocks = 1024;  // resident 256-thread blocks for the persistent kernels (set from LDS use at upload)
	uint32_t grid_closest[2] = { 0, 0 }, grid_any[2] = { 0, 0 }; // wgrid: path counts below which a walk launch keeps 3, then 2 blocks per CU resident (rtmi_create)
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
	int count_lag = 1;           // bounces after which a small batch's path counts are read (see render_batch)
	// small_paths: a batch with no more paths than this leaves the wavefront loop at once.  Renderer::Tick at 1280x720
	// (0.92 M paths per frame): 2^17 5.92 ms, 2^18 5.32, 2^19 4.91, 2^20 4.72 -- below a million paths a wavefront bounce is
	// launch latency (0.28 ms each) and the tail kernel's 32 bounces in one launch are cheaper; the 20-step bench does not care
	uint32_t small_paths = 1u << 20;
	int carry_bounce = 16;       // with carry: the wavefront loop runs at least this many bounces (paths are late from vertex 7; the bounces up to the 16th are large enough to be worth running here)
	uint32_t tail_paths = 1u << 18; // finish a batch with k_tail once no more than this many paths are alive (0 = never)
	int tail_lanes = 16;         // fewest paths per wave in the tail rounds after the first (power of two <= 64)
	int sblock = 512;            // threads per block of k_primary / k_shade: blocks of 16 waves wait for a CU with 4 free
	                             // slots on every SIMD, which the long-lived tail waves of earlier batches rarely leave
	int tail_waves = 16;         // resident k_tail waves per CU
	int tail_block = 4;          // waves per k_tail workgroup
	uint32_t tail_dense = 4096;  // k_tail: 64 paths per wave above tail_dense live paths, else tail_lanes
	bool log_bounces = false;   // option log_bounces: print the path count per bounce (diagnostics)
	int budget = 32;            // tail steps after which a still-walking ray is parked (0 = never)
	int batch_opt = 0;          // option batch: samples per render batch (0 = as many as 2^RTMI_SLOT_BITS paths allow)
	int fit_samples = 0;        // > 0: the largest batch whose queues were found to fit in device memory (rtmi_render)
	int blocks_per_cu_opt = 0;  // resident walk blocks per CU (0 = min(4, what the LDS stack allows); 3 beside a second lane, set_lanes)
	int any_blocks_per_cu = 0;  // ... of the any-hit walk (0 = the same; 2 beside a second lane: see the option lanes)
	bool tail_paths_set = false;
	bool check_launches = false; // option check_launches: synchronise and check after every bounce (localises a fault; slow)
	int hook_kernel = 0;        // option hook_kernel: rtmi_trace_rays runs 0 = k_trace_rays (one ray per lane over the reference's pairs),
	                            // 1 = the render path's own queues and persistent walk (k_trace: 4-wide nodes where built),
	                            // 2 = the end-of-render walker, one ray per wave (k_rays_wave = k_tail_wave's device code)
	int last_round_bounces = 1 << 16; // see launch_tail_rounds
	// option preview_reads: rtmi_finalize shows the image as far as it has got instead of finishing every path first -- the frame
	// loop's display path (Renderer::Tick = one sample and a finalize per frame).  The paths of a frame that are still bouncing
	// when it is shown (a few thousand of 2 M on the 1 M-triangle scene, inside the mirror and glass knots for thousands of
	// bounces) ride the next frames' batches; every other read, rtmi_flush and a camera change join as before.
	bool preview_reads = false;
	bool tail_team = true;      // option tail_team: draining rounds with no more than team_below live paths run k_tail_team (four lanes per path)
	uint32_t team_below = 1u << 16; // Measured at the end of an 8 x 64-spp render (108 k ... 6 k paths alive over the rounds),
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
	bool prof_events = false; int count_nodes = 0; // count_nodes: 1 = the counting kernels walk the reference's pairs, 2 = the 4-wide nodes the timed kernels walk
	rtmi_stats stats = {};
	struct TimedEv { hipEvent_t a, b; int cls; };
	std::vector<TimedEv> timed;
	std::vector<hipEvent_t> ev_pool;
	hipEvent_t prof_base = 0;    // recorded when event timing is switched on: launch intervals are taken relative to it
	// [start, end) of the timed launches since rtmi_reset_stats, ms after prof_base, per class: 0 closest-hit walk, 1 k_shade / k_primary,
	// 2 any-hit walk, 3 other, 4 the walk windows.  Two lanes' launches of one kernel overlap: the stats give sum and union.
	std::vector<std::pair<float, float>> intervals[5];
};


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
			if (t.cls == 0) c->stats.ms_extend += ms, c->stats.n_extend_launches++;
			else if (t.cls == 1) c->stats.ms_shade += ms, c->stats.n_shade_launches++;
			else if (t.cls == 2) c->stats.ms_connect += ms, c->stats.n_connect_launches++;
			else if (t.cls == 4) c->stats.ms_walk += ms, c->stats.n_walk_windows++;
			else c->stats.ms_other += ms;
			float t0 = 0;
			if (t.cls >= 0 && t.cls <= 4 && t.cls != 3 && c->prof_base && hipEventElapsedTime( &t0, c->prof_base, t.a ) == hipSuccess) c->intervals[t.cls].push_back( { t0, t0 + ms } );
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

static double now_ms() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
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
		rtk_tail_decide( dim3( 1 ), dim3( 1 ), 0, ts, ctr, &ctr->n_state[cur], c->draining_dev, team_below, wave_below );
		rtk_tail( c->ds.qnodes != 0, dim3( blocks ), dim3( 64 * wpb ), lds_words * sizeof( uint32_t ), ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit,
			round == 0 ? 0u : c->tail_dense, c->tail_lanes, lds_words, team_below, ctr, stop_late );
		if (team_below > wave_below)
		{
			// (the draining kernels have the chip to themselves: 16 waves per CU whatever tail_waves says about tails beside a bulk)
			const uint32_t twaves = std::max( 1u, std::min( drain_waves, std::min( upper, team_below ) ) ), tblocks = ( twaves + 3 ) / 4;
			const size_t tlds = (size_t)c->walk_depth * 64u * sizeof( uint32_t ); // 4 waves x 16 teams, one column each
			rtk_tail_team( c->ds.qnodes != 0, dim3( tblocks ), dim3( 256 ), tlds, ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit, ctr );
		}
		if (wave_below)
		{
			// one wave per path, as many waves as the round can have paths (a wave that finds the pool empty leaves at once)
			// (the rounds of 4096 bounces and more are for a handful of paths: fewer waves, cheaper to dispatch when they find nothing)
			const uint32_t wwaves = std::max( 1u, std::min( round >= 7 ? 1024u : drain_waves, std::min( upper, wave_below ) ) ), wblocks = ( wwaves + WAVE_WG - 1 ) / WAVE_WG;
			rtk_tail_wave( c->ds.qnodes != 0, dim3( wblocks ), dim3( 64 * WAVE_WG ), 0, ts, c->ds, st[cur], st[cur ^ 1], sinks, &ctr->n_state[cur], &ctr->n_state[cur ^ 1], limit, ctr );
		}
		if (c->log_bounces && drain) // diagnostics: the survivors of every draining round (synchronises)
		{
			uint32_t left = 0;
			(void)hipStreamSynchronize( ts );
			(void)hipMemcpy( &left, &ctr->n_state[cur ^ 1], 4, hipMemcpyDeviceToHost );
			fprintf( stderr, "rtmi: drain round %d (%d bounces): %u paths survive (lane %d, host clock %.1f ms)\n", round, limit, left, c->is_child ? 1 : 0, std::fmod( now_ms(), 1e6 ) );
		}
		rtk_next_bounce( dim3( 1 ), dim3( 1 ), 0, ts, ctr, cur, (uint32_t*)0, round == 11 ? 1 : 0 );
		cur ^= 1;
	}
	return cur;
}

extern "C" { static int render_batch( rtmi_ctx* c, BatchSlot& b, int sample_begin, int S, int stride, uint64_t ticket ); } // (defined among the entry points below)

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
		rtk_fold_late( dim3( nblocks( (size_t)c->W * c->H ) ), dim3( BLOCK ), 0, c->stream, c->accum, c->late, c->odd, c->W * c->H );
		HIPCHK( hipGetLastError() );
		return RTMI_OK; // (carry_pending stays: the next full join finishes the carried paths)
	}
	if (pending && c->has_scene)
	{
		// The late paths that the last batches handed on have no batch to ride with: they get a batch of their own, without
		// camera rays (S = 0): wavefront bounces while they are many, then k_tail.  One per lane, enqueued side by side.
		// (a lane whose next slot has no queues -- freed by a resize that failed, never sized -- gets the smallest ones first: its carried
		// paths are finished by this call or the call fails, they are never silently left out of the image)
		const bool mine = c->carry_on && c->paths_carried, kids = k && k->carry_on && k->paths_carried;
		int rk = RTMI_OK, r = RTMI_OK;
		if (mine && !c->slots[c->next_slot].E && ( r = ensure_batch( c, c->slots[c->next_slot], 1 ) )) return r;
		if (kids && !k->slots[k->next_slot].E && ( rk = ensure_batch( k, k->slots[k->next_slot], 1 ) )) { c->err = k->err; return rk; }
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
		rtk_fold_late( dim3( nblocks( (size_t)c->W * c->H ) ), dim3( BLOCK ), 0, c->stream, c->accum, c->late, c->odd, c->W * c->H );
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
	ok = ok && hipEventCreateWithFlags( &b.ev_q0, hipEventDisableTiming ) == hipSuccess && hipEventCreateWithFlags( &b.ev_q1, hipEventDisableTiming ) == hipSuccess;
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
	{ "log_bounces", 0, 1 }, { "overlap", 0, 1 }, { "wide", 0, 1 }, { "plane_pairs", 0, 1 }, { "qbvh", 0, 1 }, { "carry", 0, 1 }, { "slots", 1, MAX_SLOTS }, { "tail_paths", 0, 1ll << 32 },
	{ "carry_cap", 1, 1ll << 24 }, { "left_cap", 1, 1ll << 20 }, { "batch", 0, 1 << 20 }, { "check_launches", 0, 1 }, { "hook_kernel", 0, 2 }, { "tail_team", 0, 1 },
	{ "tail_wave", 0, 1 }, { "wave_below", 0, 1 << 18 }, { "carry_min", 0, 1 << RTMI_SLOT_BITS }, { "lanes", 1, 2 }, { "last_round_bounces", 0, 1 << 30 }, { "preview_reads", 0, 1 },
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
	else if (is( "qbvh" )) c->qbvh_on = v != 0;
	else if (is( "carry" )) { c->carry_on = v != 0; if (!c->tail_paths_set) c->tail_paths = c->carry_on ? 1u << 22 : c->small_paths; }
	else if (is( "tail_paths" )) c->tail_paths = (uint32_t)std::min<long long>( v, 0xFFFFFFFFll ), c->tail_paths_set = true;
	else if (is( "batch" )) c->batch_opt = (int)v;
	else if (is( "check_launches" )) c->check_launches = v != 0;
	else if (is( "hook_kernel" )) c->hook_kernel = (int)v;
	else if (is( "tail_team" )) c->tail_team = v != 0;
	else if (is( "tail_wave" )) c->tail_wave = v != 0;
	else if (is( "wave_below" )) c->wave_below = (uint32_t)v;
	else if (is( "carry_min" )) c->carry_min = (uint32_t)v;
	else if (is( "preview_reads" )) c->preview_reads = v != 0;
	else if (is( "last_round_bounces" )) c->last_round_bounces = (int)v;
	else if (is( "lanes" ))
	{
		c->lanes = (int)v;
		// beside another lane's kernels 3 resident walk blocks per CU and 8 tail waves per CU do better than 4 and 16
		// (20 bench steps: 10 358 -> 10 623 Mrays/s; alone, 4 and 16 are the optimum)
		// and the shadow walk 2: it runs beside this lane's next closest-hit walk AND the other lane's kernels, and the fewer of its waves sit
		// on a SIMD the sooner the closest-hit walk and k_shade, which the bounce waits for, get theirs (closest / any blocks per CU, 20 steps:
		// 3 / 3 12 046 Mrays/s, 3 / 2 12 309, 2 / 2 12 089, 2 / 3 11 893, 4 / 2 11 983, 3 / 1 11 984; 64 steps: 13 152 -> 13 535; one lane
		// stays at 4 / 4: 10 971 against 10 860 with 4 / 3 and 10 484 with 4 / 2)
		c->blocks_per_cu_opt = c->lanes == 2 ? 3 : 0, c->any_blocks_per_cu = c->lanes == 2 ? 2 : 0, c->tail_waves = c->lanes == 2 ? 8 : 16;
		if (live) { set_max_blocks( c ); int r = set_lanes( c ); if (r) return r; }
	}
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
	if (is( "log_bounces" )) return c->log_bounces;
	if (is( "overlap" )) return c->overlap_on;
	if (is( "wide" )) return c->wide_on;
	if (is( "plane_pairs" )) return c->plane_pairs_on;
	if (is( "qbvh" )) return c->qbvh_on;
	if (is( "carry" )) return c->carry_on;
	if (is( "slots" )) return c->n_slots;
	if (is( "tail_paths" )) return c->tail_paths;
	if (is( "carry_cap" )) return c->carry_cap;
	if (is( "left_cap" )) return c->left_cap;
	if (is( "batch" )) return c->batch_opt;
	if (is( "check_launches" )) return c->check_launches;
	if (is( "hook_kernel" )) return c->hook_kernel;
	if (is( "tail_team" )) return c->tail_team;
	if (is( "tail_wave" )) return c->tail_wave;
	if (is( "wave_below" )) return c->wave_below;
	if (is( "carry_min" )) return c->carry_min;
	if (is( "lanes" )) return c->lanes;
	if (is( "last_round_bounces" )) return c->last_round_bounces;
	if (is( "preview_reads" )) return c->preview_reads;
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
	k->overlap_on = c->overlap_on && k->side != 0, k->wide_on = c->wide_on, k->plane_pairs_on = c->plane_pairs_on, k->qbvh_on = c->qbvh_on, k->refill = c->refill, k->max_blocks = c->max_blocks;
	for (int i = 0; i < 2; i++) k->grid_closest[i] = c->grid_closest[i], k->grid_any[i] = c->grid_any[i];
	k->cam = c->cam, k->seeds = c->seeds, k->accum = c->accum, k->late = c->late, k->odd = c->odd;
	k->carry_on = c->carry_on, k->carry_min = c->carry_min, k->small_paths = c->small_paths, k->count_lag = c->count_lag, k->carry_bounce = c->carry_bounce, k->tail_paths = c->tail_paths;
	k->tail_lanes = c->tail_lanes, k->sblock = c->sblock, k->tail_waves = c->tail_waves, k->tail_block = c->tail_block, k->tail_dense = c->tail_dense;
	k->log_bounces = c->log_bounces, k->budget = c->budget, k->batch_opt = c->batch_opt, k->blocks_per_cu_opt = c->blocks_per_cu_opt, k->any_blocks_per_cu = c->any_blocks_per_cu, k->check_launches = c->check_launches;
	k->tail_team = c->tail_team, k->team_below = c->team_below, k->tail_wave = c->tail_wave, k->wave_below = c->wave_below;
	k->prof_events = c->prof_events, k->count_nodes = c->count_nodes, k->prof_base = c->prof_base, k->last_round_bounces = c->last_round_bounces, k->preview_reads = c->preview_reads;
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
	a.dropped_paths += b.dropped_paths, a.extend_wide_nodes += b.extend_wide_nodes, a.extend_confirms += b.extend_confirms, a.shadow_wide_nodes += b.shadow_wide_nodes, a.shadow_confirms += b.shadow_confirms;
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
	if (rtk_scratch_address( device_id, &c->noalloc_scratch ) != hipSuccess) { (void)hipStreamDestroy( c->own_stream ); return bail( RTMI_ERR_HIP ); }
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
		// (One lane unless the caller asks for two -- option lanes, or RTMI_LANES in the environment.  Two lanes pay only when their
		// streams do not share hardware queues: 2 x (main, side, 4 tails) + the caller's, and HIP gives a process GPU_MAX_HW_QUEUES
		// of them, 4 unless the process raised it BEFORE the runtime started, which this library cannot see or check: measured at
		// 20 bench steps, 2 lanes on 16 queues +8.5 % rays/s, on 4 queues -2 %; and a second lane doubles the queue memory.
		// bench.py sets the variable before it imports torch and asks for two lanes.)
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
		if (b.ev_q0) (void)hipEventDestroy( b.ev_q0 );
		if (b.ev_q1) (void)hipEventDestroy( b.ev_q1 );
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
	// the diagnostics take effect at once -- also on the second lane -- so that the END of a render can be looked at
	if (!strcmp( key, "log_bounces" )) { int r = apply_option( c, key, value, true ); if (!r && c->child) c->child->log_bounces = c->log_bounces; return r; }
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
	int maxDepth = 0, wideNeed = 0, qdepthNeed = 0;
	ds.n_quads = d->n_quads, ds.n_spheres = d->n_spheres, ds.n_cubes = d->n_cubes, ds.n_planes = d->n_planes, ds.n_tris = d->n_triangles, ds.n_bvh_prims = nbvh;
	if (nbvh > 0)
	{
		const int nn = d->n_nodes;
		if (( nn - 3 ) % 2 != 0) return fail( c, RTMI_ERR_BVH, "n_nodes must be 3 + 2*pairs" );
		const int npairs = ( nn - 3 ) / 2;
		// Option qbvh: the reference's DEFAULT traversal (Scene::useQBVH = true, scene.h:585) differs from its binary one in two
		// things a result can depend on: the order in which leaves are visited (the first of equally distant primitives wins,
		// primitive.h:68) and, for rays with an infinite reciprocal direction component, which boxes are tested and how the SSE
		// min / max treat the NaN of 0 * inf (bvh.h:465-486).  So with the option on the library rebuilds the
		// reference's QBVH from the node array (BuildQBVH / CollapseBVH, bvh.h:75-249), (1) lays the leaf-ordered primitive
		// records out in the QBVH's visiting order, so that the (t, position) rule of every walker breaks ties as IntersectQBVH
		// does, and (2) uploads the QBVH nodes: rays with an infinite reciprocal component walk them with the SSE semantics
		// (rt_device.h: traverse_qbvh), all others walk the 4-wide / pair nodes as always (same boxes bar the order of tests).
		std::vector<int> newpos; // reference position k (primitiveIdx[k]) -> device position; identity without the option
		std::vector<rtmi::QbvhNode> qn;
		int qdepth = 0;
		if (c->qbvh_on)
		{
			// BuildQbvh recurses over the node array: only over one that is a tree (checked in full further down, with messages)
			std::vector<uint8_t> seen( (size_t)nn, 0 );
			std::vector<int> stk = { 2 };
			bool tree = true;
			size_t covered = 0;
			while (!stk.empty() && tree)
			{
				const int i = stk.back();
				stk.pop_back();
				if (i < 2 || i >= nn || seen[i]) { tree = false; break; }
				seen[i] = 1;
				const rtmi_bvh_node& n = d->nodes[i];
				if (n.primitiveCount > 0) { if (n.leftFirst < 0 || (long long)n.leftFirst + n.primitiveCount > nbvh) tree = false; else covered += (size_t)n.primitiveCount; }
				else if (n.primitiveCount < 0 || n.leftFirst < 3 || n.leftFirst + 1 >= nn) tree = false;
				else stk.push_back( n.leftFirst ), stk.push_back( n.leftFirst + 1 );
			}
			if (tree && covered == (size_t)nbvh && rtmi::BuildQbvh( d->nodes, nn, qn ))
			{
				// visiting order of IntersectQBVH (bvh.h:607-638): children in slot order, depth first
				newpos.assign( (size_t)nbvh, -1 );
				int next = 0;
				std::function<void( int, int )> visit = [&]( int node, int depth ) {
					qdepth = std::max( qdepth, depth );
					for (int sl = 0; sl < 4; sl++)
					{
						rtmi::QbvhNode& q = qn[node];
						if (q.child[sl] == -1) return;
						if (q.count[sl] > 0)
						{
							const int first = q.child[sl];
							for (int j = 0; j < q.count[sl]; j++) newpos[(size_t)first + j] = next + j;
							q.child[sl] = next, next += q.count[sl];
						}
						else visit( q.child[sl], depth + 1 );
					} };
				visit( qn[0].child[1], 1 );
				if (next != nbvh) { qn.clear(), newpos.clear(), qdepth = 0; } // (cannot happen for a tree that covers every position once)
			}
			else qn.clear();
		}
		auto device_pos = [&]( int k ) { return newpos.empty() ? k : newpos[(size_t)k]; };
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
				ref = chunked( device_pos( n.leftFirst ), n.primitiveCount, n ); // (a leaf's primitives stay together and in order)
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
			const size_t at = (size_t)device_pos( k );
			prims[3 * at] = a, prims[3 * at + 1] = b, prims[3 * at + 2] = cc;
		}
		if (!qn.empty())
		{
			float4* dqn = 0;
			static_assert( sizeof( rtmi::QbvhNode ) == 128, "SIMD_BVH_Node layout (bvh.h:18-27)" );
			if (( r = dev_upload( c, &dqn, (const float4*)qn.data(), qn.size() * 8, &allocs ) )) return bail( r );
			ds.qnodes = dqn, ds.qroot = (uint32_t)qn[0].child[1];
			if (c->log_bounces) fprintf( stderr, "rtmi: the reference's QBVH: %zu nodes, depth %d\n", qn.size(), qdepth );
			qdepthNeed = qdepth + 2;
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
	c->stack_depth = std::max( c->stack_depth, qdepthNeed ), c->walk_depth = std::max( c->walk_depth, qdepthNeed ); // (traverse_qbvh: one stack entry per level)
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
	int blocks = std::min( c->max_blocks, c->n_cus * per_cu );
	if (any && c->any_blocks_per_cu > 0) blocks = std::min( blocks, c->n_cus * c->any_blocks_per_cu );
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
	a.rec_cap = b.rec_cap, a.rec_stride = b.rec_stride, a.refill = c->refill, a.budget = budget, a.count_wide = c->count_nodes == 2;
	ScopedTimer t( c, any ? 2 : 0, s );
	// option qbvh: the few rays that follow the reference's QBVH (flagged queue entries, which k_trace retires untouched) are walked
	// by k_qbvh_fix over the same queue.  They are few but long (the camera rays' centre row: thousands of nodes each under the NaN
	// semantics, 3.7 ms for a batch's first bounce), so beside a closest-hit walk the kernel runs on the side stream, not behind it.
	const dim3 qgrid( (unsigned)std::min<size_t>( (size_t)c->n_cus * 4, ( (size_t)upper + BLOCK - 1 ) / BLOCK + 1 ) );
	const bool beside = c->ds.qnodes && !any && c->overlap_on && c->side && s != c->side && b.ev_q0;
	if (beside)
	{
		(void)hipEventRecord( b.ev_q0, s );
		(void)hipStreamWaitEvent( c->side, b.ev_q0, 0 );
		rtk_qbvh_fix( false, qgrid, dim3( BLOCK ), lds, c->side, c->ds, a );
		(void)hipEventRecord( b.ev_q1, c->side );
	}
	rtk_trace( any, c->count_nodes != 0, wgrid( c, any, upper ), dim3( BLOCK ), lds, s, c->ds, a, ctr );
	if (beside) (void)hipStreamWaitEvent( s, b.ev_q1, 0 );
	else if (c->ds.qnodes) rtk_qbvh_fix( any, qgrid, dim3( BLOCK ), lds, s, c->ds, a );
}
static void launch_primary( rtmi_ctx* c, BatchSlot& b, uint32_t N, int sample_begin, int stride )
{
	ScopedTimer t( c, 3, c->stream ); // (class "other": ms_shade is k_shade alone)
	rtk_primary( dim3( nsblocks( c, N ) ), dim3( c->sblock ), 0, c->stream, c->ds, b.st[0], b.E, (float4*)b.Es, N, c->cam, c->W, c->H, 1.0f / c->W, 1.0f / c->H, // camera.h:31-32
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
	if (c->carry_on && c->paths_carried)
	{
		// the late paths that earlier batches handed on bounce along with this batch's (entries N, N+1, ...): the carry buffer
		// (the previous batch's), and what outlived the first tail round of the batch that used this slot before -- of all
		// slots when this is the last batch
		auto take = [&]( PathState src, uint32_t* n_src, uint32_t cap ) {
			rtk_carry_in( dim3( ( cap + 511 ) / 512 ), dim3( 512 ), 0, s, c->ds, src, n_src, cap, b.st[0], &ctr->n_state[0], b.wq[0], ctr );
			rtk_carry_commit( dim3( 1 ), dim3( 1 ), 0, s, &ctr->n_state[0], n_src, cap );
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
			rtk_shade( dim3( nsblocks( c, upper ) ), dim3( c->sblock ), 0, s, c->ds, b.st[cur], b.st[cur ^ 1], b.sq[cur], make_sinks( c, b ), &ctr->n_state[cur], &ctr->n_state[cur ^ 1],
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
			rtk_next_bounce_post( dim3( 1 ), dim3( 1 ), 0, c->side, ctr, cur, b.host_counts_dev + slot );
			HIPCHK( hipEventRecord( b.count_ev[slot], c->side ) );
			HIPCHK( hipEventRecord( b.ev_any, c->side ) );
			side_busy = true;
			rtk_next_bounce_pre( dim3( 1 ), dim3( 1 ), 0, s, ctr, cur );
			if (c->prof_events) { window = get_event( c ); if (window) (void)hipEventRecord( window, s ); }
		}
		else
		{
			launch_trace( c, b, true, cur, upper, budget ); // this bounce's shadow rays (at most `upper` + the carried ones)
			rtk_next_bounce( dim3( 1 ), dim3( 1 ), 0, s, ctr, cur, b.host_counts_dev + slot, 0 );
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
			if (c->log_bounces && ( bounce < 64 || ( bounce & ( bounce - 1 ) ) == 0 )) fprintf( stderr, "rtmi: bounce %d starts with %u paths (lane %d, host clock %.1f ms)\n", bounce + 1 - lag, b.host_counts[old], c->is_child ? 1 : 0, std::fmod( now_ms(), 1e6 ) );
		}
		if (upper == 0) break;
	}
	if (side_busy) { HIPCHK( hipStreamWaitEvent( s, b.ev_any, 0 ) ); side_busy = false; }
	close_window();
	if (tail && c->carry_on && !last)
	{
		c->paths_carried = true;
		// late paths -> carry buffer (they join the next batch, or join_batches finishes them); the others -> the other state set
		rtk_carry_out( dim3( ( upper + 511 ) / 512 ), dim3( 512 ), 0, s, b.st[cur], &ctr->n_state[cur], b.st[cur ^ 1], &ctr->n_state[cur ^ 1],
			c->carry[0], &c->carry_ctr->n_state[0], c->carry_cap );
		if (c->log_bounces)
		{
			uint32_t rest = 0, carried = 0;
			(void)hipStreamSynchronize( s );
			(void)hipMemcpy( &rest, &ctr->n_state[cur ^ 1], 4, hipMemcpyDeviceToHost ), (void)hipMemcpy( &carried, &c->carry_ctr->n_state[0], 4, hipMemcpyDeviceToHost );
			fprintf( stderr, "rtmi: hand-over after %d bounces: %u paths carried on (buffer %u), %u stay for this batch's tail\n", (int)c->stats.bounce_launches, carried, c->carry_cap, rest );
		}
		rtk_next_bounce( dim3( 1 ), dim3( 1 ), 0, s, ctr, cur, (uint32_t*)0, 0 );
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
			// (a path leaves that round as soon as it is late: k_tail, stop_late)
			cur = launch_tail_rounds( c, ts, b.st, cur, ctr, make_sinks( c, b ), upper, 0, 0, false, 1 );
			rtk_carry_out( dim3( ( upper + 511 ) / 512 ), dim3( 512 ), 0, ts, b.st[cur], &ctr->n_state[cur], b.st[cur ^ 1], &ctr->n_state[cur ^ 1], b.left, b.n_left, c->left_cap );
			rtk_next_bounce( dim3( 1 ), dim3( 1 ), 0, ts, ctr, cur, (uint32_t*)0, 0 );
			cur ^= 1;
			launch_tail_rounds( c, ts, b.st, cur, ctr, make_sinks( c, b ), 4096, 1, 11 ); // normally nothing: a small persistent grid
		}
		else launch_tail_rounds( c, ts, b.st, cur, ctr, make_sinks( c, b ), upper, 0, 11, last );
		c->stats.bounce_launches++;
	}
	turn.enter(); // accumulate in batch order: the previous batch (of either lane) has recorded its event
	if (c->chain->last_acc) HIPCHK( hipStreamWaitEvent( ts, c->chain->last_acc, 0 ) );
	if (S != 0)
	{
		ScopedTimer t( c, 3, ts );
		rtk_accumulate( dim3( nblocks( WH ) ), dim3( BLOCK ), 0, ts, c->accum, b.E, b.Es, WH, S );
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
		for (int k = 0; k < 5; k++)
		{
			c->intervals[k].insert( c->intervals[k].end(), c->child->intervals[k].begin(), c->child->intervals[k].end() );
			c->child->intervals[k].clear();
		}
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
		c->stats.dropped_paths += h.dropped, c->stats.extend_wide_nodes += h.wide_nodes[0], c->stats.shadow_wide_nodes += h.wide_nodes[1], c->stats.extend_confirms += h.confirms[0], c->stats.shadow_confirms += h.confirms[1];
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
		rtk_whitted( c->ds.qnodes != 0, dim3( nblocks( WH ) ), dim3( BLOCK ), stack_bytes( c ), c->stream, c->ds, c->cam, c->W, c->H, 1.0f / c->W, 1.0f / c->H, mode, max_depth,
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
	rtk_copy_hits( dim3( nblocks( WH ) ), dim3( BLOCK ), 0, c->stream, b.st[0], WH, dobj, dt );
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
		rtk_trace_rays( c->ds.qnodes != 0, dim3( nblocks( n ) ), dim3( BLOCK ), stack_bytes( c ), c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
		HIPCHK( hipGetLastError() );
		return RTMI_OK;
	}
	if (c->hook_kernel == 2) // the end-of-render walker: one ray per wave (k_tail_wave's device code)
	{
		const int blocks = std::max( 1, std::min( ( n + 3 ) / 4, c->n_cus * 4 ) );
		rtk_rays_wave( c->ds.qnodes != 0, dim3( blocks ), dim3( 256 ), 0, c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
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
	rtk_rays_begin( dim3( nsblocks( c, (size_t)n ) ), dim3( c->sblock ), 0, c->stream, c->ds, n, dO, dD, dT, any_hit, b.st[0], b.sq[0], (float4*)b.Es, b.wq[0], dobj, dt, b.ctr );
	launch_trace( c, b, any_hit != 0, 0, (uint32_t)n, 0 ); // no step budget: every ray finishes in this launch
	if (any_hit) rtk_rays_any_result( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, b.Es, n, dobj );
	else rtk_copy_hits( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, b.st[0], n, dobj, dt );
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

int rtmi_intersect_primitive( rtmi_ctx* c, int kind, const float* data, int n, const float* O3, const float* D3, const float* t_in, int32_t* hit_out, float* t_out )
{
	if (!c || kind < 0 || kind > 4 || !data || n < 0 || ( n && ( !O3 || !D3 || !t_in || !hit_out || !t_out ) )) return RTMI_ERR_INVALID;
	if (n == 0) return RTMI_OK;
	HIPCHK( hipSetDevice( c->device ) );
	OnePrim p = {};
	p.kind = kind;
	auto f4 = [&]( int at, int k ) { float4 v = make_float4( 0, 0, 0, 0 ); memcpy( &v, data + at, (size_t)k * 4 ); return v; };
	if (kind == 0) p.d[0] = f4( 0, 3 ), p.d[1] = f4( 3, 3 ), p.d[2] = f4( 6, 3 );
	else if (kind == 1 || kind == 2) p.d[0] = f4( 0, 4 );
	else if (kind == 3) p.d[0] = f4( 0, 4 ), p.d[1] = f4( 4, 4 ), p.d[2] = f4( 8, 4 ), p.d[3] = f4( 12, 1 );
	else p.d[0] = f4( 0, 4 ), p.d[1] = f4( 4, 4 ), p.d[2] = f4( 8, 4 ), p.d[3] = f4( 12, 3 ), p.d[4] = f4( 15, 3 );
	int r = hook_reserve( c, (size_t)n * 9 * 4 );
	if (r) return r;
	float* dO = (float*)c->hook_buf, *dD = dO + 3 * (size_t)n, *dT = dD + 3 * (size_t)n, *dt = dT + n;
	int* dhit = (int*)( dt + n );
	HIPCHK( hipMemcpyAsync( dO, O3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dD, D3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dT, t_in, (size_t)n * 4, hipMemcpyHostToDevice, c->stream ) );
	rtk_isect_one( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, p, n, dO, dD, dT, dhit, dt );
	HIPCHK( hipGetLastError() );
	hipError_t e1 = hipMemcpyAsync( hit_out, dhit, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e2 = hipMemcpyAsync( t_out, dt, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e3 = hipStreamSynchronize( c->stream );
	HIPCHK( e1 ); HIPCHK( e2 ); HIPCHK( e3 );
	return RTMI_OK;
}
int rtmi_hit_attributes( rtmi_ctx* c, int n, const int32_t* obj, const float* I3, const float* wo3, float* N3_out, float* albedo3_out, int32_t* material_out )
{
	if (!c || n < 0 || ( n && ( !obj || !I3 || !wo3 || !N3_out || !albedo3_out || !material_out ) )) return RTMI_ERR_INVALID;
	if (!c->has_scene) return fail( c, RTMI_ERR_NO_SCENE, "rtmi_hit_attributes before rtmi_upload_scene" );
	if (n == 0) return RTMI_OK;
	const long long total = (long long)c->ds.n_quads + c->ds.n_spheres + c->ds.n_cubes + c->ds.n_planes + c->ds.n_tris;
	for (int i = 0; i < n; i++) if (obj[i] < 0 || obj[i] >= total) return fail( c, RTMI_ERR_INVALID, "rtmi_hit_attributes: object index out of range" );
	HIPCHK( hipSetDevice( c->device ) );
	int r = hook_reserve( c, (size_t)n * 14 * 4 );
	if (r) return r;
	float* dI = (float*)c->hook_buf, *dW = dI + 3 * (size_t)n, *dN = dW + 3 * (size_t)n, *dA = dN + 3 * (size_t)n;
	int* dobj = (int*)( dA + 3 * (size_t)n ), *dmat = dobj + n;
	HIPCHK( hipMemcpyAsync( dI, I3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dW, wo3, (size_t)n * 12, hipMemcpyHostToDevice, c->stream ) );
	HIPCHK( hipMemcpyAsync( dobj, obj, (size_t)n * 4, hipMemcpyHostToDevice, c->stream ) );
	rtk_hit_attribs( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, c->ds, n, dobj, dI, dW, dN, dA, dmat );
	HIPCHK( hipGetLastError() );
	hipError_t e1 = hipMemcpyAsync( N3_out, dN, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream );
	hipError_t e2 = hipMemcpyAsync( albedo3_out, dA, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream );
	hipError_t e4 = hipMemcpyAsync( material_out, dmat, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream );
	hipError_t e3 = hipStreamSynchronize( c->stream );
	HIPCHK( e1 ); HIPCHK( e2 ); HIPCHK( e4 ); HIPCHK( e3 );
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
		rtk_rays_wave( c->ds.qnodes != 0, dim3( std::max( 1, std::min( ( n + 3 ) / 4, c->n_cus * 4 ) ) ), dim3( 256 ), 0, c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
	else
		rtk_trace_rays( c->ds.qnodes != 0, dim3( nblocks( n ) ), dim3( BLOCK ), stack_bytes( c ), c->stream, c->ds, n, dO, dD, dT, any_hit, dobj, dt );
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
		rtk_gamma( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, px, n, c->gamma_lut );
	}
	if (fx->chroma_on)
	{
		rtk_chroma( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, px, tmp, W, H, fx->chroma_intensity );
		HIPCHK( hipMemcpyAsync( px, tmp, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream ) ); // postprocessing.h:75
	}
	if (fx->vignette_on) rtk_vignette( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, px, W, H, fx->vignette_intensity );
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
		rtk_finalize( dim3( nblocks( n ) ), dim3( BLOCK ), 0, c->stream, c->accum, c->pixels, n, frames );
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
	// Several handles may share one communicator (frames in flight, host/frames.h): the caller issues their reduces in the same order
	// on every rank, and here each one is also made to EXECUTE after the one before it on that communicator, whatever stream it is on
	static std::mutex chain_m;
	static std::unordered_map<void*, hipEvent_t> last_on_comm;
	std::lock_guard<std::mutex> lock( chain_m );
	hipEvent_t& last = last_on_comm[comm];
	if (last) HIPCHK( hipStreamWaitEvent( c->stream, last, 0 ) );
	else HIPCHK( hipEventCreateWithFlags( &last, hipEventDisableTiming ) );
	const int e = root >= 0 ? g_rccl.reduce( c->accum, c->accum, count, ncclFloat, ncclSum, root, comm, c->stream )
		: g_rccl.allreduce( c->accum, c->accum, count, ncclFloat, ncclSum, comm, c->stream );
	HIPCHK( hipEventRecord( last, c->stream ) );
	if (e != 0) return fail( c, RTMI_ERR_HIP, std::string( root >= 0 ? "ncclReduce: " : "ncclAllReduce: " ) + ( g_rccl.errstr ? g_rccl.errstr( e ) : "error" ) );
	return RTMI_OK;
}

int rtmi_set_whitted_lights( rtmi_ctx* c, const float pos[8], const float color[8], float intensity )
{
	if (!c || !pos || !color) return RTMI_ERR_INVALID;
	if (!c->has_scene) return fail( c, RTMI_ERR_NO_SCENE, "rtmi_set_whitted_lights before rtmi_upload_scene" );
	// the scene descriptor travels to k_whitted as a kernel argument: the next launch takes the new values, launches in flight keep theirs
	memcpy( c->ds.wl_pos, pos, 32 ), memcpy( c->ds.wl_color, color, 32 );
	c->ds.wl_intensity = intensity;
	return RTMI_OK;
}
int rtmi_memory_info( rtmi_ctx* c, uint64_t* free_bytes, uint64_t* total_bytes )
{
	if (!c || !free_bytes || !total_bytes) return RTMI_ERR_INVALID;
	HIPCHK( hipSetDevice( c->device ) );
	size_t f = 0, t = 0;
	HIPCHK( hipMemGetInfo( &f, &t ) );
	*free_bytes = f, *total_bytes = t;
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
	c->prof_events = ev != 0, c->count_nodes = cn < 0 ? 0 : cn > 2 ? 2 : cn;
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
	c->stats.ms_extend_union = union_ms( c->intervals[0] ), c->stats.ms_shade_union = union_ms( c->intervals[1] ), c->stats.ms_connect_union = union_ms( c->intervals[2] ),
		c->stats.ms_walk_union = union_ms( c->intervals[4] );
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
	for (auto& v : c->intervals) v.clear();
	return RTMI_OK;
}

} // extern "C"
