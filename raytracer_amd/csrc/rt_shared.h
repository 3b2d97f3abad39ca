// rt_shared.h -- what the host side of librtmi.so (rtmi_host.cpp: plain C++, no device code) and its kernels (rt_kernels.hip)
// have in common: the device scene, the queue layouts, the counters, and one launcher per kernel (rtk_<name>: grid, block,
// dynamic LDS bytes, stream, then the kernel's own arguments).  The launchers are all the host knows of the kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>
#include <stddef.h>

#define RT_DONE 0xFFFFFFFFu
#define RT_LEAF 0x80000000u
#define RT_LEAF_FIRST_BITS 25
#define RT_LEAF_FIRST_MASK ( ( 1u << RT_LEAF_FIRST_BITS ) - 1u )
#define RT_LEAF_MAX_COUNT 64

struct f3 { float x, y, z; };

// ---- device scene ------------------------------------------------------------------------------------------
struct DMaterial { float4 color; float4 absorption; float ior; int type; int pad0, pad1; }; // 48 B

struct DScene
{
	// BVH: pair p holds the two children of one interior node = two reference nodes (bvh.h:11-16) side by side,
	// 64-byte aligned.  float4 x4: {Lmin.xyz, Lmax.x} {Lmax.yz, Lref, Lcnt} {Rmin.xyz, Rmax.x} {Rmax.yz, Rref, Rcnt}
	const float4* __restrict__ pairs;
	// leaf-ordered primitive records (position k == primitiveIdx[k]), 48 B: triangle {p1, tag 0} {e1, objIdx} {e2, -}
	// sphere {pos, tag 1} {r2,0,0, objIdx} {-}
	const float4* __restrict__ prims;
	// 4-wide interior nodes with quantised boxes over the same leaves (wide_step below; null: not built), the exact box of
	// every leaf at [2 * first primitive position], and the wide root
	const float4* __restrict__ wide;
	const float4* __restrict__ leafbox;
	float root_min[3], root_max[3];
	uint32_t root_ref, wide_root;
	int n_quads, n_spheres, n_cubes, n_planes, n_tris, n_bvh_prims;
	const float4* __restrict__ quads;   // per quad 4 x float4: invT rows 0..2, {size,0,0,0}
	const float4* __restrict__ cubes;   // per cube 8 x float4: invM rows 0..2, b0, b1, M rows 0..2
	const float4* __restrict__ planes;  // (N, d)
	// shading
	const float4* __restrict__ primitives; // [total] scene.h:120-181 (xyz); w = index into `materials` (uint bits, set on upload): one gather per hit, not two
	const DMaterial* __restrict__ materials;
	const float* __restrict__ sphere_invr;
	const uint32_t* __restrict__ texture;
	int tex_w, tex_h;
	float4 light_c1, light_c2, light_c3, light_emission, light_N;
	float4 light_u, light_w; // normalize(c1 - c2), normalize(c3 - c2): the same for every sample, computed once on upload with the same operations
	float light_A, light_s;
	float4 wl_pos[2], wl_color[2];
	float wl_intensity;
	// quads[0] (the only quad the reference tests, scene.h:361) as kernel arguments: invT rows 0..2, {size,0,0,0}
	float4 quad0[4];
	// plane_pairs != 0: the scene has exactly the six planes +x, -x, +y, -y, +z, -z in this order (the reference's room,
	// scene.h:70-75); pair_d[k] = d of plane k.  See isect_planes.
	int plane_pairs;
	float pair_d[6];
	// option qbvh: the reference's own 4-wide QBVH (SIMD_BVH_Node, bvh.h:18-27: 8 x float4 per node -- bminx4, bmaxx4, bminy4,
	// bmaxy4, bminz4, bmaxz4, child[4], count[4]; leaf children hold device positions) for the rays whose reciprocal direction has
	// an infinite component (rt_device.h: traverse_qbvh); null: off.  qroot = the node IntersectQBVH starts at (bvh.h:603).
	const float4* __restrict__ qnodes;
	uint32_t qroot;
};

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
	unsigned long long wide_nodes[2], confirms[2]; // counting instantiation over the 4-wide nodes (closest / any): 64-byte records fetched, 32-byte leaf boxes fetched to confirm a hit
	unsigned long long dropped; // paths still alive after the last tail round (option last_round_bounces): ended there
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
#define SQ_INEXACT 0x10000000u      // shadow queue slot word / walk queue entry (WQ_INEXACT): option qbvh, a ray with an infinite reciprocal
#define WQ_INEXACT 0x40000000u      // direction component -- k_trace leaves it alone, k_qbvh_fix walks the reference's QBVH for it
#define META_SUSPENDED 0x20000000u // the closest-hit walk of this path was parked (k_trace); k_shade passes it through
#define RESUME_BIT 0x80000000u     // walk queue entry / shadow slot field: resume from a suspend record
#define REC_HEADER 6               // suspend record: cur, sp, k, h.obj, h.pos, -, then the stack

#define ES_SCALE 1099511627776.0f // 2^40

struct Sinks { float4* E; long long* Es; long long* late; float4* odd; uint32_t WH; };

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
	int count_wide;                 // counting instantiation: walk (and count) the 4-wide nodes the timed kernel walks instead of the reference's pairs
};

#define WAVE_WG 8 // waves per k_tail_wave workgroup: an empty round costs the dispatch of its workgroups, so few large ones

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

// one primitive handed over by value (rtmi_intersect_primitive): kind 0 triangle {p1} {p2} {p3}, 1 sphere {pos, r^2}, 2 plane {N, d},
// 3 quad {invT rows 0..2} {size}, 4 cube {invM rows 0..2} {b0} {b1}
struct OnePrim { int kind; int pad[3]; float4 d[5]; };

#ifndef RTMI_NOALLOC_MAX_RAYS
#define RTMI_NOALLOC_MAX_RAYS ( 1 << 18 )
#endif
// this device's address of the file-scope scratch array of rtmi_trace_rays_noalloc (rt_kernels.hip); the caller has made `device` current
hipError_t rtk_scratch_address( int device, void** out );

// ---- launchers (rt_kernels.hip) ----------------------------------------------------------------------------------------------
void rtk_primary( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState st, float4* E, float4* Es, uint32_t N, CamProp cam, int W, int H, float invW, float invH, const uint32_t* seeds, int sample_begin, int sample_stride, uint32_t* wq, DevCounters* ctr );
void rtk_trace( bool any, bool count, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, TraceArgs a, DevCounters* ctr );
void rtk_qbvh_fix( bool any, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, TraceArgs a );
void rtk_shade( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, ShadowQueue sq, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, uint32_t* wq_out, uint32_t* n_walk_out, uint32_t* n_shadow, DevCounters* ctr );
void rtk_tail( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, int max_bounces, uint32_t dense_above, int min_lanes, uint32_t lds_words, uint32_t team_below, DevCounters* ctr, int stop_late );
void rtk_tail_decide( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, const uint32_t* n_in, const uint32_t* draining, uint32_t team_below, uint32_t wave_below );
void rtk_tail_team( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, int max_bounces, DevCounters* ctr );
void rtk_tail_wave( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState in, PathState out, Sinks sinks, const uint32_t* n_in, uint32_t* n_out, int max_bounces, DevCounters* ctr );
void rtk_rays_wave( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int* obj_out, float* t_out );
void rtk_carry_out( dim3 grid, dim3 block, size_t lds, hipStream_t stream, PathState in, const uint32_t* n_in, PathState rest, uint32_t* n_rest, PathState carry, uint32_t* n_carry, uint32_t cap );
void rtk_carry_in( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, PathState carry, const uint32_t* n_carry, uint32_t cap, PathState st, const uint32_t* n_st, uint32_t* wq, DevCounters* ctr );
void rtk_carry_commit( dim3 grid, dim3 block, size_t lds, hipStream_t stream, uint32_t* n_st, uint32_t* n_carry, uint32_t cap );
void rtk_fold_late( dim3 grid, dim3 block, size_t lds, hipStream_t stream, float4* acc, long long* late, float4* odd, int WH );
void rtk_next_bounce( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, int consumed, uint32_t* report, int drop );
void rtk_next_bounce_pre( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, int consumed );
void rtk_next_bounce_post( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DevCounters* ctr, int consumed, uint32_t* report );
void rtk_accumulate( dim3 grid, dim3 block, size_t lds, hipStream_t stream, float4* acc, const float4* E, const long long* Es, int WH, int S );
void rtk_finalize( dim3 grid, dim3 block, size_t lds, hipStream_t stream, const float4* acc, uint32_t* out, int n, int frames );
void rtk_gamma( dim3 grid, dim3 block, size_t lds, hipStream_t stream, uint32_t* px, int n, const uint32_t* lut );
void rtk_chroma( dim3 grid, dim3 block, size_t lds, hipStream_t stream, const uint32_t* src, uint32_t* dst, int W, int H, float intensity );
void rtk_vignette( dim3 grid, dim3 block, size_t lds, hipStream_t stream, uint32_t* px, int W, int H, float intensity );
void rtk_trace_rays( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int* obj_out, float* t_out );
void rtk_copy_hits( dim3 grid, dim3 block, size_t lds, hipStream_t stream, PathState st, int n, int* obj_out, float* t_out );
void rtk_rays_begin( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const float* O3, const float* D3, const float* tmax, int any_hit, PathState st, ShadowQueue sq, float4* Es, uint32_t* wq, int* obj_out, float* t_out, DevCounters* ctr );
void rtk_rays_any_result( dim3 grid, dim3 block, size_t lds, hipStream_t stream, const long long* Es, int n, int* obj_out );
void rtk_isect_one( dim3 grid, dim3 block, size_t lds, hipStream_t stream, OnePrim p, int n, const float* O3, const float* D3, const float* tin, int* hit, float* tout );
void rtk_hit_attribs( dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, int n, const int* obj, const float* I3, const float* wo3, float* N3, float* alb3, int* mat );
void rtk_whitted( bool qbvh, dim3 grid, dim3 block, size_t lds, hipStream_t stream, DScene s, CamProp cam, int W, int H, float invW, float invH, int mode, int max_depth, int aa, WFrame* frames, float4* acc );
