/* rtmi.h -- C ABI of the MI355X trace-loop library (librtmi.so).
 *
 * This is the drop-in boundary for the per-pixel trace loop of Li-Yeun/Raytracer.  In the reference
 * the host (`Renderer::Init()/Tick()`, renderer.cpp:6-70, 446-643; `Scene::Scene()`, template/scene.h:94-217)
 * talks to the device only through the OpenCL wrappers `Buffer` and `Kernel`
 * (template/precomp.h:1351-1553).  Every entry point below names the reference interface it replaces.
 * Plain C: opaque handle, plain pointers and sizes, caller owns every host array (the library copies
 * during the call, like `Buffer::CopyToDevice`, template.cpp:1283-1287).  Every function returns
 * RTMI_OK (0) or a negative rtmi_status; no exceptions, no exit() (the reference's FatalError/CHECKCL,
 * template.cpp:1099-1190, become status codes + rtmi_last_error).
 *
 * One host thread per handle.  Calls are synchronous on return unless stated.
 */
#ifndef RTMI_H
#define RTMI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtmi_ctx rtmi_ctx;

typedef enum rtmi_status {
	RTMI_OK = 0,
	RTMI_ERR_INVALID = -1,    /* bad argument / call order */
	RTMI_ERR_HIP = -2,        /* a HIP runtime call failed (message in rtmi_last_error) */
	RTMI_ERR_NO_SCENE = -3,   /* render/trace before rtmi_upload_scene */
	RTMI_ERR_BVH = -4,        /* malformed node array (cycle, out-of-range index, depth > RTMI_MAX_BVH_DEPTH) */
	RTMI_ERR_NOMEM = -5
} rtmi_status;

/* == Renderer::VisualizationMode, renderer.h:9 */
typedef enum rtmi_mode { RTMI_NORMAL = 0, RTMI_DISTANCE = 1, RTMI_ALBEDO = 2, RTMI_RAYTRACING = 3, RTMI_PATHTRACING = 4 } rtmi_mode;

/* == Material::MaterialType, material.h:6 */
typedef enum rtmi_material { RTMI_DIFFUSE = 0, RTMI_MIRROR = 1, RTMI_GLASS = 2, RTMI_SUBSTANCE = 3, RTMI_LIGHT = 4 } rtmi_material;

#define RTMI_MAX_BVH_DEPTH 64 /* == stack[64] of BVH::IntersectBVHStack, bvh.h:540 */

/* == GPUBVHNode, bvh.h:11-16 (32 bytes).  Node 0 dummy root, node 1 = plane leaf, node 2 = primitive root,
 * children of an interior node are leftFirst and leftFirst+1 (bvh.h:56, 380-381). */
typedef struct rtmi_bvh_node {
	float aabbMinx, aabbMiny, aabbMinz;
	float aabbMaxx, aabbMaxy, aabbMaxz;
	int32_t leftFirst, primitiveCount;
} rtmi_bvh_node;

/* Scene upload: exactly the arrays the reference marshals in Scene::Scene() (template/scene.h:94-217) and hands
 * to its kernels in Renderer::Tick (renderer.cpp:456-510).  Global object index order: quads, spheres, cubes,
 * planes, triangles.  `total` = sum of the five counts. */
typedef struct rtmi_scene_desc {
	int32_t n_quads, n_spheres, n_cubes, n_planes, n_triangles; /* renderer.cpp:457 */
	const float* quad_inv_T;        /* [n_quads][16]  mat4 row-major, scene.h:117 */
	const float* quad_size;         /* [n_quads]      half size, scene.h:118 */
	const float* sphere_info;       /* [n_spheres][4] pos.xyz, r^2, scene.h:132 */
	const float* sphere_invr;       /* [n_spheres]    scene.h:135 */
	const float* cube_M;            /* [n_cubes][16]  scene.h:153 */
	const float* cube_inv_M;        /* [n_cubes][16]  scene.h:148 */
	const float* cube_b;            /* [n_cubes][2][4] scene.h:149-150 */
	const float* tri_verts;         /* [n_triangles][3][4] pos1,pos2,pos3 (w=0), scene.h:177-179 */
	const float* primitives;        /* [total][4] quad N | sphere centre | cube 0 | plane (N,d) | triangle N, scene.h:120-181 */
	const float* albedo;            /* [total][4] material.color, scene.h:121-182 */
	const int32_t* material_type;   /* [total]    rtmi_material, scene.h:116-176 */
	const float* ior;               /* [total]    scene.h:123-184 */
	const float* absorption;        /* [total][4] scene.h:124-185 */
	const rtmi_bvh_node* nodes;     /* [n_nodes]  bvh->gpuBvhNode, scene.h:216 */
	int32_t n_nodes;                /* bvh->nodesUsed */
	const int32_t* prim_idx;        /* [n_spheres+n_triangles] bvh->primitiveIdx, scene.h:217; < n_spheres => sphere */
	float light_corners[12];        /* quads[0].c1,c2,c3 as float4, scene.h:208 */
	float light_A, light_s;         /* renderer.cpp:483 */
	float light_emission[4];        /* renderer.cpp:483 */
	const uint32_t* texture;        /* [tex_w*tex_h] logo, 0x00RRGGBB, scene.h:212-213 */
	int32_t tex_w, tex_h;           /* 128 x 64 */
	/* Whitted point lights, Scene::GetLightPos/GetLightColor/intensity (scene.h:343-358, 562-564) */
	float whitted_light_pos[2][4];
	float whitted_light_color[2][4];
	float whitted_intensity;
} rtmi_scene_desc;

/* == the three PostProcessing passes in the order the app applies them (template.cpp:310-317):
 * gamma -> chromatic aberration -> vignette.  A pass whose enable flag is 0 is skipped. */
typedef struct rtmi_postfx {
	int32_t gamma_on, chroma_on, vignette_on;
	float gamma, chroma_intensity, vignette_intensity;
} rtmi_postfx;

typedef struct rtmi_stats {
	uint64_t primary_samples;   /* pixel-samples started since rtmi_reset_stats */
	uint64_t extend_rays;       /* closest-hit rays traversed */
	uint64_t shadow_rays;       /* any-hit rays traversed */
	/* the four below are filled only while count_nodes is 1 (counting instantiation of the kernels over the reference's pairs), else 0 */
	uint64_t extend_pairs;      /* 64-byte two-child node records fetched by closest-hit traversal */
	uint64_t extend_prims;      /* 48-byte leaf primitive records tested by closest-hit traversal */
	uint64_t shadow_pairs;      /* same for any-hit traversal */
	uint64_t shadow_prims;
	uint64_t bounce_launches;   /* extend/shade/connect rounds */
	double   ms_extend, ms_shade, ms_connect, ms_other; /* hipEvent time per kernel class (profiling mode only) */
	uint64_t n_extend_launches, n_shade_launches, n_connect_launches;
	uint64_t extend_walkers;    /* closest-hit rays that passed the root box test and walked the BVH (the rest end in the analytic tests) */
	uint64_t shadow_walkers;    /* same for any-hit rays */
	uint64_t suspended;         /* walks parked at the end of a launch and resumed in the next bounce (tail control) */
	uint64_t max_extend_steps, max_shadow_steps; /* longest single walk in steps (pairs + leaf records); count_nodes only */
	double   ms_walk;           /* hipEvent time between consecutive shade launches: a bounce's shadow walk and the next bounce's closest-hit walk
	                               share the chip in it (profiling mode only); ms_extend / ms_connect each span their own launch and overlap */
	uint64_t n_walk_windows;
	double   ms_drain;          /* host time spent inside the read-back / flush calls waiting for the last paths (always measured) */
	uint64_t n_drains;
	double   ms_extend_union;   /* profiling mode: time during which at least one closest-hit walk launch was running.  With one lane
	                               it equals ms_extend; with option lanes = 2 two launches (one per lane) overlap and each one's
	                               own duration includes the time it shared the chip with the other */
	double   ms_walk_union;     /* the same for the walk windows of ms_walk */
	double   ms_shade_union, ms_connect_union; /* the same for the k_shade and the any-hit walk launches */
	/* count_nodes = 2 (rtmi_set_profiling): the counting kernels walk the tree the timed kernels walk -- 64-byte 4-wide records with
	 * quantised boxes, every accepted hit confirmed against the 32-byte exact box of its leaf -- instead of the reference's pairs.
	 * Rays the 4-wide nodes cannot take (an infinite reciprocal direction component) still count under *_pairs. */
	uint64_t extend_wide_nodes, extend_confirms, shadow_wide_nodes, shadow_confirms;
	uint64_t dropped_paths;     /* paths ended by the bounce limit of the last tail round (option last_round_bounces; the reference's
	                               Sample has no limit and does not return from such a path, renderer.cpp:251): 0 = every path of the
	                               frames rendered since rtmi_reset_stats ended by itself, the image is exact */
} rtmi_stats;

/* Replaces Kernel::InitCL + `new Kernel`x6 + the W*H-sized `new Buffer`s of Renderer::Init (renderer.cpp:10-49,
 * template.cpp:1513-1665).  One handle = one GPU (`device_id` is the HIP ordinal). */
int rtmi_create( rtmi_ctx** out, int device_id, int width, int height );
int rtmi_destroy( rtmi_ctx* );
/* Run on a caller-owned HIP stream (a torch stream's cuda_stream) instead of the handle's own.  The value is taken as
 * it is: NULL is HIP's null (legacy default) stream, not "none".  rtmi_use_own_stream goes back to the handle's own
 * (non-blocking) stream.  Both first wait for everything enqueued so far. */
int rtmi_set_stream( rtmi_ctx*, void* hip_stream );
int rtmi_use_own_stream( rtmi_ctx* );
/* Tuning and diagnostics per handle (the reference has compile-time #defines and ImGui toggles; nothing of
 * this kind crosses its Kernel/Buffer seam).  Keys are the lower-case names listed in DESIGN.md 3 ("wide", "overlap",
 * "carry", "batch", "slots", "tail_team", "hook_kernel", "check_launches", ...); an environment variable RTMI_<KEY>
 * only supplies the default a handle starts with.  Waits for the work in flight; options that size buffers
 * reallocate them; "wide" takes effect at the next rtmi_upload_scene.  Unknown key or bad value: RTMI_ERR_INVALID. */
int rtmi_set_option( rtmi_ctx*, const char* key, long long value );
int rtmi_get_option( rtmi_ctx*, const char* key, long long* value_out );

/* Replaces the first-tick SetArguments + CopyToDevice block (renderer.cpp:453-513).  The library repacks the
 * reference-layout arrays into its own 64-byte-aligned two-child node records and leaf-ordered primitive records. */
int rtmi_upload_scene( rtmi_ctx*, const rtmi_scene_desc* );
/* Replaces cameraPropBuffer->CopyToDevice (renderer.cpp:24, 63, 631): 4 x float4 = camPos, topLeft, topRight,
 * bottomLeft (camera.h:25, 125-128). */
int rtmi_set_camera( rtmi_ctx*, const float cam[16] );
/* Replaces seedBuffer (renderer.cpp:51-57): W*H xorshift32 outputs.  State of (pixel p, sample s) is
 * WangHash(seeds[p] + s*0x9E3779B9) (0 -> 1), SURVEY A.5. */
int rtmi_set_seeds( rtmi_ctx*, const uint32_t* seeds );
/* Replaces memset(accumulator) + accumulatorBuffer->CopyToDevice (renderer.cpp:622-633, renderer.h:43-77).  Ordered after the
 * accumulates in flight; paths of earlier renders that are still bouncing are dropped with the image they belonged to (not
 * finished first), so a camera that moves every frame does not pay for them.  Asynchronous on the handle's stream. */
int rtmi_reset_accumulator( rtmi_ctx* );
/* Accumulate into a caller-owned device buffer of W*H float4 (e.g. a torch tensor that torch.distributed reduces
 * over RCCL) instead of the handle's own accumulatorBuffer (renderer.cpp:23).  NULL restores the internal one. */
int rtmi_set_accumulator( rtmi_ctx*, void* device_float4 );

/* Replaces the per-frame kernel sequence of Renderer::Tick (renderer.cpp:522-557: GeneratePrimaryRays,
 * InitialExtend, Shade, Connect, {Extend, Shade, Connect}*), WITHOUT Finalize.
 *   mode RTMI_PATHTRACING: accumulator[p] += Sample(primary ray p) for sample indices
 *        sample_begin, sample_begin+sample_stride, ... (sample_count of them)   -- renderer.cpp:246-365, 559-567
 *   other modes: accumulator[p] = Trace(primary ray p, 0) with recursion limit max_depth (renderer.cpp:133-244,
 *        578-592); sample_* ignored; anti_aliasing != 0 selects the 4-sample pattern of renderer.cpp:594-618.
 * Asynchronous; any read-back call synchronises (see rtmi_flush). */
int rtmi_render( rtmi_ctx*, int mode, int sample_begin, int sample_count, int sample_stride, int max_depth, int anti_aliasing );

/* Parity hooks (no reference counterpart as an API; they expose Scene::FindNearest / Scene::IsOccluded,
 * scene.h:359-379, for the primary rays of the current camera and for caller rays).  prim ids are global object
 * indices, -1 = miss. */
int rtmi_trace_primary( rtmi_ctx*, int32_t* prim_id_out, float* t_out );
int rtmi_trace_rays( rtmi_ctx*, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int32_t* prim_id_out, float* t_out );

/* The reference's per-object methods on the device, for hosts that keep calling them one ray at a time (compat/primitive.h,
 * compat/scene.h) -- there is no CPU implementation behind them:
 * rtmi_intersect_primitive = Triangle / Sphere / Plane / Quad / Cube::Intersect (primitive.h:50-72, 108-127, 154-158, 284-295,
 *   210-236) of ONE primitive handed over by value against n rays.  kind / data: 0 triangle, 9 floats pos1 pos2 pos3; 1 sphere, 4:
 *   pos, r^2; 2 plane, 4: N, d; 3 quad, 13: rows 0..2 of invT (12 floats, row-major) + half size; 4 cube, 18: rows 0..2 of invM,
 *   b[0], b[1].  t_in[i] = ray.t before the call; hit_out[i] != 0: the primitive was hit closer than that, t_out[i] = the new ray.t
 *   (else t_in[i]).  Needs no scene.
 * rtmi_hit_attributes = Scene::GetNormal (flipped against wo, scene.h:412) / GetAlbedo / GetMaterial().type (scene.h:380-478) of the
 *   uploaded scene for n hits: global object index, intersection point I, ray direction wo. */
int rtmi_intersect_primitive( rtmi_ctx*, int kind, const float* data, int n, const float* O3, const float* D3, const float* t_in, int32_t* hit_out, float* t_out );
int rtmi_hit_attributes( rtmi_ctx*, int n, const int32_t* obj, const float* I3, const float* wo3, float* N3_out, float* albedo3_out, int32_t* material_out );

/* No counterpart in the reference.  rtmi_trace_rays for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, a process that has handed all free HBM to other users): same arguments, same results, but
 * the rays are staged in a fixed device array that is part of the library's image (9 MB per GPU, held until the process ends)
 * instead of memory the handle allocates: no hipMalloc / hipFree, in this call or behind it.  It still copies from the caller's
 * host arrays and waits for the result (hipMemcpyAsync + hipStreamSynchronize), so it is NOT legal under HIP graph capture.
 * Kernels: option hook_kernel 0 and 2 as in rtmi_trace_rays; hook_kernel = 1 (the render path's queues, which may have to be
 * allocated) is served by kernel 0 here.  At most RTMI_NOALLOC_MAX_RAYS rays per call (more: RTMI_ERR_INVALID).  The array is
 * shared by every call on that GPU: calls that overlap on one GPU race. */
#define RTMI_NOALLOC_MAX_RAYS ( 1 << 18 )
int rtmi_trace_rays_noalloc( rtmi_ctx*, int n, const float* O3, const float* D3, const float* tmax, int any_hit, int32_t* prim_id_out, float* t_out );

/* Replaces Finalize + deviceBuffer->CopyFromDevice (renderer.cpp:553-556; RGBF32_to_RGB8, precomp.h:431-465) and,
 * when fx != NULL, the PostProcessing passes (postprocessing.h:16-99).  frames <= 0 selects the one-argument
 * RGBF32_to_RGB8 (Whitted path). */
int rtmi_finalize( rtmi_ctx*, uint32_t* rgb8_out, int frames, const rtmi_postfx* fx );
/* Post-processing alone on a caller image (PostProcessing ctor + passes, postprocessing.h:6-99). */
int rtmi_postprocess( rtmi_ctx*, uint32_t* rgb8_inout, int width, int height, const rtmi_postfx* fx );
/* accumulatorBuffer->CopyFromDevice equivalent: W*H*4 floats. */
int rtmi_read_accumulator( rtmi_ctx*, float* rgba_out );
int rtmi_synchronize( rtmi_ctx* );
/* No reference counterpart (its queues are in-order and Tick blocks on every read-back, renderer.cpp:538-556).
 * rtmi_render returns with work still enqueued on the handle's stream AND on internal streams, and with paths still
 * in flight: paths that are alive after many bounces travel on with the next render batches, and what they contribute
 * is kept in exact per-pixel sums beside the accumulator.  Every rtmi_* call that reads or resets the accumulator first
 * finishes those paths, adds the sums to the accumulator and orders itself after all of it; a caller that consumes a
 * caller-owned accumulator (rtmi_set_accumulator) on its own stream calls rtmi_flush first: it enqueues that completion
 * and makes the handle's stream wait for everything rendered so far, without waiting for it on the host -- except that,
 * when earlier batches handed paths on, those paths' first wavefront bounces are paced by the host like any batch's (it
 * reads each bounce's path count two bounces late: a few ms); the long part, the tail rounds, is only enqueued.  The
 * accumulator is the sum of whole samples only after one of these calls. */
int rtmi_flush( rtmi_ctx* );
/* Multi-GPU inside the boundary (no reference counterpart: it has one OpenCL context, precomp.h:1544-1546).  Samples are
 * split by index over one handle per GPU (rtmi_render's sample_begin / sample_stride); this sums the float4[W*H]
 * accumulators of all ranks of `rccl_comm` (an ncclComm_t created by the caller with ncclCommInitRank / ncclCommInitAll;
 * librccl is loaded on first use) onto rank `root` with ONE ncclReduce(sum, float) over xGMI, in place, on the handle's
 * stream, after finishing the paths in flight like rtmi_flush.  root < 0: ncclAllReduce (every rank gets the sum).
 * Asynchronous like rtmi_render; rtmi_finalize on the root with frames = total samples completes the frame.
 * Several handles may share one communicator (frames in flight: raytracer_amd/host/frames.h): issue their reduces in the same order
 * on every rank; each is made to execute after the previous one on that communicator, whichever handle's stream it is on. */
int rtmi_reduce_accumulator( rtmi_ctx*, void* rccl_comm, int root );

/* Replaces the per-frame re-read of Scene::GetLightPos / GetLightColor / intensity by Scene::DirectIllumination (scene.h:343-358,
 * 490-515, 562-564): the ImGui panel calls SetLightIntensity / SetLightPos / SetLightColor / SetRoofLightColor every frame
 * (template.cpp:350-465, scene.h:536-551) and the reference's CPU path sees the new values at its next Trace.  Takes effect at the next
 * rtmi_render of a Whitted mode; no re-upload of the scene.  pos / color: 2 x float4 (light 0 = the quad's centre, 1 = lightPos). */
int rtmi_set_whitted_lights( rtmi_ctx*, const float pos[8], const float color[8], float intensity );
/* Free and total device memory of the handle's GPU as the runtime reports them (hipMemGetInfo): what is left beside this
 * handle's queues (up to 19 GB per batch slot at 1080p), and the check that a call allocated nothing. */
int rtmi_memory_info( rtmi_ctx*, uint64_t* free_bytes, uint64_t* total_bytes );

/* Counters and timing (the reference only prints a frame time, renderer.cpp:635-641).  count_nodes: 0 off, 1 = the walks run
 * their counting instantiation over the reference's pairs (rtmi_stats.*_pairs, *_prims), 2 = over the 4-wide nodes (see rtmi_stats). */
int rtmi_set_profiling( rtmi_ctx*, int hip_event_timing, int count_nodes );
int rtmi_get_stats( rtmi_ctx*, rtmi_stats* out );
int rtmi_reset_stats( rtmi_ctx* );
const char* rtmi_last_error( rtmi_ctx* );
const char* rtmi_version( void );

#ifdef __cplusplus
}
#endif
#endif
