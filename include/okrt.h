/*
 * okrt.h — C ABI of the MI355X path-tracing core ("okrt" = Okay ray tracer device layer).
 *
 * This is the drop-in boundary: everything the reference's `RayTracer` host class asks of its
 * DirectX11 device layer for the compute path, restated as plain C entry points over HIP.
 * All `file:line` citations are relative to the reference checkout (Olowerr/GPU-Raytracer).
 *
 *   reference call                                           ->  okrt entry point
 *   -------------------------------------------------------------------------------------------
 *   Okay::initiateDX11              source/DirectX/DX11.cpp:21-42      okrt_create
 *   Okay::shutdownDX11 / RayTracer::shutdown  RayTracer.cpp:34-58      okrt_destroy
 *   Okay::createStructuredBuffer    DX11.h:24, DX11.cpp:104-130        okrt_set_table
 *   GPUStorage::initiate/update/updateRaw  GPUStorage.h:13-17,34-47    okrt_set_table
 *   Okay::createTextureArray        DX11.h:36, DX11.cpp:270-313        okrt_set_textures
 *   Okay::scaleTexture              DX11.h:34, DX11.cpp:232-268        okrt_scale_texture_rgba8
 *   cube-map creation               RayTracer.cpp:403-431              okrt_set_envmap
 *   RenderTexture::initiate/resize  RenderTexture.h:44-57, RayTracer.cpp:73,494-501   okrt_resize
 *   ClearUnorderedAccessViewFloat(accum)  RayTracer.h:203-209          okrt_reset_accumulation
 *   Okay::updateBuffer(RenderData) + Dispatch  RayTracer.cpp:491,219-260   okrt_render
 *   Okay::getCPUTextureData         DX11.h:32, DX11.cpp:196-230        okrt_read_result_rgba8
 *   ImGui::Image(result SRV) + present()  Application.cpp:106-128      okrt_peek_result_rgba8
 *   (no reference counterpart: float accumulation readback)            okrt_read_accum_f32x4
 *   Okay::createShader/reloadShader DX11.h:18-22                       (nothing: kernels are AOT)
 *
 * Conventions
 *   - Every call returns an `okrt_status` (0 = OK, negative = error); nothing aborts. The
 *     message of the last failure is available from okrt_last_error(). (The reference returns
 *     `bool` wrapped in OKAY_ASSERT, Utilities.h:9-13.)
 *   - The caller owns every host pointer; the library copies during the call (D3D11 copies
 *     pSysMem at create time, DX11.cpp:113-118). The library owns device memory.
 *   - A context is not thread-safe; one context per GPU; calls are ordered on one HIP stream;
 *     the read_* calls synchronise. (The reference has one immediate context, DX11.cpp:12-19.)
 *   - Table byte layouts are exactly the reference's structured-buffer layouts
 *     (resources/shaders/GPU-Structs.hlsli:7-179); element sizes are checked.
 *   - No torch / C++ types cross this boundary: plain pointers and sizes only.
 */
#ifndef OKRT_H
#define OKRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKRT_ABI_VERSION 4

typedef struct okrt_ctx okrt_ctx;

typedef enum okrt_status {
    OKRT_OK = 0,
    OKRT_ERR_INVALID_ARG = -1, /* bad pointer / size / enum / element stride */
    OKRT_ERR_DEVICE = -2,      /* a HIP runtime call failed (message has the HIP error) */
    OKRT_ERR_OOM = -3,         /* allocation failed */
    OKRT_ERR_STATE = -4,       /* call order: e.g. render before resize */
    OKRT_ERR_RANGE = -5        /* a table index points outside its table (validated at commit) */
} okrt_status;

/* Table slots are numbered like the reference's t-registers
 * (resources/shaders/ShaderResourceRegisters.h:9-21,34-46). Slots 3 (textures) and 4
 * (environment map) are images and have their own setters. */
typedef enum okrt_table {
    OKRT_TABLE_TRI_POS = 0,   /* t0  Triangle          36 B  GPU-Structs.hlsli:95-98   */
    OKRT_TABLE_TRI_INFO = 1,  /* t1  TriangleInfo     132 B  GPU-Structs.hlsli:100-111 */
    OKRT_TABLE_BVH = 2,       /* t2  BvhNode           36 B  GPU-Structs.hlsli:159-165 */
    OKRT_TABLE_OCT = 5,       /* t5  OctTreeNode       48 B  GPU-Structs.hlsli:167-179 */
    OKRT_TABLE_SPHERE = 6,    /* t6  Sphere            84 B  GPU-Structs.hlsli:86-93   */
    OKRT_TABLE_MESH = 7,      /* t7  Mesh             232 B  GPU-Structs.hlsli:113-125 */
    OKRT_TABLE_DIR_LIGHT = 8, /* t8  DirectionalLight  32 B  GPU-Structs.hlsli:127-135 */
    OKRT_TABLE_POINT_LIGHT = 9, /* t9  PointLight      32 B  GPU-Structs.hlsli:137-145 */
    OKRT_TABLE_SPOT_LIGHT = 10  /* t10 SpotLight       48 B  GPU-Structs.hlsli:147-157 */
} okrt_table;

#define OKRT_SIZEOF_TRIANGLE 36u
#define OKRT_SIZEOF_TRIANGLE_INFO 132u
#define OKRT_SIZEOF_BVH_NODE 36u
#define OKRT_SIZEOF_OCT_NODE 48u
#define OKRT_SIZEOF_SPHERE 84u
#define OKRT_SIZEOF_MESH 232u
#define OKRT_SIZEOF_DIR_LIGHT 32u
#define OKRT_SIZEOF_POINT_LIGHT 32u
#define OKRT_SIZEOF_SPOT_LIGHT 48u
#define OKRT_SIZEOF_MATERIAL 68u
#define OKRT_SIZEOF_RENDER_DATA 240u

/* The b0 constant buffer, byte for byte (resources/shaders/RaytracerCS.hlsl:19-47, host mirror
 * source/Graphics/RayTracer.h:129-157). Matrices are what the host uploads:
 * glm::transpose(glm::inverse(M)) in glm's column-major memory order (RayTracer.cpp:295-296,
 * 306-307), i.e. the 16 floats are inverse(M) in row-major order. */
typedef struct okrt_render_data {
    uint32_t accumulationEnabled;   /*   0 */
    uint32_t numAccumulationFrames; /*   4  already incremented by the host (RayTracer.cpp:489) */
    uint32_t numSpheres;            /*   8 */
    uint32_t numMeshes;             /*  12  unused by the kernel */
    uint32_t numDirLights;          /*  16 */
    uint32_t numPointLights;        /*  20 */
    uint32_t numSpotLights;         /*  24 */
    uint32_t debugMode;             /*  28  0 none, 1 bbCheckCount, 2 triCheckCount (RayTracer.h:74-79) */
    uint32_t textureDims[2];        /*  32  FULL-frame W,H (also when rendering a tile) */
    float viewPlaneDims[2];         /*  40  unused by the live code path */
    float cameraInverseProjectionMatrix[16]; /*  48 */
    float cameraInverseViewMatrix[16];       /* 112 */
    float cameraPosition[3];        /* 176 */
    float cameraNearZ;              /* 188 */
    float cameraUpDir[3];           /* 192 */
    float dofStrength;              /* 204 */
    float cameraRightDir[3];        /* 208 */
    float dofDistance;              /* 220 */
    uint32_t debugMaxCount;         /* 224 */
    float pad0[3];                  /* 228 */
} okrt_render_data;

/* Options (okrt_set_option). */
typedef enum okrt_option {
    /* 0 = reference-order traversal over the uploaded octree -> mesh -> BVH tables
     *     (RaytracerCS.hlsl:256-334, bit-identical visit order; produces the reference's
     *     bbCheckCount / triCheckCount);
     * 1 = the MI355X traversal structure derived from the same tables at commit: instances are baked to world space, so it
     *     returns the world-space closest hit. For rigid instances (every benchmark scene; identity instances bit for bit)
     *     that IS the reference's result up to equal-distance ties. Under Transform.scale != 1 the reference culls per-mesh
     *     BVH boxes by a local-space distance against the world-space best hit (quirk Q3, RaytracerCS.hlsl:273-287 vs
     *     :311-318): conservative — same picture — when every scale factor is >= 1, but with a factor < 1 the reference
     *     skips boxes that hold the closest hit. Mode 1 does not reproduce those drop-outs
     *     (tests/test_gpu_fullsize.py::test_scaled_instances_*); mode 0 does, exactly;
     * 2 = (default, OKRT_TRAVERSAL_AUTO) parity first: mode 1 wherever it shows the reference's picture, mode 0 for a scene
     *     whose instance table has OKRT_BUILD_FLAG_SHRUNK_INSTANCES (okrt_get_build_info) — many times slower there, and
     *     said once through okrt_last_error. A host that prefers the true closest hit at full speed selects mode 1. */
    OKRT_OPT_TRAVERSAL = 0,
    /* NUM_BOUNCES of RaytracerCS.hlsl:6 (default 1 = HEAD; a path has max_bounces+1 segments). */
    OKRT_OPT_MAX_BOUNCES = 1,
    /* 1 = sum the traversal counters of every render into okrt_counters (default 0). `pixels` and
     * `segments` are always counted by the fast pipeline (one atomic per kernel launch). */
    OKRT_OPT_COUNTERS = 2,
    /* n > 0 = record HIP events around every kernel of each render (ring of the last n renders) so
     * okrt_last_render_ms / okrt_get_stage_ms work; 0 = off (default). */
    OKRT_OPT_TIMING = 3,
    /* Who derives the traversal structure when a geometry / instance / octree table changed:
     * 0 = (default) the host (binned SAH over the baked instances, then uploaded);
     * 1 = the GPU, straight from the tables in HBM (Morton sort + parallel locally-ordered clustering + collapse to
     *     4-wide nodes): the step the reference spends in BvhBuilder.cpp:72-97 / the octree rebuild of
     *     Application.cpp:312-317 on every scene edit, at interactive rates. Both cover exactly the triangles the
     *     reference shader reaches, so renders agree up to equal-distance ties. */
    OKRT_OPT_BUILDER = 4,
    /* n >= 2 (default 16): okrt_render calls that continue the progressive accumulation of an unchanged scene (same
     * RenderData apart from numAccumulationFrames = previous + 1, fast traversal, no debug mode, library-owned stream
     * and framebuffers) are deferred and up to n of them traced by one set of kernel launches — the reference's
     * main loop `for (;;) render();` (Application.cpp:106-128) gets the throughput of okrt_render_frames without a
     * change to the host. Arguments and tables are validated inside the deferred call itself, and the queues its launch
     * set will need are allocated as soon as a second frame waits (an allocator refusal is reported there, inside the
     * host's loop, after smaller set sizes have been tried); what can only surface later (at the call that executes the frames) is a failing launch. Every other entry point
     * (read-backs, synchronize, uploads that change a table, options, timers, ...) executes the waiting frames first,
     * so nothing can observe the deferral; the framebuffers are bit-identical to immediate execution. 0 or 1 = launch
     * every frame inside its own okrt_render.
     * Footprint: a launch set of F frames keeps (76 B per ray + 16 B per pixel) x F of queues in flight (132 B per ray
     * when NUM_BOUNCES >= 2): 1920x1080 -> 0.19 GB per frame, 3.0 GB at the default of 16, 12 GB at the maximum of 64;
     * 3840x2160 -> 0.76 GB per frame. OKRT_OPT_QUEUE_MEMORY_MB bounds it. */
    OKRT_OPT_DEFER = 5,
    /* Upper bound, in MiB, for the ray queues of one launch set (76 or 132 bytes per ray and frame in flight + 16 per
     * pixel and frame, see OKRT_OPT_DEFER; 1080p x 64 frames = 12 GiB). Default, when the host never sets it: a quarter of the
     * device memory that is free at okrt_resize (72 GB on an otherwise empty 288 GB part: 64-frame sets at 3840x2160 still
     * fit; on a crowded device the sets shrink instead of starving the host's other allocations). 0 = no bound but the
     * allocator's. A launch set that does not fit — by this bound or by the allocator — is split in halves until it does; a
     * single frame is always attempted. */
    OKRT_OPT_QUEUE_MEMORY_MB = 6,
    /* Test hook: n > 0 = the n-th ray-queue allocation from now on fails as hipMalloc would when memory runs out (so that
     * the launch-set split on a REAL allocator failure can be tested); 0 = off. Process-wide. */
    OKRT_OPT_DEBUG_FAIL_ALLOC = 7,
    /* n > 0 (default 4, at most 32) = launch sets in flight together: consecutive launch sets of at most n frames that
     * continue a progressive accumulation (fast traversal, no debug mode, library-owned stream and framebuffers) take turns
     * on four sets of ray queues, each with its own HIP stream, so the next sets' camera rays fill the GPU while the previous
     * sets' longest rays — a chain of dependent steps that no amount of parallelism shortens — are still draining. Only the
     * accumulation is ordered between the sets (each adds its frames in one kernel that waits for the previous set's), so
     * the framebuffers are bit-identical to one set at a time. Every entry point other than okrt_render /
     * okrt_render_frames joins the streams first: nothing else can observe the overlap. What it buys depends on how long
     * the scene's longest rays are against a set's bulk (1080p, ms per frame, off -> on): one frame per set 0.91 -> 0.66
     * (Sponza-class), 1.88 -> 0.67 (871 k-triangle mesh); sets of 4: 0.64 -> 0.59, 0.66 -> 0.33; sets of 16:
     * 0.52 -> 0.57, 0.32 -> 0.29 — large persistent grids sharing the GPU cost each other more than the overlap returns
     * unless the tail is long, hence the default of 4; a host with a tail-heavy scene sets 32.
     * Small tiles (round 4): a LARGER launch set that traces at most 12 Mi camera rays — one rank's stripes of an N-way split —
     * is issued as two sub-sets that ride the slots the same way (8-way split of a 1080p frame, 20 frames: 1.67 -> 1.58 ms per
     * rank; a full 1080p frame, where it would cost 4.5 %, is not cut). Same framebuffers bit for bit.
     * Footprint: the queues of up to FOUR launch sets in flight: sets of up to n frames (default: 4 x 0.76 GB at 1080p), and
     * the sub-sets of a small tile's larger set, each up to 32 frames on its own slot (a 64-frame set of a 1/8 tile keeps
     * about twice the queue memory of one undivided set). 0 = one launch set at a time. */
    OKRT_OPT_PIPELINE = 8,
    /* 1 (default) = work whose result is provably +-0 is not executed on the LAST path segment. When the frame's RenderData
     * holds no light (RaytracerCS.hlsl:448-487 never runs), every uploaded Material (t6, t7) has emission colour * power ==
     * (0, 0, 0) and |albedo| <= 1, a hit on the last segment contributes `emission * contribution * (1 - transparencyFactor)`
     * = +-0 (:509-510) and the sampled bounce direction is never traced: the fast pipeline then skips the shading record, the
     * texture taps and the BSDF draws of those hits, the bounce-ray kernel stores only hit-or-miss, and only the misses (the
     * environment lookup of :489-493) are shaded. Framebuffers are bit-identical to 0 = every segment shaded in full
     * (tests/test_gpu_properties.py); any emitter, any light, an albedo outside [-1, 1] or a NaN in a material, and
     * OKRT_OPT_COUNTERS (tex_taps / mesh_hits count the reference's work) select the full path by themselves. */
    OKRT_OPT_LAST_SEGMENT_SHORTCUT = 9,
    /* 1 = (only where the shortcut above applies) the last segment's ray ends at the FIRST accepted hit instead of the closest:
     * hit-or-miss is all the picture needs there, so framebuffers stay bit-identical, but those rays are no longer the
     * closest-hit segments the Mrays/s metric counts — bench.py reports this mode under its own name, never as the
     * headline. Default 0. */
    OKRT_OPT_LAST_SEGMENT_ANY_HIT = 10
} okrt_option;

#define OKRT_BUILDER_HOST 0
#define OKRT_BUILDER_DEVICE 1

#define OKRT_TRAVERSAL_REFERENCE 0
#define OKRT_TRAVERSAL_FAST 1
#define OKRT_TRAVERSAL_AUTO 2

/* Sums over all rendered pixels since the last okrt_reset_counters. The first five are defined
 * by the reference traversal order (they are what the reference's debug heat maps count,
 * RaytracerCS.hlsl:261,267,284,305) and are only filled in OKRT_TRAVERSAL_REFERENCE mode. */
typedef struct okrt_counters {
    uint64_t pixels;     /* samples started */
    uint64_t segments;   /* findClosestHit invocations = rays traced */
    uint64_t n_oct;      /* octree nodes popped        (RaytracerCS.hlsl:258-261) */
    uint64_t n_mesh;     /* instances entered          (RaytracerCS.hlsl:265-272) */
    uint64_t n_node;     /* BVH nodes popped           (RaytracerCS.hlsl:280-284) */
    uint64_t n_tri;      /* triangles tested           (RaytracerCS.hlsl:297-306) */
    uint64_t mesh_hits;  /* segments whose closest hit is a mesh triangle (TriangleInfo fetch) */
    uint64_t sphere_hits;
    uint64_t tex_taps;   /* bilinear material/normal-map taps */
    uint64_t env_taps;   /* environment lookups on a miss */
    uint64_t fast_nodes; /* OKRT_TRAVERSAL_FAST: wide-node visits (diagnostic) */
    uint64_t fast_tris;  /* OKRT_TRAVERSAL_FAST: triangle tests (diagnostic) */
} okrt_counters;

/* ---- lifetime --------------------------------------------------------------------------- */
int okrt_abi_version(void);
/* Number of GPUs this process can open (IDXGIFactory adapter enumeration has no counterpart in the reference: it takes
 * the default adapter, DX11.cpp:21-42). A multi-GPU host sizes its rank count from this. */
okrt_status okrt_device_count(int* out_count);
okrt_status okrt_create(int device, okrt_ctx** out_ctx);
okrt_status okrt_destroy(okrt_ctx* ctx);
/* never NULL; "" when nothing was reported; ctx == NULL: why the last okrt_create failed. Besides the message of the last
 * failed call it carries ONE kind of warning with status OKRT_OK (text starts with "warning:"): the first render under
 * OKRT_TRAVERSAL_AUTO of an instance table with OKRT_BUILD_FLAG_SHRUNK_INSTANCES — the default then renders with the slow
 * reference-order kernel to keep the reference's picture (see OKRT_OPT_TRAVERSAL) — unless the host has called okrt_get_build_info. */
const char* okrt_last_error(const okrt_ctx* ctx);

/* ---- scene upload (the Scene->Graphics interface) ------------------------------------------ */
/* Replace a whole table. `elem_bytes` must equal the slot's reference stride. count==0 or
 * data==NULL empties the table. A different count reallocates (GPUStorage.h:36-37). */
okrt_status okrt_set_table(okrt_ctx* ctx, okrt_table slot, const void* data,
                           uint32_t elem_bytes, uint32_t count);
/* Texture2DArray of `count` RGBA8 slices of width x height (linear, no sRGB: DX11.cpp:281). */
okrt_status okrt_set_textures(okrt_ctx* ctx, const uint8_t* rgba8, uint32_t count,
                              uint32_t width, uint32_t height);
/* TextureCube: 6 RGBA8 faces of face x face in D3D order +X,-X,+Y,-Y,+Z,-Z
 * (RayTracer.cpp:377-399). NULL / face==0 unbinds it (environment reads black). */
okrt_status okrt_set_envmap(okrt_ctx* ctx, const uint8_t* rgba8_6faces, uint32_t face);
/* One bilinear tap per output texel, CLAMP addressing, RGBA8 out (DX11.cpp:232-268 with
 * resources/shaders/Scale/ScalePS.hlsl). src/dst are host pointers. */
okrt_status okrt_scale_texture_rgba8(okrt_ctx* ctx, const uint8_t* src, uint32_t src_w,
                                     uint32_t src_h, uint8_t* dst, uint32_t dst_w, uint32_t dst_h);

/* ---- framebuffer --------------------------------------------------------------------------- */
/* (Re)allocate result (RGBA8) + accumulation (float4) for a W x H frame and zero the
 * accumulation (RayTracer::onResize, RayTracer.cpp:494-501). Resets the tile to the full frame. */
okrt_status okrt_resize(okrt_ctx* ctx, uint32_t width, uint32_t height);
/* Image-space split for multi-GPU (new; the reference is single-GPU). The frame is cut into
 * horizontal stripes of `stripe_rows` rows; this context owns stripes
 * stripe_index, stripe_index+stripe_stride, ... Its framebuffers hold only the owned rows,
 * packed in order. Seeds use GLOBAL pixel coordinates, so owned pixels are bit-identical to a
 * full-frame render (RaytracerCS.hlsl:430). (H,0,1) = full frame. Zeroes the accumulation. */
okrt_status okrt_set_tile(okrt_ctx* ctx, uint32_t stripe_rows, uint32_t stripe_index,
                          uint32_t stripe_stride);
/* Number of rows this context owns under the current tile. */
okrt_status okrt_tile_rows(const okrt_ctx* ctx, uint32_t* out_rows);
/* The frame size of the last okrt_resize (0 x 0 before it): what the read_* calls size their output by, together with
 * okrt_tile_rows (ID3D11Texture2D::GetDesc of the target texture in the reference, DX11.cpp:200-203). */
okrt_status okrt_frame_size(const okrt_ctx* ctx, uint32_t* out_width, uint32_t* out_height);
/* Render into caller-owned device memory instead (e.g. a torch tensor used by an RCCL gather).
 * accum_dev: rows*W float4, result_dev: rows*W RGBA8; NULL selects the internal buffer.
 * Must be re-bound after okrt_resize / okrt_set_tile. */
okrt_status okrt_bind_framebuffers(okrt_ctx* ctx, void* accum_dev, void* result_dev);
okrt_status okrt_reset_accumulation(okrt_ctx* ctx);

/* ---- render + readback --------------------------------------------------------------------- */
/* One frame = one sample per owned pixel; asynchronous on the context's stream. */
okrt_status okrt_render(okrt_ctx* ctx, const okrt_render_data* rd);
/* `frames` consecutive frames in one call: exactly what `frames` okrt_render calls with
 * rd->numAccumulationFrames, +1, +2, ... would leave in the framebuffers (bit-identical), but traced by shared
 * kernel launches — the progressive-accumulation loop of the reference (Application.cpp:106-128 calling
 * RayTracer::render once per frame) without paying the per-frame latency floor. With accumulation disabled, a debug
 * mode or the reference-order traversal it simply loops. */
okrt_status okrt_render_frames(okrt_ctx* ctx, const okrt_render_data* rd, uint32_t frames);
okrt_status okrt_synchronize(okrt_ctx* ctx);
/* Owned rows only, packed: rows*W*4 bytes / rows*W*4 floats. Row 0 = top of the image. */
okrt_status okrt_read_result_rgba8(okrt_ctx* ctx, uint8_t* dst);
okrt_status okrt_read_accum_f32x4(okrt_ctx* ctx, float* dst);
/* For a host that PRESENTS every frame (the reference shows the result texture through ImGui and calls present() once
 * per loop iteration, Application.cpp:106-128): the newest RESOLVED picture, without executing deferred frames and
 * without waiting for launch sets still in flight. okrt_read_result_rgba8 is exact but flushes, which costs a per-frame
 * caller the per-call latency (1.0 ms instead of 0.58 ms per 1080p frame); a display loop `render(); peek(); present();`
 * keeps the batched rate and shows a picture that is at most one launch set (OKRT_OPT_DEFER frames, default 16) plus the
 * set in flight behind the accumulation. *out_frames (optional) = the numAccumulationFrames of the picture (0 = nothing
 * resolved yet: zeros). From the second call on the library keeps two device-side copies of the result, refreshed at the
 * end of every launch set (8 bytes per pixel extra), and this call only synchronises its own copy stream. */
okrt_status okrt_peek_result_rgba8(okrt_ctx* ctx, uint8_t* dst, uint32_t* out_frames);
/* Restore a float accumulation saved with okrt_read_accum_f32x4 (owned rows, packed): together with the host's
 * numAccumulationFrames this resumes a long progressive render ("checkpoint / resume"; the reference
 * has none). */
okrt_status okrt_write_accum_f32x4(okrt_ctx* ctx, const float* src);

/* ---- options, counters, timing ------------------------------------------------------------- */
okrt_status okrt_set_option(okrt_ctx* ctx, okrt_option opt, int64_t value);
okrt_status okrt_get_counters(okrt_ctx* ctx, okrt_counters* out); /* synchronises */
okrt_status okrt_reset_counters(okrt_ctx* ctx);
/* Use a caller-provided hipStream_t (NULL restores the context's own stream). */
okrt_status okrt_set_stream(okrt_ctx* ctx, void* hip_stream);
/* HIP-event timing on the context's stream: begin/end bracket any number of okrt_render calls;
 * okrt_timer_elapsed_ms synchronises on the end event. */
okrt_status okrt_timer_begin(okrt_ctx* ctx);
okrt_status okrt_timer_end(okrt_ctx* ctx);
okrt_status okrt_timer_elapsed_ms(okrt_ctx* ctx, float* out_ms);
/* With OKRT_OPT_TIMING: device time of the kernels of the most recent launch set: one okrt_render, or the frames that
 * were traced together (okrt_render_frames, deferred okrt_render calls). */
okrt_status okrt_last_render_ms(okrt_ctx* ctx, float* out_ms);
/* With OKRT_OPT_TIMING = n: device time summed over the renders still in the ring (at most n), split
 * by kernel family: out_ms[0] = traversal kernels (the camera-segment kernel — which generates, traces AND shades
 * the camera rays — plus the trace kernels of the later segments; or the reference-order megakernel),
 * out_ms[1] = the shading kernels of the later segments, out_ms[2] = everything else inside the frame (memsets,
 * gaps). *out_frames = number of frames summed. Synchronises. */
okrt_status okrt_get_stage_ms(okrt_ctx* ctx, float out_ms[3], uint32_t* out_frames);
/* The same ring, kernel by kernel: out_ms[k] = device time between event k and k + 1 of a launch set, summed over the
 * launch sets in the ring that have the newest one's shape (same number of intervals AND the same number of frames, so
 * *out_frames / *out_launches is the real launch-set size). For the fast pipeline the
 * intervals are, in launch order: 0 = the camera-segment kernel, 1 = (frame-slot accumulation when max_bounces == 0),
 * then per later segment b = 1.. : 2b = traceKernel, 2b + 1 = shadeKernel (+ the frame-slot accumulation after the last);
 * the reference-order traversal has one interval. *out_count = number of intervals (at most `capacity` are written),
 * *out_launches = launch sets summed, *out_frames = frames they traced. Synchronises. */
okrt_status okrt_get_interval_ms(okrt_ctx* ctx, float* out_ms, uint32_t capacity, uint32_t* out_count, uint32_t* out_launches,
                                 uint32_t* out_frames);

/* Validates the uploaded tables and (re)builds the derived traversal structure now instead of inside the next
 * okrt_render — the reference's "Rebuild BVH / refresh octree" moment (Application.cpp:312-317,598-604). Optional:
 * okrt_render commits whatever is pending. */
okrt_status okrt_commit(okrt_ctx* ctx);
typedef struct okrt_build_info {
    uint32_t builder;        /* OKRT_BUILDER_* that made the current structure */
    uint32_t num_nodes;      /* 64-byte 4-wide nodes */
    uint32_t num_triangles;  /* 48-byte baked triangles (instances x mesh triangles) */
    uint32_t max_depth;      /* node levels + 1 */
    float device_ms;         /* device builder: first to last kernel on the stream; 0 for the host builder */
    float wall_ms;           /* host wall clock of the whole build incl. upload / read-backs */
    uint32_t flags;          /* OKRT_BUILD_FLAG_*: what the instance table (t7) means for reference parity, see OKRT_OPT_TRAVERSAL */
} okrt_build_info;
/* some instance matrix is not a rigid motion (Transform.scale != 1, Components.h:7-23) */
#define OKRT_BUILD_FLAG_SCALED_INSTANCES 1u
/* some instance matrix shrinks at least one direction (a singular value < 1): there the reference's BVH cull compares a
 * LARGER local-space box distance with the world-space best hit (quirk Q3, RaytracerCS.hlsl:285 vs :311-318) and drops
 * geometry it should test; OKRT_TRAVERSAL_FAST returns the true closest hit instead (documented deviation), the default
 * OKRT_TRAVERSAL_AUTO renders such a scene in reference order */
#define OKRT_BUILD_FLAG_SHRUNK_INSTANCES 2u
/* every uploaded Material (t6, t7) has emission colour * power == (0, 0, 0) and |albedo| <= 1: the material half of the condition
 * under which OKRT_OPT_LAST_SEGMENT_SHORTCUT applies (the other half is a RenderData without lights) */
#define OKRT_BUILD_FLAG_DARK_MATERIALS 4u
/* the camera segment of this structure is traced ray by ray instead of one packet of 64 camera rays per wave: chosen by the first launch
 * set after a build from a probe of that frame (about eight stripes of the full frame through the counting variants of both
 * kernels; gpu-raytracer_amd/csrc/okrt_fast_pipeline.hip: probeCameraSegment) — scenes of triangles far below a pixel, where a packet
 * walks the union of 64 rays' leaves. Meaningful after the first fast-traversal render. Same closest hits either way. */
#define OKRT_BUILD_FLAG_CAMERA_PER_RAY 8u
okrt_status okrt_get_build_info(okrt_ctx* ctx, okrt_build_info* out);
/* Test / tooling access: copies the derived structure to the host (64 B per node, 48 B per triangle; layouts in
 * gpu-raytracer_amd/csrc/okrt_fast.h). Either pointer may be NULL. OKRT_ERR_STATE if nothing is built. */
okrt_status okrt_get_fast_structure(okrt_ctx* ctx, void* nodes64, uint32_t node_capacity, void* tris48, uint32_t tri_capacity);

/* Known-answer probes: runs one leaf function of the device code (gpu-raytracer_amd/csrc/okrt_device.h) over `n` records,
 * so that tests can compare it bit for bit with the CPU oracle's restatement of the same reference function. `in`
 * and `out` are host arrays of n * in_floats / n * out_floats floats (integers travel as bit patterns):
 *   PCG_HASH        seed                      -> hash, new seed          (GPU-Utilities.hlsli:18-25)
 *   RANDOM_FLOAT    seed                      -> value, new seed         (:51-54)
 *   RAY_SPHERE      o[3] d[3] centre[3] r     -> t                       (:98-114)
 *   RAY_TRIANGLE    o[3] d[3] p0 p1 p2        -> t, u, v                 (:117-155)
 *   RAY_AABB        o[3] d[3] min[3] max[3]   -> distance | FLT_MAX      (:183-197)
 *   REFRACT         d[3] n[3] ratio           -> direction[3]            (:85-93)
 *   SAMPLE_TEXTURE  slice(bits) u v           -> rgb  (the uploaded texture array; RaytracerCS.hlsl:133-136)
 *   SAMPLE_ENV      d[3]                      -> rgb  (the uploaded cube map; :86-88)
 *   CLOSEST_HIT     o[3] d[3] numSpheres(bits) -> distance, hit?, bbCheckCount, triCheckCount, hitIdx, triIdx
 *                   (the reference-order traversal over the uploaded tables; RaytracerCS.hlsl:218-334)
 *   CLOSEST_HIT_FAST o[3] d[3] numSpheres(bits) -> distance, hit?, nodes visited, hitIdx, triIdx
 *                   (the derived traversal structure, walked by one thread per ray)
 *   CLOSEST_HIT_TRACE o[3] d[3] numSpheres(bits) -> distance, hit?, 0, hitIdx, triIdx
 *                   (the same structure, but through traceKernel ITSELF — the persistent bounce-ray kernel with its queue shards,
 *                   lane refill, step blocks and LDS stack: the rays are put on a sharded queue, one launch traces them and
 *                   the hit records are read back; numSpheres is taken from the first record)
 * and, containing transcendentals (device libm vs host libm: last-ulp agreement only):
 *   RANDOM_VECTOR   seed                      -> v[3], new seed          (GPU-Utilities.hlsli:56-70)
 *   REFLECTANCE     cosine, index             -> value                   (RaytracerCS.hlsl:100-105)
 *   REFLECT_DIRECTION    d[3] n[3] roughness seed -> direction[3], new seed   (:107-112)
 *   TRANSPARENCY_BOUNCE  d[3] n[3] ior seed       -> direction[3], new seed   (:114-131)
 *   PRIMARY_RAY     px(bits) py(bits) RenderData[60 floats] -> origin[3], direction[3], seed   (:177-216, 430)
 *   SEGMENT_RAY     px(bits) py(bits) k(bits) RenderData[60 floats] -> alive?, origin[3], direction[3], seed
 *                   (the ray entering path segment k of that pixel along the reference-order path, :444-514; k = 0 is
 *                   the camera ray; alive = 0 when the path ended before segment k)
 * and, comparing two device-side computations with each other (no CPU counterpart):
 *   IEEE_SHORTCUTS  first(bits) count(bits)   -> n_rcp(bits), n_sqrt(bits), first_rcp(bits), first_sqrt(bits): among the
 *                   `count` consecutive float bit patterns from `first`, how many make the short reciprocal / square root
 *                   of okrt_device.h differ from the compiler's IEEE `1.f / x` / `sqrtf(x)`, and the first such pattern */
typedef enum okrt_kat {
    OKRT_KAT_PCG_HASH = 0, OKRT_KAT_RANDOM_FLOAT = 1, OKRT_KAT_RAY_SPHERE = 2, OKRT_KAT_RAY_TRIANGLE = 3, OKRT_KAT_RAY_AABB = 4,
    OKRT_KAT_REFRACT = 5, OKRT_KAT_SAMPLE_TEXTURE = 6, OKRT_KAT_SAMPLE_ENV = 7, OKRT_KAT_CLOSEST_HIT = 8,
    OKRT_KAT_CLOSEST_HIT_FAST = 9, OKRT_KAT_RANDOM_VECTOR = 10, OKRT_KAT_REFLECTANCE = 11, OKRT_KAT_REFLECT_DIRECTION = 12,
    OKRT_KAT_TRANSPARENCY_BOUNCE = 13, OKRT_KAT_PRIMARY_RAY = 14, OKRT_KAT_SEGMENT_RAY = 15, OKRT_KAT_IEEE_SHORTCUTS = 16,
    OKRT_KAT_CLOSEST_HIT_TRACE = 17
} okrt_kat;
okrt_status okrt_debug_kat(okrt_ctx* ctx, okrt_kat op, uint32_t n, const float* in, uint32_t in_floats, float* out, uint32_t out_floats);

/* ---- multi-GPU: image-space split + RCCL gather over xGMI (new; the reference is single-GPU) ------------------------
 * Every rank renders its stripes (okrt_set_tile(stripe_rows, rank, nranks)) with NO per-frame collective; whenever the
 * host wants the whole frame, the packed float4 accumulation tiles are gathered to one root GPU — one ncclRecv per peer
 * on the root, one ncclSend per peer, inside one ncclGroupStart/End, on the context's stream — and de-interleaved there
 * into a full H x W frame. RCCL (librccl.so.1) is loaded on first use; a host that never calls these needs no RCCL.
 *   one process per GPU:   rank 0: okrt_comm_unique_id -> distribute the 128 bytes out of band (MPI_Bcast, a file, a
 *                          socket) -> every rank: okrt_comm_create(ctx, id, rank, nranks) -> okrt_mgpu_gather_accum
 *   one process, n GPUs:   okrt_comm_create_all(ctxs, n, comms) -> okrt_mgpu_gather_accum_all(ctxs, comms, n, root, ...)
 * A communicator of one rank never touches RCCL (the gather is a device-to-device copy). */
#define OKRT_COMM_ID_BYTES 128
typedef struct okrt_comm okrt_comm;
/* Transports. "rccl": the sends / receives above (default). "copy" (okrt_comm_create_all only): the root's stream waits
 * for each peer's stream and pulls the tile with hipMemcpyPeerAsync — chosen when two contexts share a device (RCCL
 * refuses two ranks on one GPU; this is how an N-rank split is rehearsed on fewer GPUs) or with
 * OKRT_MGPU_TRANSPORT=copy in the environment. okrt_comm_transport names the one in use ("single" for one rank). */
const char* okrt_comm_transport(const okrt_comm* comm);
okrt_status okrt_comm_unique_id(uint8_t out_id[OKRT_COMM_ID_BYTES]);
okrt_status okrt_comm_create(okrt_ctx* ctx, const uint8_t id[OKRT_COMM_ID_BYTES], uint32_t rank, uint32_t nranks, okrt_comm** out_comm);
/* out_comms receives n communicators, rank r for ctxs[r]. Distinct devices -> RCCL; shared devices -> copy transport. */
okrt_status okrt_comm_create_all(okrt_ctx* const* ctxs, uint32_t n, okrt_comm** out_comms);
okrt_status okrt_comm_destroy(okrt_comm* comm);
const char* okrt_comm_last_error(const okrt_comm* comm); /* comm == NULL: why the last create on this thread failed */
/* On the root of the most recent okrt_mgpu_gather_accum[_all] / okrt_mgpu_reduce_accum[_all]: device time on the root's stream
 * from the point where the exchange was enqueued (behind the root's own frames) to the assembled full frame — what the root
 * waited for its peers, the transfers, the de-interleave. Synchronises with that exchange only. OKRT_ERR_STATE on a
 * communicator that has not been a root yet. (No reference counterpart: the reference renders on one GPU, RayTracer.cpp:256.) */
okrt_status okrt_comm_exchange_ms(okrt_comm* comm, float* out_ms);
/* Collective over the communicator. Every context must have the same frame size and the tile (stripe_rows, rank, nranks)
 * of its rank. On `root`, full_frame_dev (device memory, H * W float4; NULL = a library-owned buffer, read it with
 * okrt_mgpu_read_gathered) receives the whole accumulation, row 0 = top. Asynchronous on each context's stream. */
okrt_status okrt_mgpu_gather_accum(okrt_ctx* ctx, okrt_comm* comm, uint32_t root, void* full_frame_dev);
/* The same for n contexts driven by one thread (one RCCL group around all sends and receives). n must be the
 * communicator's whole rank count (or 1): a subset would post sends without their receive. */
okrt_status okrt_mgpu_gather_accum_all(okrt_ctx* const* ctxs, okrt_comm* const* comms, uint32_t n, uint32_t root, void* full_frame_dev);
/* Samples split: every rank renders the WHOLE frame (full-frame tile) for its own block of frame
 * indices; this sums the float4 accumulations onto `root` (ncclReduce; copy transport: pull + add in rank order) into
 * full_frame_dev or the library-owned frame (okrt_mgpu_read_gathered). alpha of the sum = total frames. */
okrt_status okrt_mgpu_reduce_accum(okrt_ctx* ctx, okrt_comm* comm, uint32_t root, void* full_frame_dev);
okrt_status okrt_mgpu_reduce_accum_all(okrt_ctx* const* ctxs, okrt_comm* const* comms, uint32_t n, uint32_t root, void* full_frame_dev);
/* Root only: copies the library-owned gathered frame (H * W * 4 floats) to the host; synchronises. */
okrt_status okrt_mgpu_read_gathered(okrt_ctx* ctx, okrt_comm* comm, float* dst);
/* The root-side step on its own: `tiles_dev` holds nranks packed tiles, tile r at r * tile_stride_rows rows of W float4
 * (rows of stripes r, r + nranks, ... in order); writes the H x W frame to full_frame_dev. Uses the context's frame size. */
okrt_status okrt_mgpu_deinterleave(okrt_ctx* ctx, const void* tiles_dev, uint32_t nranks, uint32_t stripe_rows, uint32_t tile_stride_rows,
                                   void* full_frame_dev);

/* ---- entry-point table -------------------------------------------------------------------------
 * The same entry points as a struct of function pointers, so a host written against this ABI
 * (gpu-raytracer_amd/host: the mirror of the reference's RayTracer class) binds to the device
 * layer at run time instead of at link time. okrt_get_api() returns the HIP implementation. */
typedef struct okrt_api {
    uint32_t abi_version;
    okrt_status (*create)(int device, okrt_ctx** out_ctx);
    okrt_status (*destroy)(okrt_ctx* ctx);
    const char* (*last_error)(const okrt_ctx* ctx);
    okrt_status (*set_table)(okrt_ctx*, okrt_table, const void*, uint32_t, uint32_t);
    okrt_status (*set_textures)(okrt_ctx*, const uint8_t*, uint32_t, uint32_t, uint32_t);
    okrt_status (*set_envmap)(okrt_ctx*, const uint8_t*, uint32_t);
    okrt_status (*scale_texture_rgba8)(okrt_ctx*, const uint8_t*, uint32_t, uint32_t, uint8_t*, uint32_t, uint32_t);
    okrt_status (*resize)(okrt_ctx*, uint32_t, uint32_t);
    okrt_status (*set_tile)(okrt_ctx*, uint32_t, uint32_t, uint32_t);
    okrt_status (*tile_rows)(const okrt_ctx*, uint32_t*);
    okrt_status (*reset_accumulation)(okrt_ctx*);
    okrt_status (*render)(okrt_ctx*, const okrt_render_data*);
    okrt_status (*synchronize)(okrt_ctx*);
    okrt_status (*read_result_rgba8)(okrt_ctx*, uint8_t*);
    okrt_status (*read_accum_f32x4)(okrt_ctx*, float*);
    okrt_status (*set_option)(okrt_ctx*, okrt_option, int64_t);
    okrt_status (*get_counters)(okrt_ctx*, okrt_counters*);
    okrt_status (*reset_counters)(okrt_ctx*);
    okrt_status (*render_frames)(okrt_ctx*, const okrt_render_data*, uint32_t);
    /* ABI 2: the rest of the surface. An implementation may leave entries it does not provide NULL (callers check). */
    okrt_status (*write_accum_f32x4)(okrt_ctx*, const float*);
    okrt_status (*bind_framebuffers)(okrt_ctx*, void*, void*);
    okrt_status (*set_stream)(okrt_ctx*, void*);
    okrt_status (*commit)(okrt_ctx*);
    okrt_status (*get_build_info)(okrt_ctx*, okrt_build_info*);
    okrt_status (*get_fast_structure)(okrt_ctx*, void*, uint32_t, void*, uint32_t);
    okrt_status (*timer_begin)(okrt_ctx*);
    okrt_status (*timer_end)(okrt_ctx*);
    okrt_status (*timer_elapsed_ms)(okrt_ctx*, float*);
    okrt_status (*last_render_ms)(okrt_ctx*, float*);
    okrt_status (*get_stage_ms)(okrt_ctx*, float*, uint32_t*);
    okrt_status (*get_interval_ms)(okrt_ctx*, float*, uint32_t, uint32_t*, uint32_t*, uint32_t*);
    okrt_status (*debug_kat)(okrt_ctx*, okrt_kat, uint32_t, const float*, uint32_t, float*, uint32_t);
    okrt_status (*comm_unique_id)(uint8_t*);
    okrt_status (*comm_create)(okrt_ctx*, const uint8_t*, uint32_t, uint32_t, okrt_comm**);
    okrt_status (*comm_create_all)(okrt_ctx* const*, uint32_t, okrt_comm**);
    okrt_status (*comm_destroy)(okrt_comm*);
    const char* (*comm_last_error)(const okrt_comm*);
    okrt_status (*mgpu_gather_accum)(okrt_ctx*, okrt_comm*, uint32_t, void*);
    okrt_status (*mgpu_gather_accum_all)(okrt_ctx* const*, okrt_comm* const*, uint32_t, uint32_t, void*);
    okrt_status (*mgpu_read_gathered)(okrt_ctx*, okrt_comm*, float*);
    okrt_status (*mgpu_deinterleave)(okrt_ctx*, const void*, uint32_t, uint32_t, uint32_t, void*);
    /* ABI 3 */
    okrt_status (*device_count)(int*);
    const char* (*comm_transport)(const okrt_comm*);
    okrt_status (*mgpu_reduce_accum)(okrt_ctx*, okrt_comm*, uint32_t, void*);
    okrt_status (*mgpu_reduce_accum_all)(okrt_ctx* const*, okrt_comm* const*, uint32_t, uint32_t, void*);
    okrt_status (*peek_result_rgba8)(okrt_ctx*, uint8_t*, uint32_t*);
    okrt_status (*frame_size)(const okrt_ctx*, uint32_t*, uint32_t*);
    /* ABI 4 */
    okrt_status (*comm_exchange_ms)(okrt_comm*, float*);
} okrt_api;

const okrt_api* okrt_get_api(void);

#ifdef __cplusplus
}
#endif
#endif /* OKRT_H */
