/*
 * niagara_vis.h — C ABI of the MI355X-native visibility front-end.
 *
 * Drop-in replacement for the compute passes niagara records between its scene
 * upload and its indirect draws:
 *
 *     drawcull -> tasksubmit -> clustercull -> clustersubmit      (+ depthreduce)
 *
 * Every struct below keeps the byte layout of the reference (citations are
 * relative to the reference tree, file:line).  Every entry point names the
 * reference interface it replaces.  All pointers named d_* are DEVICE pointers
 * owned by the caller; the library only owns scratch inside nv_context.
 *
 * Conventions (mirrors SURVEY.md §8b):
 *   - all functions return 0 on success, a negative NV_E* code or a positive
 *     hipError_t otherwise; nothing throws across this boundary;
 *   - work is enqueued asynchronously on the caller's hipStream_t (passed as
 *     void* so this header needs no HIP include);
 *   - the caller zeroes count words before each pass (src/niagara.cpp:1541,1586)
 *     and zeroes drawVisibility / meshletVisibility once (src/niagara.cpp:1450-1468);
 *   - stage ordering == stream order (replaces the Vulkan barriers at
 *     src/niagara.cpp:1561,1571,1601,1610,1725);
 *   - overflow past NV_TASK_WGLIMIT / NV_CLUSTER_LIMIT is dropped silently,
 *     exactly as drawcull.comp.glsl:128 and clustercull.comp.glsl:137 do.
 *
 * Output order.  The reference appends with unordered global atomics; this
 * library appends in ascending invocation order (draw index; command index then
 * lane), which is one valid serialisation of the reference and makes the lists
 * bit-reproducible.
 */
#ifndef NIAGARA_VIS_H
#define NIAGARA_VIS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compile-time contract (src/config.h:2-28) ---- */
#define NV_TASK_WGSIZE 64u           /* src/config.h:2  */
#define NV_MESH_MAXVTX 64u           /* src/config.h:14 */
#define NV_MESH_MAXTRI 96u           /* src/config.h:15 */
#define NV_CLUSTER_TILE 16u          /* src/config.h:22 */
#define NV_TASK_WGLIMIT (1u << 22)   /* src/config.h:25 */
#define NV_CLUSTER_LIMIT (1u << 24)  /* src/config.h:28 */
#define NV_MAX_LODS 8u               /* src/scene.h:92  */
#define NV_MAX_MIPS 16u              /* sampler maxLod 16: src/resources.cpp:305 */

/* ---- status codes ---- */
#define NV_OK 0
#define NV_EINVAL (-1)   /* bad argument */
#define NV_ENOMEM (-2)   /* scratch allocation failed */
#define NV_ESTATE (-3)   /* reserved (no pass has a device-side protocol any more); never returned */
#define NV_ENODEV (-4)   /* no HIP device */
#define NV_EIO (-5)      /* file cannot be opened / read */
#define NV_EFORMAT (-6)  /* not a scene cache this build understands (magic, version, meshlet limits, truncated) */

/* ---- layouts ---- */

/* src/scene.h:10-23, src/shaders/mesh.h:11-24; 24 B, align 8. cull reads bytes 0-11 only */
typedef struct NvMeshlet
{
	uint16_t center[3]; /* fp16 bits */
	uint16_t radius;    /* fp16 bits */
	int8_t cone_axis[3];
	int8_t cone_cutoff;
	uint32_t dataOffset;
	uint32_t baseVertex;
	uint8_t vertexCount;
	uint8_t triangleCount;
	uint8_t shortRefs;
	uint8_t padding;
} NvMeshlet;

/* src/scene.h:39-49, src/shaders/mesh.h:92-102; 48 B, align 16; orientation is (x,y,z,w) */
typedef struct NvMeshDraw
{
	float position[3];
	float scale;
	float orientation[4];
	uint32_t meshIndex;
	uint32_t meshletVisibilityOffset;
	uint32_t postPass;
	uint32_t materialIndex;
} NvMeshDraw;

/* src/scene.h:68-75, src/shaders/mesh.h:53-60; 20 B */
typedef struct NvMeshLod
{
	uint32_t indexOffset;
	uint32_t indexCount;
	uint32_t meshletOffset;
	uint32_t meshletCount;
	float error;
} NvMeshLod;

/* src/scene.h:77-93, src/shaders/mesh.h:62-78; 208 B, align 16 */
typedef struct NvMesh
{
	float center[3];
	float radius;
	uint32_t vertexOffset;
	uint32_t vertexCount;
	uint32_t ommIndexData;
	uint32_t ommIndexBase;
	uint32_t lodCount;
	uint32_t lodRT;
	uint32_t padding[2];
	NvMeshLod lods[NV_MAX_LODS];
} NvMesh;

/* src/niagara.cpp:227-231, src/shaders/mesh.h:104-114; 24 B */
typedef struct NvMeshDrawCommand
{
	uint32_t drawId;
	uint32_t indexCount;
	uint32_t instanceCount;
	uint32_t firstIndex;
	uint32_t vertexOffset;
	uint32_t firstInstance;
} NvMeshDrawCommand;

/* src/niagara.cpp:233-240, src/shaders/mesh.h:116-123; 20 B */
typedef struct NvMeshTaskCommand
{
	uint32_t drawId;
	uint32_t taskOffset;
	uint32_t taskCount;
	uint32_t lateDrawVisibility;
	uint32_t meshletVisibilityOffset;
} NvMeshTaskCommand;

/* src/niagara.cpp:242-260, src/shaders/mesh.h:26-44; 144 B (136 used), align 16.
 * view is column-major: view[4*col + row]. */
typedef struct NvCullData
{
	float view[16];
	float P00, P11, znear, zfar;
	float frustum[4];
	float lodTarget;
	float pyramidWidth, pyramidHeight;
	uint32_t drawCount;
	int32_t cullingEnabled;
	int32_t lodEnabled;
	int32_t occlusionEnabled;
	int32_t clusterOcclusionEnabled;
	int32_t clusterBackfaceEnabled;
	uint32_t postPass;
	uint32_t _pad[2];
} NvCullData;

/* Replaces the (depthPyramid image, depthSampler) descriptor pair
 * (src/niagara.cpp:1339-1350 image + per-mip views; src/niagara.cpp:629 sampler =
 * LINEAR filter, NEAREST mip, CLAMP_TO_EDGE, MIN reduction).  The mip chain is one
 * linear fp32 buffer; level i is max(1,width>>i) x max(1,height>>i), row-major, at
 * d_base + mipOffset[i] (offsets in floats). */
typedef struct NvPyramidDesc
{
	float* d_base;
	uint32_t width;   /* level-0 width  = previousPow2(depth target width)  */
	uint32_t height;  /* level-0 height = previousPow2(depth target height) */
	uint32_t levels;  /* getImageMipLevels(width, height), src/resources.cpp:280-292 */
	uint32_t mipOffset[NV_MAX_MIPS];
	uint32_t totalTexels;
} NvPyramidDesc;

typedef struct nv_context nv_context;

/* ---- lifetime (replaces createProgram/createComputePipeline, src/niagara.cpp:652-762) ---- */
int nv_create(nv_context** out_ctx, int device);
void nv_destroy(nv_context* ctx);
const char* nv_version(void);
/* Synchronises the stream and returns its state: NV_OK or the HIP error.  (The library has no inter-workgroup waits
 * and therefore no device-side error word any more; NV_ESTATE is kept for ABI stability.) */
int nv_status(nv_context* ctx, void* stream);

/* Kernel-level timing with HIP events recorded on the launch stream (replaces the vkCmdWriteTimestamp pairs around
 * each pass, src/niagara.cpp:1537,1573,1705,1732).  While enabled, nv_clustercull brackets its cull kernel, its
 * occlusion stage (late pass) and its scatter kernel, nv_drawcull and nv_depthreduce their launches.  nv_profile_read synchronises the recorded events,
 * returns the accumulated milliseconds and launch count per slot and resets the accumulators. */
#define NV_PROF_CLUSTER_CULL 0
#define NV_PROF_CLUSTER_SCATTER 1
#define NV_PROF_DRAWCULL 2
#define NV_PROF_DEPTHREDUCE 3
#define NV_PROF_CLUSTER_HIZ 4 /* late pass with HiZ: the occlusion stage between the cull and the scatter kernel */
#define NV_PROF_SLOTS 5
int nv_profile_enable(nv_context* ctx, int enabled);
int nv_profile_read(nv_context* ctx, float out_ms[NV_PROF_SLOTS], uint32_t out_count[NV_PROF_SLOTS]);
/* Which kernel variant the passes took since the last call (the host picks per launch from the previous launches' statistics, or
 * what NV_OPT_CULL_FORM / NV_OPT_CULL_RING / NV_OPT_TASK_EMIT pin): launch counts per variant, so that a profile can say what it
 * timed.  Counted whether or not event profiling is enabled; reading resets the counts. */
#define NV_VARIANT_CULL_FILTER_RING4 0 /* cluster cull launch: conservative filter pass + certified pass, 4-deep ring */
#define NV_VARIANT_CULL_FILTER_RING8 1 /* the same with the 8-deep ring */
#define NV_VARIANT_CULL_DIRECT 2       /* no filter pass, one command per wave */
#define NV_VARIANT_CULL_LANES_BITS 3   /* early pass, one lane per set visibility bit */
#define NV_VARIANT_CULL_LANES 4        /* (0.3: early pass over a cache-resident pool, one lane per valid cluster; never chosen since 0.4) */
#define NV_VARIANT_CULL_AOS 5          /* no SoA mirror registered for the meshlet buffer: records read in place */
#define NV_VARIANT_HIZ_STAGE 6         /* late pass with HiZ: occlusion-stage launches */
#define NV_VARIANT_TASK_LIST 7         /* drawcull TASK scatter: one lane per output command */
#define NV_VARIANT_TASK_PER_DRAW 8     /* drawcull TASK scatter: per-draw form */
#define NV_VARIANT_CULL_DIRECT_PACKED 9 /* no filter pass, windows of 64 valid meshlets per wave iteration (early form without visibility bits) */
#define NV_VARIANT_SLOTS 10
int nv_profile_variants(nv_context* ctx, uint32_t out_count[NV_VARIANT_SLOTS]);

/* ---- options ----
 * NV_OPT_FUSED_COUNT_RESET (default 0): when 1, nv_drawcull and nv_clustercull start their append at index 0 whatever
 * the count word holds, i.e. they absorb the caller's vkCmdFillBuffer(count, 0, 4, 0) (src/niagara.cpp:1541,1586) and
 * the launch boundary that goes with it.  With 0 the reference contract holds: the append starts at the value found in
 * the count word, which the caller zeroes (nv_reset_count or any memset). */
#define NV_OPT_FUSED_COUNT_RESET 1
/* NV_OPT_FUSED_SUBMIT (default 0): when 1, nv_drawcull (task = 1) also leaves what tasksubmit.comp.glsl:27-47 writes — the
 * {X, 64, 1} dispatch words and the zeroed dummy commands up to the next multiple of 64 — and nv_clustercull what
 * clustersubmit.comp.glsl:25-45 writes (the {16, Y, 16} words and the ~0 padding up to the next multiple of 256): the
 * last workgroup of their scatter launches knows the final count.  nv_tasksubmit / nv_clustersubmit can then be
 * skipped (calling them anyway is harmless: they are idempotent); two launches less per frame phase. */
#define NV_OPT_FUSED_SUBMIT 2
/* NV_OPT_CULL_WORKGROUPS_PER_CU (default 6, 1..8): workgroups per CU of nv_clustercull's cull launch.  6 fill a CU and give
 * one pass its shortest time; a caller that keeps several independent passes in flight (contexts on different streams:
 * the cluster passes of several views) gets more throughput from 3, which let two passes' cull launches share the chip
 * (10 M meshlets, three passes in flight: 23.5 instead of 24.7 us per pass; a single pass: 26.3 instead of 25.0 us).
 * Speed only: the results do not depend on it. */
#define NV_OPT_CULL_WORKGROUPS_PER_CU 3
/* NV_OPT_SCATTER_WAVES (default 16; 4 or 8): waves per workgroup of nv_clustercull's scatter launch.  16 (one command per
 * lane) is the shortest launch; 4 leave the other wave slots of every CU to a neighbour pass's cull launch when several
 * passes are in flight (10 M meshlets, three passes in flight: 23.3 instead of 24.8 us per pass; a single pass: 31.5
 * instead of 30.1 us).  Speed only. */
#define NV_OPT_SCATTER_WAVES 4
/* NV_OPT_CULL_FORM (default 0) and NV_OPT_CULL_RING (default 0): pin what nv_clustercull otherwise chooses per launch from two
 * words the PREVIOUS launches of the context left in mapped host memory (frame coherence; read unsynchronised — possibly a launch
 * or two behind).  Results never depend on the choice, speed does, and a captured HIP graph freezes whatever the capture saw;
 * a caller that replays graphs, or wants run-to-run identical timing, pins it:
 *   NV_OPT_CULL_FORM  0 = by statistic (direct form when more than 35 % of the last launch's commands passed the frustum filter),
 *                     1 = always the filter form (sparse passes: few commands have survivors),
 *                     2 = always the direct form (dense passes: the commands of draws drawcull already found visible); an early
 *                         pass with visibility bits then tests one lane per SET BIT (last frame's visible clusters) instead of
 *                         one wave per command,
 *                     3 = as 2, but one wave per command also in the early pass with visibility bits;
 *                     4 = as 3, and one command per wave iteration also where the direct form would walk packed windows of 64 valid
 *                         meshlets (early form without visibility bits: the cluster pass behind drawcull's LOD select, the late pass's
 *                         first stage);
 *                     5 = as 2, but the early pass with visibility bits walks packed windows too (the valid meshlets of the commands that have a
 *                         set bit) instead of testing one lane per set bit;
 *   NV_OPT_CULL_RING  0 = by the last launch's command count, 4 = the 4-deep load ring (passes of a few hundred thousand commands),
 *                     8 = the 8-deep ring (long streams, late passes). */
#define NV_OPT_CULL_FORM 5
#define NV_OPT_CULL_RING 6
/* NV_OPT_TASK_EMIT (default 0): likewise the form in which nv_drawcull (task = 1) writes its MeshTaskCommands — 0 = by the statistic
 * of earlier task passes (list form once a pass emitted more than 4 commands per emitting draw), 1 = always per draw (every lane
 * writes its own draws' commands; fastest for meshes of one or two task groups), 2 = always the list form (one lane per output
 * command; fastest for meshes of many task groups).  Speed only. */
#define NV_OPT_TASK_EMIT 7
/* NV_OPT_DRAW_RECORDS (default 0): the order in which nv_drawcull's EARLY pass requests its inputs — 0 = by the statistic of the last
 * task pass (visibility words first once it emitted from fewer than one draw in eight), 1 = always the records together with the
 * visibility words, 2 = always the visibility words first and then only the records of last frame's visible draws (the others leave
 * at drawcull.comp.glsl:66 whatever their record holds).  Speed only; the late pass decides every draw and reads every record. */
#define NV_OPT_DRAW_RECORDS 8
int nv_set_option(nv_context* ctx, int option, int value);

/* ---- capacities ----
 * The pass entry points only enqueue: none of them allocates, frees or synchronises, so all of them can be recorded into a
 * HIP graph (stream capture) with growing arguments.  nv_create sizes the library's scratch for 1 M draws and
 * NV_TASK_WGLIMIT task commands (the size niagara allocates dcb for, src/niagara.cpp:1070); nv_reserve raises the
 * per-draw scratch to maxDraws before the first pass over that many (it may synchronise the device: call it at scene
 * load, next to createBuffer(dvb), src/niagara.cpp:1060).  nv_drawcull over more draws than were reserved returns
 * NV_ENOMEM and enqueues nothing.  maxCommands is accepted for symmetry; values above NV_TASK_WGLIMIT buy nothing (the
 * passes clamp there like the shaders do). */
int nv_reserve(nv_context* ctx, uint32_t maxDraws, uint32_t maxCommands);

/* Several contexts of ONE device (one per stream: several views or frames in flight) can use one set of scene mirrors:
 * after nv_share_scene(dst, src) the two contexts share what nv_upload_meshlets / nv_upload_meshes / nv_upload_draws /
 * nv_update_draws built or build from then on through either of them (reference-counted; dst's previous mirrors are
 * released; destroying one context leaves the other's mirrors alone).  Scratch (ballots, tile counts, result bytes)
 * stays per context.  An upload through one context is ordered on THAT call's stream only: the caller orders it
 * against passes the sharing contexts have in flight on other streams, as it would for its own buffer uploads. */
int nv_share_scene(nv_context* dst, nv_context* src);

/* ---- scene upload hook (next to uploadBuffer(mlb), src/niagara.cpp:1055) ----
 * Builds the library-owned SoA mirror of the 12 cull bytes of every meshlet
 * (bounds: 4 x fp16 = 8 B, cone: 4 x s8 = 4 B).  nv_clustercull uses the mirror when
 * its d_meshlets argument equals the pointer registered here, and reads the 24-B AoS
 * records directly otherwise.  The registration is by device pointer and the mirror is a
 * snapshot: call again after the buffer's contents change, and before the allocation is
 * freed or reused for something else — (NULL, 0) drops the registration (likewise for
 * nv_upload_meshes).  A re-upload must be ORDERED against the passes in flight over the old
 * contents (same stream, or the caller's event): next to the mirror it rewrites the pool's
 * largest |centre component| / |radius|, which the cull launch's filter margins are derived
 * from — a pass that pairs new records with the old bounds is no longer conservative (only
 * the capacity-growth path synchronises the device by itself). */
int nv_upload_meshlets(nv_context* ctx, void* stream, const NvMeshlet* d_meshlets, uint32_t meshletCount);

/* Upload hook next to uploadBuffer(mb) (src/niagara.cpp:1049): registers the Mesh table's pointer and size.  nv_drawcull
 * stages a registered table of <= 64 meshes in LDS (every draw reads its mesh's bounds, LOD errors and LOD range);
 * unregistered or larger tables are gathered from global memory.  The table is only read at pass time. */
int nv_upload_meshes(nv_context* ctx, void* stream, const NvMesh* d_meshes, uint32_t meshCount);

/* Upload hook next to uploadBuffer(db) (src/niagara.cpp:1052): builds the library-owned mirror of what a draw decision
 * reads.  Everything drawcull.comp.glsl:73-75 computes BEFORE the view transform is view-independent — the world-space
 * sphere {rotateQuat(mesh.center, orientation) * scale + position, mesh.radius * scale} — so it is evaluated once here, in
 * the reference's operation order (the intermediates are bit-identical), and stored as a 16-B stream next to
 * {scale, meshIndex} (8 B) and postPass (4 B).  The decide kernel of nv_drawcull then reads three coalesced streams
 * (28 B per draw instead of 48-B-stride records plus a Mesh gather) and is left with the view transform and the tests.
 * Used when nv_drawcull's d_draws is the pointer registered here or a record inside that buffer with drawCount records
 * behind it (a pass over a shard of the draws) AND its d_meshes is the table given here; the 48-B AoS records are read
 * in place otherwise (same results).  Same registration contract as nv_upload_meshlets ((NULL, 0, NULL) drops it); call
 * again when the Mesh table's bounds change.
 * nv_update_draws re-evaluates [first, first + count) after the caller rewrote those records — the animation path,
 * src/niagara.cpp:1385-1391, which memcpy()s single MeshDraws into the mapped draw buffer every frame. */
int nv_upload_draws(nv_context* ctx, void* stream, const NvMeshDraw* d_draws, uint32_t drawCount, const NvMesh* d_meshes);
int nv_update_draws(nv_context* ctx, void* stream, const NvMeshDraw* d_draws, uint32_t first, uint32_t count);

/* ---- the passes ---- */

/* drawcull.comp.glsl:54-156; dispatch at src/niagara.cpp:1548-1556.
 * late/task are the LATE/TASK specialisation constants (src/niagara.cpp:724-727).
 * d_commands: NvMeshTaskCommand[NV_TASK_WGLIMIT] if task else NvMeshDrawCommand[drawCount].
 * d_count4: {commandCount, groupCountX, groupCountY, groupCountZ} (dccb).
 * pyramid may be NULL unless late && occlusionEnabled. */
int nv_drawcull(nv_context* ctx, void* stream, const NvCullData* cull, int late, int task,
                const NvMeshDraw* d_draws, const NvMesh* d_meshes, void* d_commands, uint32_t* d_count4,
                uint32_t* d_drawVisibility, const NvPyramidDesc* pyramid);

/* vkCmdFillBuffer(dccb / ccb, 0, 4, 0) in front of a pass (src/niagara.cpp:1541,1586): zeroes word 0 of one or two count
 * buffers (either may be NULL).  The passes themselves never clear a counter. */
int nv_reset_count(nv_context* ctx, void* stream, uint32_t* d_count4a, uint32_t* d_count4b);

/* tasksubmit.comp.glsl:27-47; dispatch at src/niagara.cpp:1563-1568 */
int nv_tasksubmit(nv_context* ctx, void* stream, uint32_t* d_count4, NvMeshTaskCommand* d_commands);

/* clustercull.comp.glsl:56-149; indirect dispatch at src/niagara.cpp:1590-1599.
 * The grid is taken on-device from d_count4[1] (groupCountX written by nv_tasksubmit):
 * commands [0, groupCountX*64) are processed, like vkCmdDispatchIndirect(dccb, 4). */
int nv_clustercull(nv_context* ctx, void* stream, const NvCullData* cull, int late,
                   const NvMeshTaskCommand* d_commands, const uint32_t* d_count4, const NvMeshDraw* d_draws,
                   const NvMeshlet* d_meshlets, uint32_t* d_meshletVisibility, const NvPyramidDesc* pyramid,
                   uint32_t* d_clusterIndices, uint32_t* d_clusterCount4);

/* clustersubmit.comp.glsl:25-45; dispatch at src/niagara.cpp:1603-1608 */
int nv_clustersubmit(nv_context* ctx, void* stream, uint32_t* d_clusterCount4, uint32_t* d_clusterIndices);

/* meshlet.task.glsl:53-149 (cull half): same tests as nv_clustercull but compacted per
 * task command into a 64-entry payload, d_payloadCounts[c] = EmitMeshTasksEXT count. */
int nv_taskcull(nv_context* ctx, void* stream, const NvCullData* cull, int late,
                const NvMeshTaskCommand* d_commands, const uint32_t* d_count4, const NvMeshDraw* d_draws,
                const NvMeshlet* d_meshlets, uint32_t* d_meshletVisibility, const NvPyramidDesc* pyramid,
                uint32_t* d_payloads, uint32_t* d_payloadCounts);

/* ---- downstream decode ----
 * What the mesh stage does with the lists before it touches a vertex (src/shaders/meshlet.mesh.glsl:91-116): the indirect
 * grid {16, Y, 16} written by nv_clustersubmit is walked as index = x + 256 y + 16 z, the entry is decoded
 * (~0 = padding -> no outputs; else command = ci & 0xffffff, meshlet = command.taskOffset + (ci >> 24)) and the meshlet
 * header is read.  One 32-byte record per index in [0, 256 Y) and three 64-bit totals (accumulated: zero them first). */
typedef struct NvClusterRecord
{
	uint32_t drawId;       /* ~0 for a padding entry */
	uint32_t meshletIndex; /* mi */
	uint32_t vertexCount, triangleCount; /* SetMeshOutputsEXT arguments */
	uint32_t vertexOffset; /* = dataOffset */
	uint32_t indexOffset;  /* dataOffset + (shortRefs ? (vertexCount + 1) / 2 : vertexCount) */
	uint32_t baseVertex;
	uint32_t shortRefs;
} NvClusterRecord;
int nv_cluster_expand(nv_context* ctx, void* stream, const NvMeshTaskCommand* d_commands, const NvMeshlet* d_meshlets,
                      const uint32_t* d_clusterIndices, const uint32_t* d_clusterCount4, NvClusterRecord* d_records,
                      uint32_t recordCapacity, uint64_t* d_totals3 /* clusters, vertices, triangles */);

/* ---- triangle cull of the mesh stage ----
 * src/shaders/meshlet.mesh.glsl:91-198 with MESH_CULL = 1 (src/config.h:10-11; the reference ships it switched off): for
 * every slot of the consumer's grid the meshlet's vertices are transformed exactly like the mesh shader does
 * (rotateQuat * scale + position, view, projection, perspective divide to screen space) and every triangle gets the
 * shader's gl_CullPrimitiveEXT decision: back-facing / zero-area (:176-181) or missing every sample centre (:183-190),
 * provided all three vertices are in front of the perspective plane (:192-193).
 * One NvTriangleMask per slot index in [0, 256 Y): keep bit i = triangle i is NOT culled (i < triangleCount);
 * counts = triangleCount | vertexCount << 8 | kept << 16; all zero for a padding slot (~0).  Totals are accumulated
 * (zero them first): clusters, triangles, triangles kept. */
typedef struct NvVertex /* src/shaders/mesh.h:3-9, 16 bytes */
{
	uint16_t vx, vy, vz; /* fp16 position */
	uint16_t tp;         /* packed tangent */
	uint32_t np;         /* packed normal */
	uint16_t tu, tv;     /* fp16 texcoord */
} NvVertex;
typedef struct NvGlobals /* src/shaders/mesh.h:46-51, the mesh pipeline's push constants */
{
	float projection[16]; /* column-major */
	NvCullData cullData;  /* only .view is read here */
	float screenWidth, screenHeight;
	float pad_[2];
} NvGlobals;
typedef struct NvTriangleMask
{
	uint32_t keep[3]; /* MESH_MAXTRI = 96 (src/config.h:15) */
	uint32_t counts;
} NvTriangleMask;
/* SURVEY.md §8(f) N2 — meshlet bounds (src/scene.cpp:69-85: meshopt_computeMeshletBounds, meshopt_quantizeHalf,
 * cone_axis_s8 / cone_cutoff_s8).  For every Meshlet of d_meshlets[0, meshletCount) — dataOffset, baseVertex, vertexCount,
 * triangleCount and shortRefs filled in by the caller, the payload in d_meshletData as src/scene.cpp:24-47 packs it, the
 * fp16 positions in d_vertices — computes the bounding sphere and the normal cone of its triangles and writes center[3],
 * radius (fp16) and cone_axis[3], cone_cutoff (s8) into the record (bytes 0-11; the rest is left alone).  d_bounds8
 * (optional) receives the unquantised {center.xyz, radius, axis.xyz, cutoff} per meshlet.  Call nv_upload_meshlets
 * afterwards (a mirror built from these records before is dropped).
 * PARITY UNPINNED: the arithmetic is meshoptimizer's, which the reference does not vendor; the library's published
 * algorithm is restated with defined fp32 semantics here and in the CPU oracle (orc_meshlet_bounds), and the two agree bit for bit. */
int nv_meshlet_bounds(nv_context* ctx, void* stream, const NvVertex* d_vertices, const uint32_t* d_meshletData, NvMeshlet* d_meshlets,
                      uint32_t meshletCount, float* d_bounds8);

/* Buffer sizes: d_meshletData must hold at least one word and d_vertices at least one record even when every slot of d_clusterIndices is ~0 —
 * the lanes of a pass that have no vertex / triangle to fetch read element 0 of both (unconditional loads: no branch between a load and its
 * use), which the reference's mesh shader never touches for an empty slot. */
int nv_trianglecull(nv_context* ctx, void* stream, const NvGlobals* globals, const NvMeshTaskCommand* d_commands,
                    const NvMeshDraw* d_draws, const NvMeshlet* d_meshlets, const uint32_t* d_meshletData,
                    const NvVertex* d_vertices, const uint32_t* d_clusterIndices, const uint32_t* d_clusterCount4,
                    NvTriangleMask* d_masks, uint32_t maskCapacity, uint64_t* d_totals3 /* clusters, triangles, kept */);

/* depthreduce.comp.glsl:14-22 + the level loop at src/niagara.cpp:1703-1733.
 * d_depth is the width x height fp32 depth target (reverse-Z, far = 0). */
int nv_depthreduce(nv_context* ctx, void* stream, const float* d_depth, uint32_t width, uint32_t height,
                   const NvPyramidDesc* pyramid);

/* ---- host helpers mirroring src/niagara.cpp / src/resources.cpp (no device work) ---- */
uint32_t nv_previous_pow2(uint32_t v);                        /* src/niagara.cpp:439-447 */
uint32_t nv_image_mip_levels(uint32_t width, uint32_t height); /* src/resources.cpp:280-292 */
/* fills width/height/levels/mipOffset/totalTexels from the depth target size; d_base untouched */
int nv_pyramid_desc_init(NvPyramidDesc* desc, uint32_t depthWidth, uint32_t depthHeight);
/* the multiplier with which the kernels divide by a launch constant d (grid sizes): mulhi(n, m) >> 7 == n / d for every
 * n < 2^39 / d; 0 (the kernels then divide) for d outside 256 .. 8192.  Exported so that the bound can be tested. */
uint32_t nv_division_magic(uint32_t d);
/* src/niagara.cpp:424-437,1487-1516: CullData from a camera (position, orientation quat xyzw, fovY,
 * znear), viewport, draw distance and pyramid size; flags are left 0 for the caller to set. */
int nv_build_cull_data(NvCullData* out, const float cameraPosition[3], const float cameraOrientation[4],
                       float fovY, float znear, float drawDistance, uint32_t viewportWidth,
                       uint32_t viewportHeight, uint32_t pyramidWidth, uint32_t pyramidHeight,
                       uint32_t drawCount, int debugLodStep);
/* src/niagara.cpp:1002-1020: per-draw meshletVisibilityOffset prefix; returns the slot count
 * through out_slots (meshletVisibility bytes = (slots+31)/32*4) and the postPass mask. */
int nv_assign_visibility_offsets(NvMeshDraw* draws, uint32_t drawCount, const NvMesh* meshes,
                                 uint32_t meshCount, uint32_t* out_slots, uint32_t* out_postPassMask);
/* src/niagara.cpp:449-481,969-998: PCG32-seeded synthetic scene (state 0x42) */
int nv_synth_draws(NvMeshDraw* draws, uint32_t drawCount, uint32_t meshCount, float sceneRadius);

/* src/scene.cpp:207-220: Mesh.center = mean of the positions (fp32, accumulated in array order), Mesh.radius = the
 * largest distance from it — the bounding sphere drawcull tests (the part of the asset pipeline that is the reference's
 * own arithmetic; meshlet bounds are meshoptimizer's).  positions = vertexCount x {x, y, z}. */
int nv_mesh_bounds(const float* positions, uint32_t vertexCount, float out_center[3], float* out_radius);

/* ---- multi-GPU helper: contiguous shard of `total` units for `rank` ---- */
void nv_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t* begin, uint64_t* end);
/* Where the next nv_clustercull calls also leave {0, d_count4[0], final cluster count} as three u64 — the words
 * nv_pack_counts(NULL, d_count4, d_clusterCount4, d_out3) would write, from the scatter launch that computes the count,
 * so a sharded caller feeds its all-reduce without one more launch per pass.  NULL (the default) turns it off. */
int nv_set_counts_sink(nv_context* ctx, uint64_t* d_out3);
/* packs {count4a[0], count4b[0], count4c[0]} (any may be NULL -> 0) into d_out[3] as u64,
 * the payload of the one ncclAllReduce(sum) per phase */
int nv_pack_counts(nv_context* ctx, void* stream, const uint32_t* d_countA, const uint32_t* d_countB,
                   const uint32_t* d_countC, uint64_t* d_out3);

/* ---- scene cache ingestion ----
 * niagara's `.cache` file (src/scenecache.cpp: header :16-55, section order :163-203, load-time checks :273-293) holds
 * the three arrays the visibility path consumes — Meshlet, Mesh, MeshDraw — as RAW struct arrays even in compressed mode
 * (:170, :181, :183); only the vertex / index / meshlet-data / RT-vertex streams are meshopt-coded, and their byte sizes
 * are in the header, so they can be stepped over without a decoder.  nv_scenecache_info validates the header like
 * loadSceneCache does (magic 'SCNC', version 7, MESH_MAXVTX / MESH_MAXTRI; hashMeta, clrtMode and ommStates are
 * returned for the caller to compare with its own settings) and locates the arrays; nv_scenecache_read copies them
 * out (any destination may be NULL).  Host-only: no device work. */
typedef struct NvSceneCacheInfo
{
	uint32_t version, compressed, clrtMode, ommStates;
	uint64_t hashMeta;
	uint32_t meshletMaxVertices, meshletMaxTriangles;
	uint32_t vertexCount, indexCount, meshletCount, meshletdataCount, meshletvtx0Count, meshCount;
	uint32_t materialCount, drawCount, texturePathCount, lightCount, animationCount, keyframeCount;
	float cameraPosition[3];    /* Camera (src/scene.h:111-117) */
	float cameraOrientation[4]; /* quat x, y, z, w */
	float cameraFovY, cameraZnear;
	float sunDirection[3];
	uint64_t fileSize;
	uint64_t vertexOffset, indexOffset, meshletOffset, meshletdataOffset, meshOffset, drawOffset; /* byte offsets in the file */
	uint64_t vertexBytes, indexBytes, meshletdataBytes; /* stored (possibly compressed) sizes of those streams */
} NvSceneCacheInfo;
int nv_scenecache_info(const char* path, NvSceneCacheInfo* out);
int nv_scenecache_read(const char* path, const NvSceneCacheInfo* info, NvMesh* meshes, NvMeshlet* meshlets, NvMeshDraw* draws);

/* ---- verification probe (tests only): per-meshlet scalar intermediates of the cluster cull
 * (view-space centre xyz, radius, cone lhs, cone rhs, aabb[4], mip level, sampled depth,
 * depthSphere, flags) = 16 floats per lane, for the <=1-ULP scalar parity tests. */
int nv_probe_cluster_scalars(nv_context* ctx, void* stream, const NvCullData* cull,
                             const NvMeshTaskCommand* d_commands, uint32_t commandCount,
                             const NvMeshDraw* d_draws, const NvMeshlet* d_meshlets,
                             const NvPyramidDesc* pyramid, float* d_out16);

#ifdef __cplusplus
}

static_assert(sizeof(NvMeshlet) == 24, "Meshlet layout (src/scene.h:10-23)");
static_assert(sizeof(NvMeshDraw) == 48, "MeshDraw layout (src/scene.h:39-49)");
static_assert(sizeof(NvMeshLod) == 20, "MeshLod layout (src/scene.h:68-75)");
static_assert(sizeof(NvMesh) == 208, "Mesh layout (src/scene.h:77-93)");
static_assert(offsetof(NvMesh, lods) == 48, "Mesh.lods offset");
static_assert(sizeof(NvMeshDrawCommand) == 24, "MeshDrawCommand layout (src/niagara.cpp:227-231)");
static_assert(sizeof(NvMeshTaskCommand) == 20, "MeshTaskCommand layout (src/niagara.cpp:233-240)");
static_assert(sizeof(NvCullData) == 144, "CullData layout (src/niagara.cpp:242-260)");
static_assert(offsetof(NvCullData, P00) == 64 && offsetof(NvCullData, frustum) == 80, "CullData offsets");
static_assert(offsetof(NvCullData, lodTarget) == 96 && offsetof(NvCullData, drawCount) == 108, "CullData offsets");
static_assert(offsetof(NvCullData, cullingEnabled) == 112 && offsetof(NvCullData, postPass) == 132, "CullData offsets");
static_assert(sizeof(NvVertex) == 16, "Vertex layout (src/shaders/mesh.h:3-9)");
static_assert(sizeof(NvGlobals) == 224 && offsetof(NvGlobals, cullData) == 64 && offsetof(NvGlobals, screenWidth) == 208, "Globals layout (src/shaders/mesh.h:46-51)");
static_assert(sizeof(NvTriangleMask) == 16, "one mask per grid slot");
static_assert(sizeof(NvSceneCacheInfo) == 208, "NvSceneCacheInfo is mirrored by niagara_amd/_lib.py");
#endif

#endif /* NIAGARA_VIS_H */
