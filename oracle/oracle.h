/*
 * oracle.h — CPU restatement of niagara's visibility passes.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (niagara_amd/, include/) never links, imports or calls anything here.
 *
 * Parity status: PINNED against the reference's own shader sources executed on the CPU
 * (oracle/_ref, built by oracle/Makefile from the .glsl files in /root/reference/src/shaders through
 * oracle/ref_translate.py + oracle/glsl_shim.h) and against the fixtures generated from
 * that build in tests/golden/.  The reference ships no tests or golden vectors of its own
 * (SURVEY.md §4, §8c).  Semantics the GLSL leaves to the implementation are DEFINED here:
 * fp32 everywhere, left-to-right evaluation as written, no FMA contraction, IEEE sqrt and
 * divide, exact ceil(log2)/exp2 via exponent bits, MIN-reduction sampler = min over the
 * non-zero-weight texels of the fp32 bilinear footprint with clamp-to-edge and nearest mip,
 * append order = ascending invocation index.
 */
#ifndef NIAGARA_ORACLE_H
#define NIAGARA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
	uint16_t center[3];
	uint16_t radius;
	int8_t cone_axis[3];
	int8_t cone_cutoff;
	uint32_t dataOffset;
	uint32_t baseVertex;
	uint8_t vertexCount;
	uint8_t triangleCount;
	uint8_t shortRefs;
	uint8_t padding;
} OrcMeshlet; /* src/scene.h:10-23 */

typedef struct
{
	float position[3];
	float scale;
	float orientation[4]; /* x y z w */
	uint32_t meshIndex;
	uint32_t meshletVisibilityOffset;
	uint32_t postPass;
	uint32_t materialIndex;
} OrcMeshDraw; /* src/scene.h:39-49 */

typedef struct
{
	uint32_t indexOffset, indexCount, meshletOffset, meshletCount;
	float error;
} OrcMeshLod; /* src/scene.h:68-75 */

typedef struct
{
	float center[3];
	float radius;
	uint32_t vertexOffset, vertexCount, ommIndexData, ommIndexBase;
	uint32_t lodCount, lodRT, padding[2];
	OrcMeshLod lods[8];
} OrcMesh; /* src/scene.h:77-93 */

typedef struct
{
	uint32_t drawId, indexCount, instanceCount, firstIndex, vertexOffset, firstInstance;
} OrcMeshDrawCommand; /* src/niagara.cpp:227-231 */

typedef struct
{
	uint32_t drawId, taskOffset, taskCount, lateDrawVisibility, meshletVisibilityOffset;
} OrcMeshTaskCommand; /* src/niagara.cpp:233-240 */

typedef struct
{
	float view[16]; /* column-major */
	float P00, P11, znear, zfar;
	float frustum[4];
	float lodTarget;
	float pyramidWidth, pyramidHeight;
	uint32_t drawCount;
	int32_t cullingEnabled, lodEnabled, occlusionEnabled, clusterOcclusionEnabled, clusterBackfaceEnabled;
	uint32_t postPass;
	uint32_t _pad[2];
} OrcCullData; /* src/niagara.cpp:242-260 */

typedef struct
{
	float* base;
	uint32_t width, height, levels;
	uint32_t mipOffset[16];
	uint32_t totalTexels;
} OrcPyramid;

/* scalar pieces (src/shaders/math.h) */
void orc_rotate_quat(const float v[3], const float q[4], float out[3]);                 /* math.h:46-49 */
int orc_project_sphere(const float c[3], float r, float znear, float P00, float P11, float aabb[4]); /* math.h:2-22 */
float orc_occlusion_mip(const float aabb[4], float pw, float ph);                      /* math.h:24-39 */
int orc_cone_cull(const float c[3], float r, const float axis[3], float cutoff);       /* math.h:41-44, camera at 0 */
float orc_half_to_float(uint16_t h);
float orc_sample_min(const OrcPyramid* p, float u, float v, float level);               /* textureLod w/ MIN sampler */
float orc_sample_min_image(const float* img, uint32_t w, uint32_t h, float u, float v); /* texture() on one level */

/* host helpers (src/niagara.cpp, src/resources.cpp) */
uint32_t orc_previous_pow2(uint32_t v);                         /* niagara.cpp:439-447 */
uint32_t orc_image_mip_levels(uint32_t w, uint32_t h);          /* resources.cpp:280-292 */
void orc_pyramid_init(OrcPyramid* p, uint32_t depthW, uint32_t depthH); /* niagara.cpp:1340-1344 */
void orc_build_cull_data(OrcCullData* out, const float camPos[3], const float camQuat[4], float fovY, float znear,
                         float drawDistance, uint32_t vw, uint32_t vh, uint32_t pw, uint32_t ph, uint32_t drawCount,
                         int debugLodStep);                     /* niagara.cpp:424-437,1487-1516 */
void orc_assign_visibility_offsets(OrcMeshDraw* draws, uint32_t n, const OrcMesh* meshes, uint32_t* slots,
                                   uint32_t* postMask);         /* niagara.cpp:1002-1020 */
uint32_t orc_pcg32(uint64_t* state, uint64_t inc);              /* niagara.cpp:460-469 */
void orc_synth_draws(OrcMeshDraw* draws, uint32_t n, uint32_t meshCount, float sceneRadius); /* niagara.cpp:969-998 */

/* the passes; all pointers are host memory */
void orc_drawcull(const OrcCullData* cull, int late, int task, const OrcMeshDraw* draws, const OrcMesh* meshes,
                  void* commands, uint32_t* count4, uint32_t* drawVisibility, const OrcPyramid* pyr); /* drawcull.comp.glsl:54-156 */
void orc_tasksubmit(uint32_t* count4, OrcMeshTaskCommand* commands);                                /* tasksubmit.comp.glsl:27-47 */
void orc_clustercull(const OrcCullData* cull, int late, const OrcMeshTaskCommand* commands, const uint32_t* count4,
                     const OrcMeshDraw* draws, const OrcMeshlet* meshlets, uint32_t* meshletVisibility,
                     const OrcPyramid* pyr, uint32_t* clusterIndices, uint32_t* clusterCount4);       /* clustercull.comp.glsl:56-149 */
void orc_clustersubmit(uint32_t* clusterCount4, uint32_t* clusterIndices);                          /* clustersubmit.comp.glsl:25-45 */
void orc_taskcull(const OrcCullData* cull, int late, const OrcMeshTaskCommand* commands, const uint32_t* count4,
                  const OrcMeshDraw* draws, const OrcMeshlet* meshlets, uint32_t* meshletVisibility,
                  const OrcPyramid* pyr, uint32_t* payloads, uint32_t* payloadCounts);               /* meshlet.task.glsl:53-149 */
void orc_depthreduce(const float* depth, uint32_t w, uint32_t h, const OrcPyramid* pyr);            /* depthreduce.comp.glsl:14-22 + niagara.cpp:1703-1733 */

/* src/shaders/meshlet.mesh.glsl:91-116: decode of the cluster list by its consumer; records = 8 x u32 per
 * index in [0, cc4[1]*cc4[2]*cc4[3]), totals3 += {clusters, vertices, triangles} */
void orc_cluster_expand(const OrcMeshTaskCommand* commands, const OrcMeshlet* meshlets, const uint32_t* clusterIndices,
                        const uint32_t* cc4, uint32_t* records8, uint32_t capacity, uint64_t* totals3);

/* src/shaders/mesh.h:3-9 */
typedef struct
{
	uint16_t vx, vy, vz;
	uint16_t tp;
	uint32_t np;
	uint16_t tu, tv;
} OrcVertex;

/* src/shaders/mesh.h:46-51 (push constants of the mesh pipeline; size rounded up to the struct's 16-byte alignment) */
typedef struct
{
	float projection[16];
	OrcCullData cullData;
	float screenWidth, screenHeight;
	float pad_[2];
} OrcGlobals;

/* src/shaders/meshlet.mesh.glsl:91-198 with MESH_CULL = 1: per grid slot 4 words = keep[3] (bit i: triangle i survives),
 * counts = triangleCount | vertexCount << 8 | kept << 16; totals3 += {clusters, triangles, kept}.
 * GLSL round() leaves the direction of .5 to the implementation: defined here as round-half-to-even (what the
 * hardware's v_rndne / the C rintf in the default rounding mode do). */
void orc_trianglecull(const OrcGlobals* globals, const OrcMeshTaskCommand* commands, const OrcMeshDraw* draws, const OrcMeshlet* meshlets,
                      const uint32_t* meshletData, const OrcVertex* vertices, const uint32_t* clusterIndices, const uint32_t* cc4,
                      uint32_t* masks4, uint32_t capacity, uint64_t* totals3);

/* per-meshlet scalar intermediates, 16 floats per lane (same record as nv_probe_cluster_scalars) */
void orc_probe_cluster_scalars(const OrcCullData* cull, const OrcMeshTaskCommand* commands, uint32_t commandCount,
                               const OrcMeshDraw* draws, const OrcMeshlet* meshlets, const OrcPyramid* pyr, float* out16);

/* meshlet bounding sphere + normal cone + fp16 / s8 quantisation (src/scene.cpp:69-85).  PARITY
 * UNPINNED: meshoptimizer's published algorithm restated (see oracle.c); the vertex array is the reference's Vertex
 * (fp16 positions), the data array its packed vertex references + index bytes */
void orc_meshlet_bounds(const OrcVertex* vertices, const uint32_t* meshletData, OrcMeshlet* meshlets, uint32_t count, float* out8);

/* multi-threaded (OpenMP) forms used only as the CPU baseline; identical output */
int orc_max_threads(void);
void orc_clustercull_mt(const OrcCullData* cull, int late, const OrcMeshTaskCommand* commands, const uint32_t* count4,
                        const OrcMeshDraw* draws, const OrcMeshlet* meshlets, uint32_t* meshletVisibility,
                        const OrcPyramid* pyr, uint32_t* clusterIndices, uint32_t* clusterCount4, int threads);
void orc_drawcull_mt(const OrcCullData* cull, int late, int task, const OrcMeshDraw* draws, const OrcMesh* meshes,
                     void* commands, uint32_t* count4, uint32_t* drawVisibility, const OrcPyramid* pyr, int threads);

#ifdef __cplusplus
}
#endif
#endif
