/*
 * ref_runner.cpp — runs the reference's translated shader main()s on the CPU (TEST INFRASTRUCTURE).
 *
 * Compiled once per shader by oracle/Makefile with -DREF_<SHADER> -DREF_GEN="<generated header>".  The generated
 * header is the reference's own source text (see ref_translate.py); this file only supplies what Vulkan would:
 * descriptor binding (pointer assignment), specialisation constants, push constants and the invocation loops,
 * serialised in ascending workgroup / local id order.
 */
#include "glsl_shim.h"

#include <stddef.h>

namespace glsl
{
static uvec3 gl_GlobalInvocationID, gl_LocalInvocationID, gl_WorkGroupID;
#ifdef REF_MESHLET_MESH
/* GL_EXT_mesh_shader built-ins the mesh stage writes (the runner reads them back after each workgroup) */
static uint gl_LocalInvocationIndex;
struct MeshVertexOut
{
	vec4 gl_Position;
};
struct MeshPrimitiveOut
{
	bool gl_CullPrimitiveEXT;
};
static MeshVertexOut gl_MeshVerticesEXT[256];
static uvec3 gl_PrimitiveTriangleIndicesEXT[256];
static MeshPrimitiveOut gl_MeshPrimitivesEXT[256];
static uint ref_outVertices, ref_outPrimitives;
static void SetMeshOutputsEXT(uint vertexCount, uint primitiveCount)
{
	ref_outVertices = vertexCount;
	ref_outPrimitives = primitiveCount;
}
static void barrier() {} /* the runner executes the workgroup twice instead (see ref_meshlet_mesh) */
#endif

#ifdef REF_MESHLET_TASK
/* GL_EXT_mesh_shader built-ins of the task stage.  Invocations are serialised in ascending order, so the last one's
 * EmitMeshTasksEXT carries the workgroup's final sharedCount; barrier() is honoured by the translator's --once rewrite of
 * the shared counter's initialisation (oracle/Makefile) and by that ordering. */
static uint ref_emitCount;
static void EmitMeshTasksEXT(uint x, uint, uint) { ref_emitCount = x; }
static void barrier() {}
#endif

namespace REF_NS
{
#include REF_GEN
}
} // namespace glsl

using namespace glsl;
using namespace glsl::REF_NS;

struct RefPyramid
{
	const float* base;
	uint32_t width, height, levels;
	uint32_t mipOffset[16];
	uint32_t totalTexels;
};

#if defined(REF_DRAWCULL) || defined(REF_CLUSTERCULL) || defined(REF_MESHLET_TASK)
static texture2D bind_pyramid(const RefPyramid* p)
{
	texture2D t = { 0, 0, 0, 0, 0 };
	if (p)
	{
		t.base = p->base;
		t.width = p->width;
		t.height = p->height;
		t.levels = p->levels;
		t.mipOffset = p->mipOffset;
	}
	return t;
}
#endif

#ifdef REF_DRAWCULL
/* vkCmdDispatch(ceil(drawCount/64)) of drawcull.comp.glsl, local_size_x = 64 (src/niagara.cpp:1548-1556) */
extern "C" void ref_drawcull(const void* cull, int late, int task, void* draws_, void* meshes_, void* commands, uint32_t* count4,
                             uint32_t* dvb, const RefPyramid* pyr)
{
	cullData = *(const CullData*)cull;
	LATE = late != 0;
	TASK = task != 0;
	draws = (MeshDraw*)draws_;
	meshes = (Mesh*)meshes_;
	drawCommands = (MeshDrawCommand*)commands;
	taskCommands = (MeshTaskCommand*)commands;
	CommandCount_buf = (CommandCount_t*)count4;
	drawVisibility = dvb;
	depthPyramid = bind_pyramid(pyr);

	uint groups = (cullData.drawCount + 63) / 64;
	for (uint g = 0; g < groups; ++g)
		for (uint l = 0; l < 64; ++l)
		{
			gl_WorkGroupID.x = g;
			gl_LocalInvocationID.x = l;
			gl_GlobalInvocationID.x = g * 64 + l;
			shader_main();
		}
}

/* layout pinning + scalar helpers of src/shaders/math.h:2-49 */
extern "C" uint32_t ref_sizeof(int what)
{
	switch (what)
	{
	case 0: return sizeof(Meshlet);
	case 1: return sizeof(MeshDraw);
	case 2: return sizeof(MeshLod);
	case 3: return sizeof(Mesh);
	case 4: return sizeof(MeshDrawCommand);
	case 5: return sizeof(MeshTaskCommand);
	case 6: return sizeof(CullData);
	case 7: return offsetof(Mesh, lods);
	case 8: return offsetof(CullData, P00);
	case 9: return offsetof(CullData, frustum);
	case 10: return offsetof(CullData, lodTarget);
	case 11: return offsetof(CullData, drawCount);
	case 12: return offsetof(CullData, cullingEnabled);
	case 13: return offsetof(CullData, postPass);
	case 14: return offsetof(MeshDraw, orientation);
	case 15: return offsetof(MeshDraw, meshIndex);
	case 16: return offsetof(Meshlet, cone_axis);
	case 17: return offsetof(Meshlet, dataOffset);
	case 18: return TASK_WGSIZE;
	case 19: return TASK_WGLIMIT;
	case 20: return CLUSTER_LIMIT;
	case 21: return CLUSTER_TILE;
	case 22: return TASK_CULL;
	}
	return 0;
}
extern "C" void ref_rotate_quat(const float v[3], const float q[4], float out[3])
{
	vec3 r = rotateQuat(vec3(v[0], v[1], v[2]), vec4(q[0], q[1], q[2], q[3]));
	out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
extern "C" int ref_project_sphere(const float c[3], float r, float znear, float P00, float P11, float aabb[4])
{
	vec4 a;
	bool ok = projectSphere(vec3(c[0], c[1], c[2]), r, znear, P00, P11, a);
	aabb[0] = a.x, aabb[1] = a.y, aabb[2] = a.z, aabb[3] = a.w;
	return ok;
}
extern "C" float ref_occlusion_mip(const float aabb[4], float pw, float ph)
{
	return getOcclusionMip(vec4(aabb[0], aabb[1], aabb[2], aabb[3]), pw, ph);
}
extern "C" int ref_cone_cull(const float c[3], float r, const float axis[3], float cutoff)
{
	return coneCull(vec3(c[0], c[1], c[2]), r, vec3(axis[0], axis[1], axis[2]), cutoff, vec3(0, 0, 0));
}
#endif

#ifdef REF_TASKSUBMIT
/* vkCmdDispatch(1,1,1) of tasksubmit.comp.glsl (src/niagara.cpp:1563-1568) */
extern "C" void ref_tasksubmit(uint32_t* count4, void* commands)
{
	CommandCount_buf = (CommandCount_t*)count4;
	taskCommands = (MeshTaskCommand*)commands;
	for (uint l = 0; l < 64; ++l)
	{
		gl_LocalInvocationID.x = l;
		gl_GlobalInvocationID.x = l;
		shader_main();
	}
}
#endif

#ifdef REF_CLUSTERCULL
/* vkCmdDispatchIndirect(dccb, 4) of clustercull.comp.glsl: grid (count4[1], count4[2], count4[3]) (src/niagara.cpp:1590-1599) */
extern "C" void ref_clustercull(const void* cull, int late, void* commands, const uint32_t* count4, void* draws_, void* meshlets_,
                                uint32_t* mvb, const RefPyramid* pyr, uint32_t* cib, uint32_t* cc4)
{
	cullData = *(const CullData*)cull;
	LATE = late != 0;
	taskCommands = (MeshTaskCommand*)commands;
	draws = (MeshDraw*)draws_;
	meshlets = (Meshlet*)meshlets_;
	meshletVisibility = mvb;
	depthPyramid = bind_pyramid(pyr);
	clusterIndices = cib;
	ClusterCount_buf = (ClusterCount_t*)cc4;

	for (uint gx = 0; gx < count4[1]; ++gx)
		for (uint gy = 0; gy < count4[2]; ++gy)
			for (uint l = 0; l < 64; ++l)
			{
				gl_WorkGroupID.x = gx;
				gl_WorkGroupID.y = gy;
				gl_LocalInvocationID.x = l;
				shader_main();
			}
}
#endif

#ifdef REF_CLUSTERSUBMIT
/* vkCmdDispatch(1,1,1) of clustersubmit.comp.glsl (src/niagara.cpp:1603-1608) */
extern "C" void ref_clustersubmit(uint32_t* cc4, uint32_t* cib)
{
	ClusterCount_buf = (ClusterCount_t*)cc4;
	clusterIndices = cib;
	for (uint l = 0; l < 256; ++l)
	{
		gl_LocalInvocationID.x = l;
		shader_main();
	}
}
#endif

#ifdef REF_DEPTHREDUCE
/* the level loop of src/niagara.cpp:1713-1728 around depthreduce.comp.glsl (local size 32x32) */
extern "C" void ref_depthreduce(const float* depth, uint32_t w, uint32_t h, float* base, uint32_t pw, uint32_t ph, uint32_t levels,
                                const uint32_t* mipOffset)
{
	static const uint zero = 0;
	texture2D src = { depth, w, h, 1, &zero };
	for (uint i = 0; i < levels; ++i)
	{
		uint lw = pw >> i, lh = ph >> i;
		lw = lw ? lw : 1;
		lh = lh ? lh : 1;
		image2D dst = { base + mipOffset[i], lw, lh };
		outImage = dst;
		inImage = src;
		imageSize = vec2((float)lw, (float)lh);
		/* dispatch() rounds the grid up to whole 32x32 groups (src/niagara.cpp:213-225); the image store of an
		 * out-of-range texel is discarded by Vulkan, so only in-range invocations are run here */
		for (uint y = 0; y < lh; ++y)
			for (uint x = 0; x < lw; ++x)
			{
				gl_GlobalInvocationID.x = x;
				gl_GlobalInvocationID.y = y;
				shader_main();
			}
		texture2D next = { dst.data, lw, lh, 1, &zero };
		src = next;
	}
}
#endif

#ifdef REF_HOST
/* verbatim host helpers: src/niagara.cpp:424-481 (projection, normalizePlane, previousPow2, PCG32),
 * src/resources.cpp:280-292 (getImageMipLevels) */
extern "C" uint32_t ref_previous_pow2(uint32_t v) { return previousPow2(v); }
extern "C" uint32_t ref_image_mip_levels(uint32_t w, uint32_t h) { return getImageMipLevels(w, h); }
extern "C" void ref_rng_seed(uint64_t state) { rngstate.state = state; }
extern "C" uint32_t ref_rand32(void) { return rand32(); }
extern "C" double ref_rand01(void) { return rand01(); }
extern "C" void ref_perspective(float fovY, float aspect, float znear, float out16[16], float frustum[4])
{
	mat4 projection = perspectiveProjection(fovY, aspect, znear);
	for (int c = 0; c < 4; ++c)
	{
		out16[4 * c + 0] = projection[c].x;
		out16[4 * c + 1] = projection[c].y;
		out16[4 * c + 2] = projection[c].z;
		out16[4 * c + 3] = projection[c].w;
	}
	/* src/niagara.cpp:1494-1497,1506-1509 (statements restated: they live inside main()) */
	mat4 projectionT = transpose(projection);
	vec4 frustumX = normalizePlane(projectionT[3] + projectionT[0]);
	vec4 frustumY = normalizePlane(projectionT[3] + projectionT[1]);
	frustum[0] = frustumX.x;
	frustum[1] = frustumX.z;
	frustum[2] = frustumY.y;
	frustum[3] = frustumY.z;
}
#endif

#ifdef REF_HOST
#include <algorithm>
#include <vector>
/* src/scene.cpp:207-220 compiled verbatim inside a function that supplies the names those statements use: `positions`
 * (the de-quantised vertex positions), `vertices` (only its size), `mesh` (center, radius) */
extern "C" void ref_mesh_bounds(const float* xyz, uint32_t count, float out_center[3], float* out_radius)
{
	std::vector<vec3> positions(count);
	for (uint32_t i = 0; i < count; ++i)
		positions[i] = vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
	std::vector<int> vertices(count);
	struct
	{
		vec3 center;
		float radius;
	} mesh;
#include "_ref/host_bounds.gen.h"
	out_center[0] = mesh.center.x;
	out_center[1] = mesh.center.y;
	out_center[2] = mesh.center.z;
	*out_radius = mesh.radius;
}
#endif

#ifdef REF_MESHLET_MESH
/* vkCmdDrawMeshTasksIndirectEXT(ccb, 4) of meshlet.mesh.glsl without a task stage (TASK = false): one workgroup of
 * MESH_WGSIZE = 64 invocations per grid cell {x < cc4[1], y < cc4[2], z < cc4[3]} (src/niagara.cpp:1664).  The shader
 * has one barrier() between its vertex and triangle phases; invocations are serialised here, so every workgroup is run
 * TWICE: the first run leaves vertexClip[] complete, the second run's triangle phase then reads what a real
 * workgroup would read after the barrier (main() has no other cross-invocation state and only overwrites its outputs). */
extern "C" void ref_meshlet_mesh(const void* globals_, void* commands, void* draws_, void* meshlets_, uint32_t* meshletData_, void* vertices_,
                                 uint32_t* clusterIndices_, const uint32_t* cc4, uint32_t* masks4, uint32_t capacity, uint64_t* totals3)
{
	/* push-constant block in the GLSL (std430) layout: mat4 @0, CullData @64 (144 bytes: a struct that contains a mat4 is
	 * 16-byte aligned, the shim's plain-float CullData is 136), screenWidth / screenHeight @208 */
	memcpy(&globals.projection, globals_, 64);
	memcpy(&globals.cullData, (const char*)globals_ + 64, sizeof(CullData));
	memcpy(&globals.screenWidth, (const char*)globals_ + 208, 4);
	memcpy(&globals.screenHeight, (const char*)globals_ + 212, 4);
	TASK = false;
	taskCommands = (MeshTaskCommand*)commands;
	draws = (MeshDraw*)draws_;
	meshlets = (Meshlet*)meshlets_;
	meshletData = meshletData_;
	meshletData16 = (uint16_t*)meshletData_;
	meshletData8 = (uint8_t*)meshletData_;
	vertices = (Vertex*)vertices_;
	clusterIndices = clusterIndices_;
	for (uint y = 0; y < cc4[2]; ++y)
		for (uint z = 0; z < cc4[3]; ++z)
			for (uint x = 0; x < cc4[1]; ++x)
			{
				gl_WorkGroupID.x = x;
				gl_WorkGroupID.y = y;
				gl_WorkGroupID.z = z;
				for (int run = 0; run < 2; ++run)
					for (uint l = 0; l < 64; ++l)
					{
						gl_LocalInvocationIndex = l;
						gl_LocalInvocationID.x = l;
						shader_main();
					}
				uint32_t index = x + y * 256 + z * CLUSTER_TILE;
				uint32_t out[4] = { 0, 0, 0, 0 };
				if (ref_outVertices || ref_outPrimitives || clusterIndices_[index] != ~0u)
				{
					uint32_t kept = 0;
					for (uint i = 0; i < ref_outPrimitives; ++i)
						if (!gl_MeshPrimitivesEXT[i].gl_CullPrimitiveEXT)
						{
							out[i >> 5] |= 1u << (i & 31);
							kept++;
						}
					out[3] = (ref_outPrimitives & 0xffu) | (ref_outVertices & 0xffu) << 8 | kept << 16;
					totals3[0] += 1;
					totals3[1] += ref_outPrimitives;
					totals3[2] += kept;
				}
				if (index < capacity)
					memcpy(masks4 + (size_t)index * 4, out, sizeof(out));
			}
}
#endif

#ifdef REF_MESHLET_TASK
/* vkCmdDrawMeshTasksIndirectEXT(dccb, 4) of meshlet.task.glsl: grid (count4[1], count4[2] = 64, 1), one workgroup of
 * TASK_WGSIZE = 64 invocations per task command (src/niagara.cpp:1660).  Per workgroup the payload's first
 * EmitMeshTasksEXT-count entries and that count are handed back. */
extern "C" void ref_meshlet_task(const void* cull, int late, void* commands, const uint32_t* count4, void* draws_, void* meshlets_, uint32_t* mvb,
                                 const RefPyramid* pyr, uint32_t* payloads, uint32_t* payloadCounts)
{
	memcpy(&globals.cullData, cull, sizeof(CullData));
	LATE = late != 0;
	taskCommands = (MeshTaskCommand*)commands;
	draws = (MeshDraw*)draws_;
	meshlets = (Meshlet*)meshlets_;
	meshletVisibility = mvb;
	depthPyramid = bind_pyramid(pyr);
	for (uint gx = 0; gx < count4[1]; ++gx)
		for (uint gy = 0; gy < count4[2]; ++gy)
		{
			ref_emitCount = 0;
			for (uint l = 0; l < 64; ++l)
			{
				gl_WorkGroupID.x = gx;
				gl_WorkGroupID.y = gy;
				gl_LocalInvocationID.x = l;
				shader_main();
			}
			const uint commandId = gx * 64 + gy;
			for (uint i = 0; i < ref_emitCount; ++i)
				payloads[(size_t)commandId * 64 + i] = payload.clusterIndices[i];
			payloadCounts[commandId] = ref_emitCount;
		}
}
#endif
