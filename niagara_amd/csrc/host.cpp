// host.cpp — host-side helpers of the C ABI (no device work): the pieces of niagara's main() that prepare what the
// cull passes consume.  Plain C++; arithmetic is fp32 in the order written so that results are reproducible.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/niagara_vis.h"

extern "C" {

// src/niagara.cpp:439-447 — largest power of two strictly below v (1 for v <= 2)
// n / d for a launch constant d: m = floor(2^39 / d) + 1 = 2^39 / d + e with 0 < e <= 1, so n m / 2^39 = n / d + n e / 2^39 and the
// floor is that of n / d while n e / 2^39 < 1 / d, i.e. for n < 2^39 / d; m fits 32 bits for d >= 256 (kernels: args.h)
uint32_t nv_division_magic(uint32_t d)
{
	return d >= 256u && d <= 8192u ? (uint32_t)((1ull << 39) / d) + 1u : 0u;
}

uint32_t nv_previous_pow2(uint32_t v)
{
	uint32_t r = 1;
	while (r * 2 < v)
		r *= 2;
	return r;
}

// src/resources.cpp:280-292
uint32_t nv_image_mip_levels(uint32_t width, uint32_t height)
{
	uint32_t levels = 1;
	while (width > 1 || height > 1)
	{
		++levels;
		width /= 2;
		height /= 2;
	}
	return levels;
}

// src/niagara.cpp:1340-1344 (+ linear mip offsets replacing the per-mip image views at :1346-1350)
int nv_pyramid_desc_init(NvPyramidDesc* desc, uint32_t depthWidth, uint32_t depthHeight)
{
	if (!desc || !depthWidth || !depthHeight)
		return NV_EINVAL;
	desc->width = nv_previous_pow2(depthWidth);
	desc->height = nv_previous_pow2(depthHeight);
	desc->levels = nv_image_mip_levels(desc->width, desc->height);
	if (desc->levels > NV_MAX_MIPS)
		return NV_EINVAL;
	uint32_t offset = 0;
	for (uint32_t i = 0; i < NV_MAX_MIPS; ++i)
	{
		desc->mipOffset[i] = offset;
		if (i < desc->levels)
		{
			uint32_t w = desc->width >> i, h = desc->height >> i;
			offset += (w ? w : 1) * (h ? h : 1);
		}
	}
	desc->totalTexels = offset;
	return NV_OK;
}

// src/niagara.cpp:424-437 (perspectiveProjection, normalizePlane) and :1487-1516 (CullData).
// view = scale(1,1,-1) * inverse(translate(pos) * mat4_cast(q)); the inverse of a rigid transform is taken in
// closed form, [R^T | -(R^T t)] (glm is not vendored in the reference snapshot, so its cofactor inverse cannot be
// matched bit for bit; DESIGN.md lists this as a defined semantic).
int nv_build_cull_data(NvCullData* out, const float cameraPosition[3], const float cameraOrientation[4], float fovY,
                       float znear, float drawDistance, uint32_t viewportWidth, uint32_t viewportHeight,
                       uint32_t pyramidWidth, uint32_t pyramidHeight, uint32_t drawCount, int debugLodStep)
{
	if (!out || !cameraPosition || !cameraOrientation || !viewportWidth || !viewportHeight)
		return NV_EINVAL;
	memset(out, 0, sizeof(*out));

	const float x = cameraOrientation[0], y = cameraOrientation[1], z = cameraOrientation[2], w = cameraOrientation[3];
	const float xx = x * x, yy = y * y, zz = z * z;
	const float xz = x * z, xy = x * y, yz = y * z;
	const float wx = w * x, wy = w * y, wz = w * z;

	// rot[column][row]
	float rot[3][3];
	rot[0][0] = 1.0f - 2.0f * (yy + zz);
	rot[0][1] = 2.0f * (xy + wz);
	rot[0][2] = 2.0f * (xz - wy);
	rot[1][0] = 2.0f * (xy - wz);
	rot[1][1] = 1.0f - 2.0f * (xx + zz);
	rot[1][2] = 2.0f * (yz + wx);
	rot[2][0] = 2.0f * (xz + wy);
	rot[2][1] = 2.0f * (yz - wx);
	rot[2][2] = 1.0f - 2.0f * (xx + yy);

	float* view = out->view;
	for (int col = 0; col < 3; ++col)
		for (int row = 0; row < 3; ++row)
			view[4 * col + row] = rot[row][col]; // transpose
	for (int row = 0; row < 3; ++row)
	{
		float t = (rot[row][0] * cameraPosition[0] + rot[row][1] * cameraPosition[1]) + rot[row][2] * cameraPosition[2];
		view[12 + row] = -t;
	}
	view[15] = 1.0f;
	for (int col = 0; col < 4; ++col)
		view[4 * col + 2] = -view[4 * col + 2]; // scale(1,1,-1) on the left flips row 2

	const float f = 1.0f / tanf(fovY / 2.0f);
	const float aspect = (float)viewportWidth / (float)viewportHeight;
	out->P00 = f / aspect;
	out->P11 = f;
	out->znear = znear;
	out->zfar = drawDistance;

	// normalizePlane(row3 + row0) = (P00,0,1,0)/len, normalizePlane(row3 + row1) = (0,P11,1,0)/len
	const float lenX = sqrtf((out->P00 * out->P00 + 0.0f * 0.0f) + 1.0f * 1.0f);
	const float lenY = sqrtf((0.0f * 0.0f + out->P11 * out->P11) + 1.0f * 1.0f);
	out->frustum[0] = out->P00 / lenX;
	out->frustum[1] = 1.0f / lenX;
	out->frustum[2] = out->P11 / lenY;
	out->frustum[3] = 1.0f / lenY;

	out->lodTarget = (2.0f / out->P11) * (1.0f / (float)viewportHeight) * (float)(1 << debugLodStep);
	out->pyramidWidth = (float)pyramidWidth;
	out->pyramidHeight = (float)pyramidHeight;
	out->drawCount = drawCount;
	return NV_OK;
}

// src/niagara.cpp:1002-1020
int nv_assign_visibility_offsets(NvMeshDraw* draws, uint32_t drawCount, const NvMesh* meshes, uint32_t meshCount,
                                 uint32_t* out_slots, uint32_t* out_postPassMask)
{
	if ((!draws && drawCount) || !meshes)
		return NV_EINVAL;
	uint32_t slots = 0, mask = 0;
	for (uint32_t i = 0; i < drawCount; ++i)
	{
		if (draws[i].meshIndex >= meshCount)
			return NV_EINVAL;
		const NvMesh& mesh = meshes[draws[i].meshIndex];
		draws[i].meshletVisibilityOffset = slots;
		uint32_t widest = 0;
		for (uint32_t l = 0; l < mesh.lodCount && l < NV_MAX_LODS; ++l)
			if (mesh.lods[l].meshletCount > widest)
				widest = mesh.lods[l].meshletCount;
		slots += widest;
		mask |= 1u << draws[i].postPass;
	}
	if (out_slots)
		*out_slots = slots;
	if (out_postPassMask)
		*out_postPassMask = mask;
	return NV_OK;
}

namespace
{
// src/niagara.cpp:449-481
struct Pcg32
{
	uint64_t state, inc;
	uint32_t next()
	{
		uint64_t old = state;
		state = old * 6364136223846793005ULL + (inc | 1);
		uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((32 - rot) & 31));
	}
	double unit() { return next() / double(1ull << 32); }
};
} // namespace

// src/niagara.cpp:969-998 (rngstate.state = 0x42, default inc); the three rand01() of the axis are drawn x, y, z
int nv_synth_draws(NvMeshDraw* draws, uint32_t drawCount, uint32_t meshCount, float sceneRadius)
{
	if ((!draws && drawCount) || !meshCount)
		return NV_EINVAL;
	Pcg32 rng = { 0x42, 0xda3e39cb94b95bdbULL };
	for (uint32_t i = 0; i < drawCount; ++i)
	{
		NvMeshDraw& d = draws[i];
		memset(&d, 0, sizeof(d));
		const uint32_t meshIndex = rng.next() % meshCount;
		for (int k = 0; k < 3; ++k)
			d.position[k] = float(rng.unit()) * sceneRadius * 2 - sceneRadius;
		d.scale = float(rng.unit()) + 1;
		d.scale *= 2;
		float ax = float(rng.unit()) * 2 - 1;
		float ay = float(rng.unit()) * 2 - 1;
		float az = float(rng.unit()) * 2 - 1;
		const float inv = 1.0f / sqrtf((ax * ax + ay * ay) + az * az);
		ax *= inv, ay *= inv, az *= inv;
		const float angle = (float(rng.unit()) * 90.f) * 0.01745329251994329576923690768489f;
		const float s = sinf(angle * 0.5f);
		d.orientation[0] = ax * s;
		d.orientation[1] = ay * s;
		d.orientation[2] = az * s;
		d.orientation[3] = cosf(angle * 0.5f);
		d.meshIndex = meshIndex;
	}
	return NV_OK;
}

// src/scene.cpp:207-220 with glm's component-wise forms: center += v; center /= float(n); radius = max(radius,
// distance(center, v)), distance = sqrt((dx*dx + dy*dy) + dz*dz).  Sequential on purpose: the sum's rounding is part of
// the result.
int nv_mesh_bounds(const float* positions, uint32_t vertexCount, float out_center[3], float* out_radius)
{
	if (!positions || !out_center || !out_radius || vertexCount == 0)
		return NV_EINVAL;
	float cx = 0.0f, cy = 0.0f, cz = 0.0f;
	for (uint32_t i = 0; i < vertexCount; ++i)
	{
		cx += positions[3 * i + 0];
		cy += positions[3 * i + 1];
		cz += positions[3 * i + 2];
	}
	const float n = (float)vertexCount;
	cx /= n;
	cy /= n;
	cz /= n;
	float radius = 0.0f;
	for (uint32_t i = 0; i < vertexCount; ++i)
	{
		const float dx = cx - positions[3 * i + 0], dy = cy - positions[3 * i + 1], dz = cz - positions[3 * i + 2];
		const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
		radius = radius < d ? d : radius; // std::max(radius, d)
	}
	out_center[0] = cx;
	out_center[1] = cy;
	out_center[2] = cz;
	*out_radius = radius;
	return NV_OK;
}

// ---- scene cache (src/scenecache.cpp) ----
// SceneHeader, src/scenecache.cpp:16-55, as the reference's compiler lays it out (vec3 / quat are plain floats; the
// uint64_t makes the struct 8-aligned): 160 bytes.
namespace
{
struct SceneCacheHeader
{
	uint32_t magic, version;
	uint64_t hashMeta;
	uint32_t meshletMaxVertices, meshletMaxTriangles;
	uint8_t clrtMode, compressed, pad[2];
	uint32_t compressedVertexBytes, compressedIndexBytes, compressedMeshletDataBytes, compressedMeshletVtx0Bytes;
	uint32_t vertexCount, indexCount, meshletCount, meshletdataCount, meshletvtx0Count, meshCount;
	uint32_t materialCount, drawCount, texturePathCount, lightCount, animationCount, keyframeCount;
	uint32_t ommArrayDataSize, ommIndexDataSize, ommDescCount, ommStates;
	float cameraPosition[3], cameraOrientation[4], cameraFovY, cameraZnear;
	float sunDirection[3];
	uint32_t tail;
};
static_assert(sizeof(SceneCacheHeader) == 160, "SceneHeader layout (src/scenecache.cpp:16-55)");
static_assert(offsetof(SceneCacheHeader, vertexCount) == 44 && offsetof(SceneCacheHeader, cameraPosition) == 108, "SceneHeader offsets");

const uint32_t kSceneCacheMagic = 0x434E4353u; // 'SCNC', src/scenecache.cpp:12
const uint32_t kSceneCacheVersion = 7;         // src/scenecache.cpp:13
const uint64_t kVertexBytes = 16;              // Vertex, src/scene.h:60-66
const uint64_t kMaterialBytes = 64;            // Material, src/scene.h:25-37
} // namespace

int nv_scenecache_info(const char* path, NvSceneCacheInfo* out)
{
	if (!path || !out)
		return NV_EINVAL;
	FILE* f = fopen(path, "rb");
	if (!f)
		return NV_EIO;
	SceneCacheHeader h;
	const size_t got = fread(&h, 1, sizeof(h), f);
	uint64_t fileSize = 0;
	if (fseek(f, 0, SEEK_END) == 0)
		fileSize = (uint64_t)ftell(f);
	fclose(f);
	// src/scenecache.cpp:276-293 (the caller-dependent fields are handed back instead of compared)
	if (got < sizeof(h) || h.magic != kSceneCacheMagic || h.version != kSceneCacheVersion || h.meshletMaxVertices != NV_MESH_MAXVTX ||
	    h.meshletMaxTriangles != NV_MESH_MAXTRI)
		return NV_EFORMAT;

	memset(out, 0, sizeof(*out));
	out->version = h.version;
	out->compressed = h.compressed;
	out->clrtMode = h.clrtMode;
	out->ommStates = h.ommStates;
	out->hashMeta = h.hashMeta;
	out->meshletMaxVertices = h.meshletMaxVertices;
	out->meshletMaxTriangles = h.meshletMaxTriangles;
	out->vertexCount = h.vertexCount;
	out->indexCount = h.indexCount;
	out->meshletCount = h.meshletCount;
	out->meshletdataCount = h.meshletdataCount;
	out->meshletvtx0Count = h.meshletvtx0Count;
	out->meshCount = h.meshCount;
	out->materialCount = h.materialCount;
	out->drawCount = h.drawCount;
	out->texturePathCount = h.texturePathCount;
	out->lightCount = h.lightCount;
	out->animationCount = h.animationCount;
	out->keyframeCount = h.keyframeCount;
	memcpy(out->cameraPosition, h.cameraPosition, sizeof(h.cameraPosition));
	memcpy(out->cameraOrientation, h.cameraOrientation, sizeof(h.cameraOrientation));
	out->cameraFovY = h.cameraFovY;
	out->cameraZnear = h.cameraZnear;
	memcpy(out->sunDirection, h.sunDirection, sizeof(h.sunDirection));
	out->fileSize = fileSize;

	// section order of saveSceneCache, src/scenecache.cpp:163-186
	uint64_t off = sizeof(h);
	out->vertexOffset = off;
	out->vertexBytes = h.compressed ? h.compressedVertexBytes : (uint64_t)h.vertexCount * kVertexBytes;
	off += out->vertexBytes;
	out->indexOffset = off;
	out->indexBytes = h.compressed ? h.compressedIndexBytes : (uint64_t)h.indexCount * 4u;
	off += out->indexBytes;
	out->meshletOffset = off;
	off += (uint64_t)h.meshletCount * sizeof(NvMeshlet);
	out->meshletdataOffset = off;
	out->meshletdataBytes = h.compressed ? h.compressedMeshletDataBytes : (uint64_t)h.meshletdataCount * 4u;
	off += out->meshletdataBytes;
	off += h.compressed ? h.compressedMeshletVtx0Bytes : (uint64_t)h.meshletvtx0Count * 2u;
	out->meshOffset = off;
	off += (uint64_t)h.meshCount * sizeof(NvMesh);
	off += (uint64_t)h.materialCount * kMaterialBytes;
	out->drawOffset = off;
	off += (uint64_t)h.drawCount * sizeof(NvMeshDraw);
	if (off > fileSize)
		return NV_EFORMAT; // truncated
	return NV_OK;
}

int nv_scenecache_read(const char* path, const NvSceneCacheInfo* info, NvMesh* meshes, NvMeshlet* meshlets, NvMeshDraw* draws)
{
	if (!path || !info)
		return NV_EINVAL;
	FILE* f = fopen(path, "rb");
	if (!f)
		return NV_EIO;
	int rc = NV_OK;
	const struct
	{
		void* dst;
		uint64_t offset, bytes;
	} parts[3] = { { meshlets, info->meshletOffset, (uint64_t)info->meshletCount * sizeof(NvMeshlet) },
		           { meshes, info->meshOffset, (uint64_t)info->meshCount * sizeof(NvMesh) },
		           { draws, info->drawOffset, (uint64_t)info->drawCount * sizeof(NvMeshDraw) } };
	for (int i = 0; i < 3 && rc == NV_OK; ++i)
	{
		if (!parts[i].dst || !parts[i].bytes)
			continue;
		if (fseek(f, (long)parts[i].offset, SEEK_SET) != 0 || fread(parts[i].dst, 1, parts[i].bytes, f) != parts[i].bytes)
			rc = NV_EIO;
	}
	fclose(f);
	return rc;
}

// contiguous ranges, remainder spread over the first ranks
void nv_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t* begin, uint64_t* end)
{
	if (!world)
		world = 1;
	uint64_t base = total / world, rem = total % world;
	uint64_t b = base * rank + (rank < rem ? rank : rem);
	uint64_t e = b + base + (rank < rem ? 1 : 0);
	if (begin)
		*begin = b;
	if (end)
		*end = e;
}

} // extern "C"
