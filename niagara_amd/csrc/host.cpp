// host.cpp — host-side helpers of the C ABI (no device work): the pieces of niagara's main() that prepare what the
// cull passes consume.  Plain C++; arithmetic is fp32 in the order written so that results are reproducible.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/niagara_vis.h"

extern "C" {

// src/niagara.cpp:439-447 — largest power of two strictly below v (1 for v <= 2)
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
