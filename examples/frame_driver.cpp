// frame_driver.cpp — the reference's frame order (src/niagara.cpp:1530-1611, 1703-1733, 1765-1788) driven from C++ through
// the C ABI alone: no Python, no torch; device memory and streams come from the HIP runtime, like a renderer would own them.
//
//   frame_driver <scene.bin> <out.bin> <frames> [fused]
//
// scene.bin (little endian, written by tests/test_frame_driver.py from the same scenes the parity tests use):
//   u32 magic 'NVSC', meshCount, meshletCount, drawCount, viewportWidth, viewportHeight
//   NvCullData (flags already set) | NvMesh[] | NvMeshlet[] | NvMeshDraw[] (offsets not yet assigned) | float depth[h*w]
// out.bin: a sequence of records {u32 tag, u32 bytes, payload} in the order the passes produce them; the test builds the same
// sequence from the oracle and compares the two files byte for byte.
//
// Per frame: early drawcull<TASK> -> tasksubmit -> clustercull -> clustersubmit; pyramid; the same four late.  Frame 0 reduces
// a cleared depth target (the reference's first frame has nothing rendered yet), later frames the scene's depth.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/niagara_vis.h"

#define CHECK_HIP(x)                                                                         \
	do                                                                                       \
	{                                                                                        \
		hipError_t e_ = (x);                                                                 \
		if (e_ != hipSuccess)                                                                \
		{                                                                                    \
			fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(2);                                                                         \
		}                                                                                    \
	} while (0)

#define CHECK_NV(x)                                                    \
	do                                                                 \
	{                                                                  \
		int e_ = (x);                                                  \
		if (e_ != 0)                                                   \
		{                                                              \
			fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, e_); \
			exit(3);                                                   \
		}                                                              \
	} while (0)

template <typename T>
static T* deviceArray(size_t count, const T* init = nullptr)
{
	T* p = nullptr;
	CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)));
	if (init && count)
		CHECK_HIP(hipMemcpy(p, init, count * sizeof(T), hipMemcpyHostToDevice));
	else
		CHECK_HIP(hipMemset(p, 0, (count ? count : 1) * sizeof(T)));
	return p;
}

template <typename T>
static void readArray(FILE* f, std::vector<T>& v, size_t count)
{
	v.resize(count);
	if (count && fread(v.data(), sizeof(T), count, f) != count)
	{
		fprintf(stderr, "scene file is truncated\n");
		exit(1);
	}
}

static void emit(FILE* out, uint32_t tag, const void* d_ptr, size_t bytes, hipStream_t stream)
{
	std::vector<uint8_t> host(bytes);
	if (bytes)
		CHECK_HIP(hipMemcpyAsync(host.data(), d_ptr, bytes, hipMemcpyDeviceToHost, stream));
	CHECK_HIP(hipStreamSynchronize(stream));
	const uint32_t head[2] = { tag, (uint32_t)bytes };
	fwrite(head, 4, 2, out);
	fwrite(host.data(), 1, bytes, out);
}

enum Tag : uint32_t
{
	TAG_PYRAMID = 1,
	TAG_COUNT4 = 2,
	TAG_COMMANDS = 3,
	TAG_CC4 = 4,
	TAG_CIB = 5,
	TAG_DVB = 6,
	TAG_MVB = 7,
};

int main(int argc, char** argv)
{
	if (argc < 4)
	{
		fprintf(stderr, "usage: %s scene.bin out.bin frames [fused]\n", argv[0]);
		return 1;
	}
	const int frames = atoi(argv[3]);
	const bool fused = argc > 4 && strcmp(argv[4], "fused") == 0;

	FILE* in = fopen(argv[1], "rb");
	if (!in)
	{
		perror(argv[1]);
		return 1;
	}
	uint32_t head[6];
	if (fread(head, 4, 6, in) != 6 || head[0] != 0x4353564eu)
	{
		fprintf(stderr, "not a scene file\n");
		return 1;
	}
	const uint32_t meshCount = head[1], meshletCount = head[2], drawCount = head[3], width = head[4], height = head[5];
	std::vector<NvCullData> cullIn;
	std::vector<NvMesh> meshes;
	std::vector<NvMeshlet> meshlets;
	std::vector<NvMeshDraw> draws;
	std::vector<float> depth;
	readArray(in, cullIn, 1);
	readArray(in, meshes, meshCount);
	readArray(in, meshlets, meshletCount);
	readArray(in, draws, drawCount);
	readArray(in, depth, (size_t)width * height);
	fclose(in);

	// src/niagara.cpp:1002-1020: visibility slots per draw; the task-command capacity is what the draws can emit at their largest LOD
	uint32_t slots = 0, postMask = 0;
	CHECK_NV(nv_assign_visibility_offsets(draws.data(), drawCount, meshes.data(), meshCount, &slots, &postMask));
	size_t taskCapacity = 64;
	for (const NvMeshDraw& d : draws)
	{
		const NvMesh& mesh = meshes[d.meshIndex];
		uint32_t groups = 0;
		for (uint32_t l = 0; l < mesh.lodCount; ++l)
			groups = std::max(groups, (mesh.lods[l].meshletCount + NV_TASK_WGSIZE - 1) / NV_TASK_WGSIZE);
		taskCapacity += groups;
	}
	taskCapacity = (taskCapacity + 63) / 64 * 64 + 64;
	const size_t clusterCapacity = taskCapacity * 64 + 256;

	nv_context* ctx = nullptr;
	CHECK_NV(nv_create(&ctx, 0));
	hipStream_t stream;
	CHECK_HIP(hipStreamCreate(&stream));

	NvMesh* mb = deviceArray(meshCount, meshes.data());
	NvMeshlet* mlb = deviceArray(meshletCount, meshlets.data());
	NvMeshDraw* db = deviceArray(drawCount, draws.data());
	float* depthTarget = deviceArray(depth.size(), depth.data());
	float* clearedDepth = deviceArray<float>(depth.size());
	uint32_t* dvb = deviceArray<uint32_t>(drawCount);             // zeroed once (src/niagara.cpp:1450-1457)
	const size_t mvbWords = (slots + 31) / 32 + 2;
	uint32_t* mvb = deviceArray<uint32_t>(mvbWords);              // zeroed once (src/niagara.cpp:1459-1468)
	NvMeshTaskCommand* dcb = deviceArray<NvMeshTaskCommand>(taskCapacity);
	uint32_t* dccb = deviceArray<uint32_t>(4);
	uint32_t* cib = deviceArray<uint32_t>(clusterCapacity);
	uint32_t* ccb = deviceArray<uint32_t>(4);

	NvPyramidDesc pyramid;
	CHECK_NV(nv_pyramid_desc_init(&pyramid, width, height));
	pyramid.d_base = deviceArray<float>(pyramid.totalTexels);

	CHECK_NV(nv_upload_meshlets(ctx, stream, mlb, meshletCount));
	CHECK_NV(nv_upload_meshes(ctx, stream, mb, meshCount));
	CHECK_NV(nv_upload_draws(ctx, stream, db, drawCount, mb)); // the draw mirror (world spheres): next to uploadBuffer(db), src/niagara.cpp:1052
	CHECK_NV(nv_set_option(ctx, NV_OPT_FUSED_COUNT_RESET, fused));
	CHECK_NV(nv_set_option(ctx, NV_OPT_FUSED_SUBMIT, fused));

	FILE* out = fopen(argv[2], "wb");
	if (!out)
	{
		perror(argv[2]);
		return 1;
	}

	NvCullData cull = cullIn[0];
	for (int frame = 0; frame < frames; ++frame)
	{
		for (int late = 0; late < 2; ++late)
		{
			if (late)
			{
				CHECK_NV(nv_depthreduce(ctx, stream, frame > 0 ? depthTarget : clearedDepth, width, height, &pyramid));
				emit(out, TAG_PYRAMID, pyramid.d_base, (size_t)pyramid.totalTexels * 4, stream);
			}

			// cull(): src/niagara.cpp:1530-1574
			cull.postPass = 0;
			if (!fused)
				CHECK_NV(nv_reset_count(ctx, stream, dccb, nullptr));
			CHECK_NV(nv_drawcull(ctx, stream, &cull, late, 1, db, mb, dcb, dccb, dvb, &pyramid));
			if (!fused)
				CHECK_NV(nv_tasksubmit(ctx, stream, dccb, dcb));

			// render(), cluster branch: src/niagara.cpp:1582-1611
			if (!fused)
				CHECK_NV(nv_reset_count(ctx, stream, ccb, nullptr));
			CHECK_NV(nv_clustercull(ctx, stream, &cull, late, dcb, dccb, db, mlb, mvb, &pyramid, cib, ccb));
			if (!fused)
				CHECK_NV(nv_clustersubmit(ctx, stream, ccb, cib));
			CHECK_NV(nv_status(ctx, stream));

			uint32_t c4[4], cc4[4];
			CHECK_HIP(hipMemcpy(c4, dccb, 16, hipMemcpyDeviceToHost));
			CHECK_HIP(hipMemcpy(cc4, ccb, 16, hipMemcpyDeviceToHost));
			emit(out, TAG_COUNT4, dccb, 16, stream);
			emit(out, TAG_COMMANDS, dcb, (size_t)c4[1] * 64 * sizeof(NvMeshTaskCommand), stream);
			emit(out, TAG_CC4, ccb, 16, stream);
			emit(out, TAG_CIB, cib, ((size_t)cc4[0] + 255) / 256 * 256 * 4, stream);
			emit(out, TAG_DVB, dvb, (size_t)drawCount * 4, stream);
			emit(out, TAG_MVB, mvb, mvbWords * 4, stream);
			printf("frame %d %s: %u task commands, %u visible clusters\n", frame, late ? "late " : "early", c4[0], cc4[0]);
		}
	}
	fclose(out);

	nv_destroy(ctx);
	CHECK_HIP(hipStreamDestroy(stream));
	return 0;
}
