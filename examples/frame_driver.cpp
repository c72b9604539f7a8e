// frame_driver.cpp — the reference's frame order (src/niagara.cpp:1530-1611, 1703-1733, 1765-1788) driven from C++ through
// the C ABI alone: no Python, no torch; device memory and streams come from the HIP runtime, like a renderer would own them.
//
//   frame_driver <scene.bin> <out.bin> <frames> [fused] [time <N>]
//
// scene.bin (little endian, written by tests/test_frame_driver.py from the same scenes the parity tests use):
//   u32 magic 'NVSC', meshCount, meshletCount, drawCount, viewportWidth, viewportHeight
//   NvCullData (flags already set) | NvMesh[] | NvMeshlet[] | NvMeshDraw[] (offsets not yet assigned) | float depth[h*w]
// out.bin: a sequence of records {u32 tag, u32 bytes, payload} in the order the passes produce them; the test builds the same
// sequence from the oracle and compares the two files byte for byte.
//
// Per frame: early drawcull<TASK> -> tasksubmit -> clustercull -> clustersubmit; pyramid; the same four late; and, when a draw
// of the scene carries a postPass bit >= 1 (`meshPostPasses >> 1`, src/niagara.cpp:1017,1781), the post phase
// cull(late, postPass = 1) + render(late, postPass = 1) (:1781-1787).  Frame 0 reduces a cleared depth target (the reference's
// first frame has nothing rendered yet), later frames the scene's depth.
//
// `time N`: after the recorded frames, N more frames are enqueued back to back without reading anything back (the frame loop a
// renderer would run) and the wall time per frame is printed as one JSON line on stdout; then one last frame is recorded into
// out.bin like the first ones, so the state the timed frames left is checked too.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
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
		fprintf(stderr, "usage: %s scene.bin out.bin frames [fused] [time N]\n", argv[0]);
		return 1;
	}
	const int frames = atoi(argv[3]);
	bool fused = false;
	int timedFrames = 0;
	for (int i = 4; i < argc; ++i)
	{
		if (strcmp(argv[i], "fused") == 0)
			fused = true;
		else if (strcmp(argv[i], "time") == 0 && i + 1 < argc)
			timedFrames = atoi(argv[++i]);
	}

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
	// the passes drop output at the reference's limits (TASK_WGLIMIT commands, CLUSTER_LIMIT indices), so larger buffers buy nothing
	taskCapacity = std::min<size_t>((taskCapacity + 63) / 64 * 64, NV_TASK_WGLIMIT) + 64;
	const size_t clusterCapacity = std::min<size_t>(taskCapacity * 64, NV_CLUSTER_LIMIT) + 256;

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

	CHECK_NV(nv_reserve(ctx, drawCount, (uint32_t)std::min<size_t>(taskCapacity, NV_TASK_WGLIMIT))); // next to the buffer creation: passes never allocate
	CHECK_NV(nv_upload_meshes(ctx, stream, mb, meshCount));
	CHECK_NV(nv_upload_meshlets(ctx, stream, mlb, meshletCount));
	CHECK_NV(nv_upload_draws(ctx, stream, db, drawCount, mb)); // the draw mirror (world spheres): next to uploadBuffer(db), src/niagara.cpp:1052
	CHECK_NV(nv_set_option(ctx, NV_OPT_FUSED_COUNT_RESET, fused));
	CHECK_NV(nv_set_option(ctx, NV_OPT_FUSED_SUBMIT, fused));

	FILE* out = fopen(argv[2], "wb");
	if (!out)
	{
		perror(argv[2]);
		return 1;
	}

	const NvCullData cullData = cullIn[0];
	const bool postPhase = (postMask >> 1) != 0; // src/niagara.cpp:1781
	int frameIndex = 0;

	// one phase of the frame: cull() + the cluster branch of render()
	auto phase = [&](int late, uint32_t postPass)
	{
		// cull(): src/niagara.cpp:1530-1574
		NvCullData cullPass = cullData;
		cullPass.clusterBackfaceEnabled = postPass == 0; // :1549
		cullPass.postPass = postPass;
		if (!fused)
			CHECK_NV(nv_reset_count(ctx, stream, dccb, nullptr));
		CHECK_NV(nv_drawcull(ctx, stream, &cullPass, late, 1, db, mb, dcb, dccb, dvb, &pyramid));
		if (!fused)
			CHECK_NV(nv_tasksubmit(ctx, stream, dccb, dcb));

		// render(), cluster branch: src/niagara.cpp:1582-1611
		NvCullData renderPass = cullData;
		renderPass.postPass = postPass; // :1595-1596
		if (!fused)
			CHECK_NV(nv_reset_count(ctx, stream, ccb, nullptr));
		CHECK_NV(nv_clustercull(ctx, stream, &renderPass, late, dcb, dccb, db, mlb, mvb, &pyramid, cib, ccb));
		if (!fused)
			CHECK_NV(nv_clustersubmit(ctx, stream, ccb, cib));
	};

	auto record = [&](const char* name)
	{
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
		printf("frame %d %s: %u task commands, %u visible clusters\n", frameIndex, name, c4[0], cc4[0]);
	};

	// src/niagara.cpp:1765-1788; recorded = every buffer is read back after every phase
	auto frame = [&](bool recorded)
	{
		phase(0, 0);
		if (recorded)
			record("early");
		CHECK_NV(nv_depthreduce(ctx, stream, frameIndex > 0 ? depthTarget : clearedDepth, width, height, &pyramid));
		if (recorded)
			emit(out, TAG_PYRAMID, pyramid.d_base, (size_t)pyramid.totalTexels * 4, stream);
		phase(1, 0);
		if (recorded)
			record("late ");
		if (postPhase)
		{
			phase(1, 1);
			if (recorded)
				record("post ");
		}
		++frameIndex;
	};

	for (int i = 0; i < frames; ++i)
		frame(true);

	if (timedFrames > 0)
	{
		for (int i = 0; i < 3; ++i)
			frame(false);
		CHECK_HIP(hipStreamSynchronize(stream));
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < timedFrames; ++i)
			frame(false);
		CHECK_HIP(hipStreamSynchronize(stream));
		const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / timedFrames;
		printf("{\"frame_driver\": \"timed\", \"frames\": %d, \"frame_us\": %.2f, \"draws\": %u, \"phases_per_frame\": %d, \"fused\": %s}\n", timedFrames, us, drawCount,
		       postPhase ? 3 : 2, fused ? "true" : "false");
		frame(true); // the state the timed frames left, checked like the first frames
	}
	fclose(out);

	nv_destroy(ctx);
	CHECK_HIP(hipStreamDestroy(stream));
	return 0;
}
