// shard_driver.cpp — the sharded cluster pass of SURVEY.md §8(e) / BASELINE config 5 as ONE C++ process on the C ABI and RCCL:
// `ncclCommInitAll` over the visible devices, one HIP stream + one nv_context per device, every device culls its contiguous
// range of task commands (nv_shard_range) with nv_clustercull, and the phase's counts {0, task commands, visible meshlets} —
// written by the scatter launch itself through nv_set_counts_sink — are summed with ONE ncclAllReduce(ncclSum) of 3 x u64 per
// phase over xGMI.  No meshlet data is exchanged; the reference has no multi-GPU path at all (one VkPhysicalDevice,
// src/device.cpp:190-248), so this host is new, not a replacement.
//
//   shard_driver [--devices N] [--shards S] [--total-meshlets T | --draws D] [--commands-per-draw C] [--steps K] [--warmup W] [--dump PREFIX]
//
//   --total-meshlets T   strong scaling: the pool of T meshlets (T / 64 full task commands) is split over the devices
//                        (100000000 on 8 devices = BASELINE config 5: 12.5 M meshlets each)
//   --draws D            weak scaling: D draws x C commands x 64 meshlets PER DEVICE (default 15625 x 10 = config 3A per device)
//   --dump PREFIX        writes PREFIX.scene (draws, per-device meshlets and commands) and PREFIX.out (per-device global ID lists,
//                        local and reduced counts) for tests/test_shard_driver.py, which holds them against the CPU oracle
//
// The synthetic pool follows SURVEY.md §8(d) config 3: draws from niagara's generator (nv_synth_draws = src/niagara.cpp:969-998),
// meshlet bounds drawn from a seeded PCG32 stream per device (centre U[-1,1]^3 and radius U[0.02,0.1] as fp16, cone axis = random
// unit vector as s8, cutoff U{0..127}).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

#define CHECK_NV(x)                                                          \
	do                                                                       \
	{                                                                        \
		int e_ = (x);                                                        \
		if (e_ != 0)                                                         \
		{                                                                    \
			fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, e_); \
			exit(3);                                                         \
		}                                                                    \
	} while (0)

#define CHECK_NCCL(x)                                                                          \
	do                                                                                         \
	{                                                                                          \
		ncclResult_t e_ = (x);                                                                 \
		if (e_ != ncclSuccess)                                                                 \
		{                                                                                      \
			fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, ncclGetErrorString(e_)); \
			exit(4);                                                                           \
		}                                                                                      \
	} while (0)

namespace
{

struct Pcg32
{
	uint64_t state, inc;
	uint32_t next()
	{
		const uint64_t old = state;
		state = old * 6364136223846793005ull + inc;
		const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		const uint32_t rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((32u - rot) & 31u));
	}
	float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); } // [0, 1)
};

// fp32 -> fp16 bits, round to nearest even (the values here are normal halfs or zero: |x| in [2^-14, 2])
uint16_t half_bits(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	const uint32_t sign = (u >> 16) & 0x8000u;
	const int32_t exp = (int32_t)((u >> 23) & 0xff) - 127 + 15;
	uint32_t man = u & 0x7fffffu;
	if (exp <= 0)
	{
		if (exp < -10)
			return (uint16_t)sign;
		man |= 0x800000u;
		const uint32_t shift = (uint32_t)(14 - exp);
		uint32_t h = man >> shift;
		const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
		if (rem > halfway || (rem == halfway && (h & 1u)))
			++h;
		return (uint16_t)(sign | h);
	}
	if (exp >= 31)
		return (uint16_t)(sign | 0x7c00u);
	uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
	const uint32_t rem = man & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
		++h;
	return (uint16_t)(sign | h);
}

void make_meshlets(std::vector<NvMeshlet>& out, size_t count, uint64_t seed)
{
	Pcg32 rng{ 0x42ull + seed * 0x9e3779b97f4a7c15ull, (0xda3e39cb94b95bdbull + 2 * seed) | 1ull };
	out.resize(count);
	for (size_t i = 0; i < count; ++i)
	{
		NvMeshlet m;
		memset(&m, 0, sizeof(m));
		for (int k = 0; k < 3; ++k)
			m.center[k] = half_bits(rng.unit() * 2.0f - 1.0f);
		m.radius = half_bits(0.02f + 0.08f * rng.unit());
		// random direction: reject points outside the unit ball, normalise
		float x, y, z, l2;
		do
		{
			x = rng.unit() * 2.0f - 1.0f;
			y = rng.unit() * 2.0f - 1.0f;
			z = rng.unit() * 2.0f - 1.0f;
			l2 = x * x + y * y + z * z;
		} while (l2 > 1.0f || l2 < 1e-4f);
		const float inv = 127.0f / sqrtf(l2);
		m.cone_axis[0] = (int8_t)lrintf(x * inv);
		m.cone_axis[1] = (int8_t)lrintf(y * inv);
		m.cone_axis[2] = (int8_t)lrintf(z * inv);
		m.cone_cutoff = (int8_t)(rng.next() & 127u);
		m.vertexCount = 64;
		m.triangleCount = 96;
		out[i] = m;
	}
}

struct Device
{
	int id;  // rank
	int dev; // HIP device the rank runs on (= id unless --shards oversubscribes the devices)
	hipStream_t stream;
	nv_context* ctx;
	uint64_t cmdBegin, cmdEnd; // the device's range of the pool's commands
	uint32_t drawBegin, drawCount;
	std::vector<NvMeshlet> meshletsHost;
	std::vector<NvMeshTaskCommand> commandsHost;
	NvMeshDraw* db;
	NvMeshlet* mlb;
	NvMeshTaskCommand* dcb;
	uint32_t* dccb;
	uint32_t* cib;
	uint32_t* ccb;
	uint64_t* counts;  // {0, task commands, visible meshlets}: written by the scatter launch, all-reduced in place
	uint64_t* local;   // copy of the device's own counts before the reduction (test output)
	NvCullData cull;
};

template <typename T>
T* deviceArray(size_t count, const T* init = nullptr)
{
	T* p = nullptr;
	CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)));
	if (init && count)
		CHECK_HIP(hipMemcpy(p, init, count * sizeof(T), hipMemcpyHostToDevice));
	else
		CHECK_HIP(hipMemset(p, 0, (count ? count : 1) * sizeof(T)));
	return p;
}

} // namespace

int main(int argc, char** argv)
{
	int wantDevices = 0, wantShards = 0, steps = 50, warmup = 5;
	uint64_t totalMeshlets = 0;
	uint32_t drawsPerDevice = 15625, cpd = 10;
	std::string dump;
	for (int i = 1; i < argc; ++i)
	{
		const bool more = i + 1 < argc;
		if (!strcmp(argv[i], "--devices") && more)
			wantDevices = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--shards") && more)
			wantShards = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--total-meshlets") && more)
			totalMeshlets = strtoull(argv[++i], nullptr, 10);
		else if (!strcmp(argv[i], "--draws") && more)
			drawsPerDevice = (uint32_t)atoi(argv[++i]);
		else if (!strcmp(argv[i], "--commands-per-draw") && more)
			cpd = (uint32_t)atoi(argv[++i]);
		else if (!strcmp(argv[i], "--steps") && more)
			steps = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--warmup") && more)
			warmup = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--dump") && more)
			dump = argv[++i];
		else
		{
			fprintf(stderr, "usage: %s [--devices N] [--shards S] [--total-meshlets T | --draws D] [--commands-per-draw C] [--steps K] [--warmup W] [--dump PREFIX]\n", argv[0]);
			return 1;
		}
	}
	if (cpd == 0 || steps < 1)
		return 1;

	int visible = 0;
	if (hipGetDeviceCount(&visible) != hipSuccess || visible == 0)
	{
		fprintf(stderr, "no HIP device\n");
		return 5;
	}
	const int D = wantDevices > 0 && wantDevices <= visible ? wantDevices : visible;
	// --shards S > devices: S ranks dealt round-robin over the D devices — the sharding, the per-rank passes and the ID rebasing of an
	// S-GPU node on fewer GPUs (config 5's eight shards on a one-GPU box).  RCCL refuses the same device twice in one communicator
	// (ncclCommInitAll: "duplicate GPU"), so the communicator always has ONE rank per device and the ranks that share a device are
	// summed into the device's first rank on the host before the collective (and handed the result after it): the communicator set-up,
	// the grouped ncclAllReduce over `comms` and everything up to it are the SAME statements in both modes — on an 8-GPU node the only
	// difference is that ncclCommInitAll's device list and the group loop have eight entries instead of one.
	if (wantShards > 0 && wantShards < D)
		fprintf(stderr, "shard_driver: --shards %d is fewer than the %d devices in use: ignored (one rank per device; use --devices %d)\n", wantShards, D, wantShards);
	const int N = wantShards > D ? wantShards : D;
	const bool oversubscribed = N > D;
	const uint64_t totalCmd = totalMeshlets ? totalMeshlets / 64 : (uint64_t)drawsPerDevice * cpd * N;
	const uint32_t totalDraws = (uint32_t)((totalCmd + cpd - 1) / cpd);

	// ---- the communicator: one rank per device, one process (ncclCommInitAll)
	std::vector<int> devlist(D);
	for (int r = 0; r < D; ++r)
		devlist[r] = r;
	std::vector<ncclComm_t> comms(D);
	CHECK_NCCL(ncclCommInitAll(comms.data(), D, devlist.data()));

	// ---- the pool: draws replicated (each device keeps the slice its commands reference), commands + meshlets sharded
	std::vector<NvMeshDraw> draws(totalDraws);
	CHECK_NV(nv_synth_draws(draws.data(), totalDraws, 1, 300.0f));
	const float camPos[3] = { 0, 0, 0 }, camRot[4] = { 0, 0, 0, 1 };

	std::vector<Device> devs(N);
	for (int r = 0; r < N; ++r)
	{
		Device& d = devs[r];
		d.id = r;
		d.dev = r % D;
		CHECK_HIP(hipSetDevice(d.dev));
		CHECK_HIP(hipStreamCreate(&d.stream));
		CHECK_NV(nv_create(&d.ctx, d.dev));
		CHECK_NV(nv_set_option(d.ctx, NV_OPT_FUSED_COUNT_RESET, 1));
		nv_shard_range(totalCmd, (uint32_t)r, (uint32_t)N, &d.cmdBegin, &d.cmdEnd);
		const uint64_t n = d.cmdEnd - d.cmdBegin;
		d.drawBegin = (uint32_t)(d.cmdBegin / cpd);
		d.drawCount = (uint32_t)((d.cmdEnd + cpd - 1) / cpd) - d.drawBegin;
		if (n == 0)
			d.drawCount = 0;
		make_meshlets(d.meshletsHost, (size_t)n * 64, 2 + (uint64_t)r);
		// full commands, padded with zeroed dummy commands to a multiple of 64 like tasksubmit leaves them; drawId is local to the
		// device's draw slice, meshlets and visibility slots are the device's own
		d.commandsHost.assign((size_t)(n + 63) / 64 * 64, NvMeshTaskCommand{ 0, 0, 0, 0, 0 });
		for (uint64_t k = 0; k < n; ++k)
		{
			NvMeshTaskCommand& c = d.commandsHost[k];
			c.drawId = (uint32_t)((d.cmdBegin + k) / cpd) - d.drawBegin;
			c.taskOffset = (uint32_t)(k * 64);
			c.taskCount = 64;
			c.lateDrawVisibility = 0;
			c.meshletVisibilityOffset = (uint32_t)(k * 64);
		}
		d.db = deviceArray(d.drawCount, draws.data() + d.drawBegin);
		d.mlb = deviceArray(d.meshletsHost.size(), d.meshletsHost.data());
		d.dcb = deviceArray(d.commandsHost.size(), d.commandsHost.data());
		// what tasksubmit leaves in dccb for n commands (tasksubmit.comp.glsl:30-38)
		const uint32_t clamped = (uint32_t)(n < NV_TASK_WGLIMIT ? n : NV_TASK_WGLIMIT);
		const uint32_t c4[4] = { (uint32_t)n, (clamped + 63) / 64 < 65535u ? (clamped + 63) / 64 : 65535u, 64, 1 };
		d.dccb = deviceArray<uint32_t>(4, c4);
		const size_t cibCap = ((size_t)n * 64 < NV_CLUSTER_LIMIT ? (size_t)n * 64 : NV_CLUSTER_LIMIT) + 256;
		d.cib = deviceArray<uint32_t>(cibCap);
		d.ccb = deviceArray<uint32_t>(4);
		d.counts = deviceArray<uint64_t>(3);
		d.local = deviceArray<uint64_t>(3);
		CHECK_NV(nv_upload_meshlets(d.ctx, d.stream, d.mlb, (uint32_t)d.meshletsHost.size()));
		CHECK_NV(nv_set_counts_sink(d.ctx, d.counts));
		CHECK_NV(nv_build_cull_data(&d.cull, camPos, camRot, 70.0f * 3.14159265358979f / 180.0f, 0.1f, 200.0f, 1024, 768, 512, 512, d.drawCount, 0));
		d.cull.cullingEnabled = 1;
		d.cull.clusterBackfaceEnabled = 1;
	}

	// one phase: every device culls its shard, then ONE all-reduce of the phase's counts (grouped: one call per rank of this process)
	auto phase = [&](bool keepLocal)
	{
		for (Device& d : devs)
		{
			CHECK_HIP(hipSetDevice(d.dev));
			CHECK_NV(nv_clustercull(d.ctx, d.stream, &d.cull, 0, d.dcb, d.dccb, d.db, d.mlb, nullptr, nullptr, d.cib, d.ccb));
			if (keepLocal)
				CHECK_HIP(hipMemcpyAsync(d.local, d.counts, 24, hipMemcpyDeviceToDevice, d.stream));
		}
		if (oversubscribed)
		{
			// more ranks than devices: the ranks of one device are summed into its first rank (rank id < D) on the host — functional
			// stand-in for the extra communicator ranks RCCL refuses; synchronises every phase
			for (int dv = 0; dv < D; ++dv)
			{
				uint64_t sum[3] = { 0, 0, 0 }, one[3];
				CHECK_HIP(hipSetDevice(dv));
				for (int r = dv; r < N; r += D)
				{
					CHECK_HIP(hipStreamSynchronize(devs[r].stream));
					CHECK_HIP(hipMemcpy(one, devs[r].counts, 24, hipMemcpyDeviceToHost));
					for (int k = 0; k < 3; ++k)
						sum[k] += one[k];
				}
				CHECK_HIP(hipMemcpy(devs[dv].counts, sum, 24, hipMemcpyHostToDevice));
			}
		}
		// ONE ncclAllReduce(ncclSum) of 3 x u64 per device and phase, grouped (one call per rank of the communicator, all from this process)
		CHECK_NCCL(ncclGroupStart());
		for (int dv = 0; dv < D; ++dv)
			CHECK_NCCL(ncclAllReduce(devs[dv].counts, devs[dv].counts, 3, ncclUint64, ncclSum, comms[dv], devs[dv].stream));
		CHECK_NCCL(ncclGroupEnd());
		if (oversubscribed)
		{
			for (int r = D; r < N; ++r) // the device's other ranks receive what its first rank holds
			{
				CHECK_HIP(hipSetDevice(devs[r].dev));
				CHECK_HIP(hipStreamSynchronize(devs[r % D].stream));
				CHECK_HIP(hipMemcpyAsync(devs[r].counts, devs[r % D].counts, 24, hipMemcpyDeviceToDevice, devs[r].stream));
			}
		}
	};
	auto syncAll = [&]()
	{
		for (Device& d : devs)
		{
			CHECK_HIP(hipSetDevice(d.dev));
			CHECK_HIP(hipStreamSynchronize(d.stream));
		}
	};

	for (int i = 0; i < warmup; ++i)
		phase(false);
	syncAll();
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < steps; ++i)
		phase(i == steps - 1);
	syncAll();
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	// every rank holds the same sums
	std::vector<uint64_t> reduced(3 * N), local(3 * N);
	for (Device& d : devs)
	{
		CHECK_HIP(hipSetDevice(d.dev));
		CHECK_HIP(hipMemcpy(&reduced[3 * d.id], d.counts, 24, hipMemcpyDeviceToHost));
		CHECK_HIP(hipMemcpy(&local[3 * d.id], d.local, 24, hipMemcpyDeviceToHost));
		CHECK_NV(nv_status(d.ctx, d.stream));
	}
	uint64_t sumVisible = 0, sumCommands = 0;
	bool agree = true;
	for (int r = 0; r < N; ++r)
	{
		sumCommands += local[3 * r + 1];
		sumVisible += local[3 * r + 2];
		agree = agree && reduced[3 * r + 1] == reduced[1] && reduced[3 * r + 2] == reduced[2];
	}
	agree = agree && reduced[1] == sumCommands && reduced[2] == sumVisible && sumCommands == totalCmd;

	printf("{\"shard_driver\": \"rccl\", \"devices\": %d, \"shards\": %d, \"meshlets_total\": %llu, \"commands_total\": %llu, \"steps\": %d, \"ms_per_step\": %.5f, "
	       "\"meshlets_per_s\": %.4e, \"timing\": \"%s\", \"scaling\": \"%s\", \"visible_total\": %llu, \"allreduce\": \"%s\", "
	       "\"counts_agree_on_all_ranks\": %s}\n",
	       D, N, (unsigned long long)(totalCmd * 64), (unsigned long long)totalCmd, steps, seconds / steps * 1e3, (double)(totalCmd * 64) * steps / seconds,
	       oversubscribed ? "host-synchronised every phase (ranks sharing a device): not comparable with one rank per device" : "asynchronous: passes and collectives queued on the devices' streams",
	       totalMeshlets ? "strong" : "weak", (unsigned long long)reduced[2],
	       oversubscribed ? "one ncclAllReduce(ncclSum) of 3 x u64 per DEVICE and phase; the ranks sharing a device are summed on the host first (RCCL refuses a device twice in one communicator)"
	                      : "one ncclAllReduce(ncclSum) of 3 x u64 per phase",
	       agree ? "true" : "false");

	if (!dump.empty())
	{
		// PREFIX.scene: u32 magic 'NVSH', devices, totalDraws, cpd | draws[] | per device: u64 cmdBegin, cmdEnd, u32 drawBegin, drawCount, NvCullData,
		//               commands (padded) count u32 + records, meshlets count u64 + records
		// PREFIX.out:   per device: u64 local[3], reduced[3], u32 visible, then `visible` GLOBAL ids (command rebased by cmdBegin, lane << 24)
		FILE* fs = fopen((dump + ".scene").c_str(), "wb");
		FILE* fo = fopen((dump + ".out").c_str(), "wb");
		if (!fs || !fo)
		{
			perror(dump.c_str());
			return 1;
		}
		const uint32_t head[4] = { 0x4853564eu, (uint32_t)N, totalDraws, cpd };
		fwrite(head, 4, 4, fs);
		fwrite(draws.data(), sizeof(NvMeshDraw), draws.size(), fs);
		for (Device& d : devs)
		{
			CHECK_HIP(hipSetDevice(d.dev));
			const uint64_t range[2] = { d.cmdBegin, d.cmdEnd };
			const uint32_t dr[2] = { d.drawBegin, d.drawCount };
			fwrite(range, 8, 2, fs);
			fwrite(dr, 4, 2, fs);
			fwrite(&d.cull, sizeof(NvCullData), 1, fs);
			const uint32_t nc = (uint32_t)d.commandsHost.size();
			fwrite(&nc, 4, 1, fs);
			fwrite(d.commandsHost.data(), sizeof(NvMeshTaskCommand), nc, fs);
			const uint64_t nm = d.meshletsHost.size();
			fwrite(&nm, 8, 1, fs);
			fwrite(d.meshletsHost.data(), sizeof(NvMeshlet), nm, fs);

			uint32_t cc4[4];
			CHECK_HIP(hipMemcpy(cc4, d.ccb, 16, hipMemcpyDeviceToHost));
			const uint32_t nv = cc4[0] < NV_CLUSTER_LIMIT ? cc4[0] : NV_CLUSTER_LIMIT;
			std::vector<uint32_t> ids(nv);
			if (nv)
				CHECK_HIP(hipMemcpy(ids.data(), d.cib, (size_t)nv * 4, hipMemcpyDeviceToHost));
			// rank-local command ids -> ids of the pool: (local + cmdBegin) | lane << 24; fits 24 bits for pools <= 2^24 commands
			for (uint32_t& id : ids)
				id = ((id & 0xffffffu) + (uint32_t)d.cmdBegin) | (id & 0xff000000u);
			fwrite(&local[3 * d.id], 8, 3, fo);
			fwrite(&reduced[3 * d.id], 8, 3, fo);
			fwrite(&nv, 4, 1, fo);
			fwrite(ids.data(), 4, nv, fo);
		}
		fclose(fs);
		fclose(fo);
	}

	for (Device& d : devs)
	{
		CHECK_HIP(hipSetDevice(d.dev));
		nv_destroy(d.ctx);
		CHECK_HIP(hipStreamDestroy(d.stream));
	}
	for (ncclComm_t c : comms)
		ncclCommDestroy(c);
	return agree ? 0 : 6;
}
