// submit.hip — indirect-argument fix-ups (single-workgroup kernels) for gfx950.
//
// Replaces src/shaders/tasksubmit.comp.glsl:27-47 and src/shaders/clustersubmit.comp.glsl:25-45.  They stay
// separate launches because they are separate dispatches of the reference's contract (src/niagara.cpp:1563-1568,
// 1603-1608) whose outputs — the {X,64,1} / {16,Y,16} grids and the padded dummy entries — a downstream consumer
// of the command / cluster lists reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/niagara_vis.h"

namespace nv
{

__global__ __launch_bounds__(64) void tasksubmit_kernel(uint32_t* __restrict__ count4, NvMeshTaskCommand* __restrict__ commands)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t raw = count4[0];
	const uint32_t count = raw < NV_TASK_WGLIMIT ? raw : NV_TASK_WGLIMIT;

	if (tid == 0)
	{
		uint32_t gx = (count + 63) / 64;
		count4[1] = gx < 65535u ? gx : 65535u;
		count4[2] = 64;
		count4[3] = 1;
	}

	const uint32_t boundary = (count + 63) & ~63u;
	if (count + tid < boundary)
		commands[count + tid] = NvMeshTaskCommand{ 0, 0, 0, 0, 0 };
}

__global__ __launch_bounds__(256) void clustersubmit_kernel(uint32_t* __restrict__ cc4, uint32_t* __restrict__ clusterIndices)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t raw = cc4[0];
	const uint32_t count = raw < NV_CLUSTER_LIMIT ? raw : NV_CLUSTER_LIMIT;

	if (tid == 0)
	{
		uint32_t gy = (count + 255) / 256;
		cc4[1] = NV_CLUSTER_TILE;
		cc4[2] = gy < 65535u ? gy : 65535u;
		cc4[3] = 256 / NV_CLUSTER_TILE;
	}

	const uint32_t boundary = (count + 255) & ~255u;
	if (count + tid < boundary)
		clusterIndices[count + tid] = ~0u;
}

// payload of the one all-reduce per phase in a sharded run
__global__ void pack_counts_kernel(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint64_t* out3)
{
	if (threadIdx.x == 0)
	{
		out3[0] = a ? (uint64_t)a[0] : 0;
		out3[1] = b ? (uint64_t)b[0] : 0;
		out3[2] = c ? (uint64_t)c[0] : 0;
	}
}

// the HIP analogue of vkCmdFillBuffer(buffer, 0, 4, 0) in front of each pass (src/niagara.cpp:1541,1586)
__global__ void reset_count_kernel(uint32_t* a, uint32_t* b)
{
	if (threadIdx.x == 0)
	{
		if (a)
			a[0] = 0;
		if (b)
			b[0] = 0;
	}
}

int launch_reset_count(hipStream_t stream, uint32_t* a, uint32_t* b)
{
	hipLaunchKernelGGL(reset_count_kernel, dim3(1), dim3(64), 0, stream, a, b);
	return (int)hipGetLastError();
}

int launch_tasksubmit(hipStream_t stream, uint32_t* count4, NvMeshTaskCommand* commands)
{
	hipLaunchKernelGGL(tasksubmit_kernel, dim3(1), dim3(64), 0, stream, count4, commands);
	return (int)hipGetLastError();
}

int launch_clustersubmit(hipStream_t stream, uint32_t* cc4, uint32_t* clusterIndices)
{
	hipLaunchKernelGGL(clustersubmit_kernel, dim3(1), dim3(256), 0, stream, cc4, clusterIndices);
	return (int)hipGetLastError();
}

int launch_pack_counts(hipStream_t stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint64_t* out3)
{
	hipLaunchKernelGGL(pack_counts_kernel, dim3(1), dim3(64), 0, stream, a, b, c, out3);
	return (int)hipGetLastError();
}

} // namespace nv
