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

// meshlet.mesh.glsl:91-116: decode of one cluster index per thread (the header part of the mesh stage)
__global__ __launch_bounds__(256) void cluster_expand_kernel(const NvMeshTaskCommand* __restrict__ commands, const NvMeshlet* __restrict__ meshlets,
                                                            const uint32_t* __restrict__ clusterIndices, const uint32_t* __restrict__ cc4,
                                                            NvClusterRecord* __restrict__ records, uint32_t capacity, unsigned long long* __restrict__ partials)
{
	// grid of the consumer: {cc4[1], cc4[2], cc4[3]} = {16, Y, 16}; index = x + 256 y + 16 z enumerates [0, 256 Y)
	const uint32_t slots = cc4[1] * cc4[2] * cc4[3];
	unsigned long long clusters = 0, vertices = 0, triangles = 0;
	for (uint32_t index = blockIdx.x * 256u + threadIdx.x; index < slots; index += gridDim.x * 256u)
	{
		const uint32_t ci = clusterIndices[index];
		NvClusterRecord r = { ~0u, 0, 0, 0, 0, 0, 0, 0 };
		if (ci != ~0u)
		{
			const NvMeshTaskCommand command = commands[ci & 0xffffffu];
			const uint32_t mi = command.taskOffset + (ci >> 24);
			const NvMeshlet m = meshlets[mi];
			r.drawId = command.drawId;
			r.meshletIndex = mi;
			r.vertexCount = m.vertexCount;
			r.triangleCount = m.triangleCount;
			r.vertexOffset = m.dataOffset;
			r.shortRefs = m.shortRefs == 1 ? 1u : 0u;
			r.indexOffset = m.dataOffset + (r.shortRefs ? (r.vertexCount + 1) / 2 : r.vertexCount);
			r.baseVertex = m.baseVertex;
			clusters += 1;
			vertices += r.vertexCount;
			triangles += r.triangleCount;
		}
		if (index < capacity)
			records[index] = r;
	}
	// totals: wave reduction, then per-workgroup partial sums with plain stores; totals3_kernel adds them up.  (One atomic
	// per wave into the caller's three adjacent counters — one cache line — serialises in its L2 channel.)
	for (int o = 32; o > 0; o >>= 1)
	{
		clusters += __shfl_xor(clusters, o, 64);
		vertices += __shfl_xor(vertices, o, 64);
		triangles += __shfl_xor(triangles, o, 64);
	}
	__shared__ unsigned long long s_tot[4][3];
	if ((threadIdx.x & 63u) == 0)
	{
		s_tot[threadIdx.x >> 6][0] = clusters;
		s_tot[threadIdx.x >> 6][1] = vertices;
		s_tot[threadIdx.x >> 6][2] = triangles;
	}
	__syncthreads();
	if (threadIdx.x < 3)
		partials[(size_t)blockIdx.x * 3 + threadIdx.x] = s_tot[0][threadIdx.x] + s_tot[1][threadIdx.x] + s_tot[2][threadIdx.x] + s_tot[3][threadIdx.x];
}

// adds the per-workgroup partial sums of a launch to the caller's three totals (one workgroup)
__global__ __launch_bounds__(256) void totals3_kernel(const unsigned long long* __restrict__ partials, uint32_t blocks, unsigned long long* __restrict__ totals)
{
	__shared__ unsigned long long s_part[4][3];
	unsigned long long t[3] = { 0, 0, 0 };
	for (uint32_t i = threadIdx.x; i < blocks; i += 256)
	{
		t[0] += partials[(size_t)i * 3];
		t[1] += partials[(size_t)i * 3 + 1];
		t[2] += partials[(size_t)i * 3 + 2];
	}
#pragma unroll
	for (int k = 0; k < 3; ++k)
		for (int o = 32; o > 0; o >>= 1)
			t[k] += __shfl_xor(t[k], o, 64);
	if ((threadIdx.x & 63u) == 0)
		for (int k = 0; k < 3; ++k)
			s_part[threadIdx.x >> 6][k] = t[k];
	__syncthreads();
	if (threadIdx.x < 3)
		totals[threadIdx.x] += s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
}

int launch_totals3(hipStream_t stream, const unsigned long long* partials, uint32_t blocks, unsigned long long* totals)
{
	hipLaunchKernelGGL(totals3_kernel, dim3(1), dim3(256), 0, stream, partials, blocks, totals);
	return (int)hipGetLastError();
}

int launch_cluster_expand(hipStream_t stream, const NvMeshTaskCommand* commands, const NvMeshlet* meshlets, const uint32_t* clusterIndices,
                          const uint32_t* cc4, NvClusterRecord* records, uint32_t capacity, uint64_t* totals, unsigned long long* partials,
                          uint32_t gridBlocks)
{
	hipLaunchKernelGGL(cluster_expand_kernel, dim3(gridBlocks), dim3(256), 0, stream, commands, meshlets, clusterIndices, cc4, records, capacity, partials);
	return launch_totals3(stream, partials, gridBlocks, reinterpret_cast<unsigned long long*>(totals));
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
