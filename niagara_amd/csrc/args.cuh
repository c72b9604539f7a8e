// args.cuh — kernel argument blocks shared by the launchers (context.hip) and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/niagara_vis.h"
#include "ordered.cuh"

namespace nv
{

struct ClusterArgs
{
	NvCullData cd;
	NvPyramidDesc pyr;
	const NvMeshTaskCommand* __restrict__ commands;
	const uint32_t* __restrict__ count4; // dccb: {commandCount, groupCountX, 64, 1}
	const NvMeshDraw* __restrict__ draws;
	const NvMeshlet* __restrict__ meshlets; // AoS (used when soaBounds == nullptr)
	const uint2* __restrict__ soaBounds;
	const uint32_t* __restrict__ soaCones;
	uint32_t* __restrict__ mvb;
	uint32_t* __restrict__ clusterIndices;
	uint32_t* __restrict__ clusterCount4;
	uint32_t* __restrict__ payloadCounts; // taskcull only
	uint64_t* __restrict__ state;
	OrderCtl* __restrict__ ctl;
	uint32_t stateCapacity;
	uint32_t commandCountOverride; // probe/taskcull: explicit command count (0 = use count4[1]*64)
	float* __restrict__ probeOut;
	uint32_t debugMode; // tuning experiments only (NV_DEBUG_MODE); 0 in production
};

struct DrawArgs
{
	NvCullData cd;
	NvPyramidDesc pyr;
	const NvMeshDraw* draws;
	const NvMesh* meshes;
	void* commands;
	uint32_t* count4;
	uint32_t* dvb;
	uint64_t* state;
	OrderCtl* ctl;
	uint32_t stateCapacity;
};

} // namespace nv
