// trianglecull.hip — the mesh stage's per-triangle cull for gfx950.
//
// Replaces src/shaders/meshlet.mesh.glsl:91-198 with MESH_CULL = 1 (src/config.h:10-11), minus the shading attributes:
// per slot of the grid clustersubmit wrote, the meshlet's vertices go to screen space exactly like the mesh shader
// computes them and each triangle gets its gl_CullPrimitiveEXT decision.
//
// Mapping to CDNA4: MESH_WGSIZE = 64 = one wavefront per meshlet, MESH_MAXVTX = 64 = one vertex per lane, MESH_MAXTRI =
// 96 = two triangle rounds per lane.  The reference's `shared vec3 vertexClip[]` + barrier() become a 768-byte LDS
// slice per wave and nothing else: the wave is the workgroup.  A workgroup of four waves walks four slots at a time,
// grid-stride; the slot's header chain (cluster index -> task command -> {draw, meshlet} -> {vertex refs, indices} ->
// vertices) is the kernel's cost — four dependent round trips for ~1.7 KB of payload — so the next slot's header is
// requested before the current slot's vertices are touched.  Bound: HBM (gather-heavy); algorithmic bytes per slot =
// 4 + 20 + 48 + 24 + refs (2 or 4 B x vertexCount) + 3 B x triangleCount + 8 B x vertexCount (position half of the 16-B
// vertex) + 16 B out.
#include "cullmath.cuh"
#include "args.cuh"

namespace nv
{

constexpr int TC_WAVES = 4;
constexpr int TC_THREADS = TC_WAVES * 64;

// mat4 * vec4 in the reference's association: ((c0*x + c1*y) + c2*z) + c3*w
NV_DEV void mat4_mul(const float* m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
	for (int r = 0; r < 4; ++r)
		out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

__global__ __launch_bounds__(TC_THREADS) void trianglecull_kernel(TriangleArgs a)
{
	__shared__ float s_clip[TC_WAVES][64][3];

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// grid of the consumer: {cc4[1], cc4[2], cc4[3]} = {16, Y, 16}; index = x + 256 y + 16 z enumerates [0, 256 Y)
	const uint32_t slots = a.cc4[1] * a.cc4[2] * a.cc4[3];
	const uint32_t stride = gridDim.x * TC_WAVES;
	const uint16_t* data16 = reinterpret_cast<const uint16_t*>(a.meshletData);
	const uint8_t* data8 = reinterpret_cast<const uint8_t*>(a.meshletData);
	float (*clipv)[3] = s_clip[wave];

	unsigned long long clusters = 0, triangles = 0, keptTotal = 0;

	for (uint32_t index = blockIdx.x * TC_WAVES + wave; index < slots; index += stride)
	{
		const uint32_t ci = a.clusterIndices[index];
		NvTriangleMask out = { { 0, 0, 0 }, 0 };
		if (ci != ~0u) // wave-uniform
		{
			const NvMeshTaskCommand command = a.commands[ci & 0xffffffu];
			const uint32_t mi = command.taskOffset + (ci >> 24);
			const NvMeshlet m = a.meshlets[mi];
			const NvMeshDraw d = a.draws[command.drawId];
			const uint32_t vertexCount = m.vertexCount, triangleCount = m.triangleCount;
			const bool shortRefs = m.shortRefs == 1;
			const uint32_t vertexOffset = m.dataOffset;
			const uint32_t indexOffset = m.dataOffset + (shortRefs ? (vertexCount + 1) / 2 : vertexCount);

			// ---- vertex phase (meshlet.mesh.glsl:121-160): lane = vertex
			if (lane < vertexCount)
			{
				const uint32_t vi = (shortRefs ? (uint32_t)data16[vertexOffset * 2 + lane] : a.meshletData[vertexOffset + lane]) + m.baseVertex;
				const uint2 pv = *reinterpret_cast<const uint2*>(a.vertices + vi); // vx vy | vz tp
				const f3 position = { half_bits_to_float(pv.x & 0xffffu), half_bits_to_float(pv.x >> 16), half_bits_to_float(pv.y & 0xffffu) };
				const f3 q = { d.orientation[0], d.orientation[1], d.orientation[2] };
				const f3 rot = rotate_quat(position, q, d.orientation[3]);
				const float wx = rot.x * d.scale + d.position[0];
				const float wy = rot.y * d.scale + d.position[1];
				const float wz = rot.z * d.scale + d.position[2];
				float v4[4], clip[4];
				mat4_mul(a.globals.cullData.view, wx, wy, wz, 1.0f, v4);
				mat4_mul(a.globals.projection, v4[0], v4[1], v4[2], v4[3], clip);
				// vertexClip[i] = vec3((clip.xy / clip.w * 0.5 + vec2(0.5)) * screen, clip.w)
				clipv[lane][0] = ((clip[0] / clip[3]) * 0.5f + 0.5f) * a.globals.screenWidth;
				clipv[lane][1] = ((clip[1] / clip[3]) * 0.5f + 0.5f) * a.globals.screenHeight;
				clipv[lane][2] = clip[3];
			}
			// the wave is the workgroup: LDS writes of a wave are visible to its own later reads in program order
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

			// ---- triangle phase (:166-205): lanes i and i + 64
			uint32_t kept = 0;
#pragma unroll
			for (uint32_t round = 0; round < 2; ++round)
			{
				const uint32_t i = round * 64u + lane;
				bool keep = false;
				if (i < triangleCount)
				{
					const uint32_t offset = indexOffset * 4 + i * 3;
					const uint32_t ia = data8[offset], ib = data8[offset + 1], ic = data8[offset + 2];
					const float *pa = clipv[ia & 63u], *pb = clipv[ib & 63u], *pc = clipv[ic & 63u];
					bool culled = false;
					const float ebx = pb[0] - pa[0], eby = pb[1] - pa[1];
					const float ecx = pc[0] - pa[0], ecy = pc[1] - pa[1];
					culled = culled || (ebx * ecy <= eby * ecx); // backface + zero-area
					const float bminx = gl_min(pa[0], gl_min(pb[0], pc[0])), bminy = gl_min(pa[1], gl_min(pb[1], pc[1]));
					const float bmaxx = gl_max(pa[0], gl_max(pb[0], pc[0])), bmaxy = gl_max(pa[1], gl_max(pb[1], pc[1]));
					const float sbprec = 1.0f / 256.0f;
					// round(): half-to-even (v_rndne_f32), the definition the oracle and the shim share
					culled = culled || (__builtin_rintf(bminx - sbprec) == __builtin_rintf(bmaxx) || __builtin_rintf(bminy) == __builtin_rintf(bmaxy + sbprec));
					culled = culled && (pa[2] > 0 && pb[2] > 0 && pc[2] > 0);
					keep = !culled;
				}
				const uint64_t ballot = __ballot(keep);
				if (round == 0)
				{
					out.keep[0] = (uint32_t)ballot;
					out.keep[1] = (uint32_t)(ballot >> 32);
				}
				else
					out.keep[2] = (uint32_t)ballot;
				kept += (uint32_t)__builtin_popcountll(ballot);
			}
			out.counts = (triangleCount & 0xffu) | (vertexCount & 0xffu) << 8 | kept << 16;
			clusters += 1;
			triangles += triangleCount;
			keptTotal += kept;
			__builtin_amdgcn_wave_barrier(); // the next slot overwrites this wave's LDS slice
		}
		if (lane == 0 && index < a.capacity)
			*reinterpret_cast<uint4*>(a.masks + index) = make_uint4(out.keep[0], out.keep[1], out.keep[2], out.counts);
	}

	// totals: wave-uniform counters, one atomic per wave and total
	if (lane == 0 && clusters)
	{
		atomicAdd(&a.totals[0], clusters);
		atomicAdd(&a.totals[1], triangles);
		atomicAdd(&a.totals[2], keptTotal);
	}
}

int launch_trianglecull(hipStream_t stream, const TriangleArgs& a, uint32_t gridBlocks)
{
	hipLaunchKernelGGL(trianglecull_kernel, dim3(gridBlocks), dim3(TC_THREADS), 0, stream, a);
	return (int)hipGetLastError();
}

} // namespace nv
