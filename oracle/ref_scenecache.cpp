// oracle/ref_scenecache.cpp — TEST INFRASTRUCTURE: the reference's own scene-cache writer and loader
// (src/scenecache.cpp, compiled verbatim from where it lies under the reference tree; never copied) behind a C ABI, so
// that nv_scenecache_info / nv_scenecache_read are pinned against files the reference's code wrote
// and against what the reference's code loads.  glm / volk / meshoptimizer are un-vendored: oracle/ref_shims/ supplies
// layout-only stand-ins, and only the raw (uncompressed) branches are reachable.
#include REF_SCENECACHE_CPP

#include <stddef.h>

// src/scene.h:141 — defined in src/scene.cpp (opacity micromaps); reached only for meshes with ommIndexData != 0
void normalizeIndicesForOMM(uint32_t*, size_t) {}

namespace
{
Geometry g_geometry;
std::vector<Material> g_materials;
std::vector<MeshDraw> g_draws;
std::vector<Light> g_lights;
std::vector<std::string> g_texturePaths;
std::vector<Animation> g_animations;
std::vector<Keyframe> g_keyframes;
Camera g_camera;
vec3 g_sun;
} // namespace

extern "C" {

uint32_t ref_scene_sizeof(int what)
{
	switch (what)
	{
	case 0: return sizeof(SceneHeader);
	case 1: return offsetof(SceneHeader, hashMeta);
	case 2: return offsetof(SceneHeader, meshletMaxVertices);
	case 3: return offsetof(SceneHeader, clrtMode);
	case 4: return offsetof(SceneHeader, compressed);
	case 5: return offsetof(SceneHeader, compressedVertexBytes);
	case 6: return offsetof(SceneHeader, compressedMeshletVtx0Bytes);
	case 7: return offsetof(SceneHeader, vertexCount);
	case 8: return offsetof(SceneHeader, meshCount);
	case 9: return offsetof(SceneHeader, materialCount);
	case 10: return offsetof(SceneHeader, keyframeCount);
	case 11: return offsetof(SceneHeader, ommArrayDataSize);
	case 12: return offsetof(SceneHeader, ommStates);
	case 13: return offsetof(SceneHeader, camera);
	case 14: return offsetof(SceneHeader, sunDirection);
	case 15: return sizeof(Camera);
	case 16: return sizeof(Vertex);
	case 17: return sizeof(Material);
	case 18: return sizeof(Light);
	case 19: return sizeof(Animation);
	case 20: return sizeof(Keyframe);
	case 21: return sizeof(Meshlet);
	case 22: return sizeof(Mesh);
	case 23: return sizeof(MeshDraw);
	case 24: return kSceneCacheMagic;
	case 25: return kSceneCacheVersion;
	case 26: return offsetof(Camera, orientation);
	case 27: return offsetof(Camera, fovY);
	default: return 0xffffffffu;
	}
}

// saveSceneCache (src/scenecache.cpp:120-260), compressed = false; the side arrays are filled with a byte pattern
int ref_save_scene_cache(const char* path, const void* meshes, uint32_t meshCount, const void* meshlets, uint32_t meshletCount, const void* draws,
                         uint32_t drawCount, uint32_t vertexCount, uint32_t indexCount, uint32_t meshletdataCount, uint32_t meshletvtx0Count,
                         uint32_t materialCount, uint32_t lightCount, uint32_t animationCount, uint32_t keyframeCount, uint32_t texturePathCount,
                         const float* camera9, const float* sun3, uint64_t hashMeta, int clrtMode, uint32_t ommStates)
{
	Geometry geo;
	geo.vertices.resize(vertexCount);
	memset(geo.vertices.data(), 0x5a, vertexCount * sizeof(Vertex));
	geo.indices.assign(indexCount, 0x01020304u);
	geo.meshlets.resize(meshletCount);
	memcpy(geo.meshlets.data(), meshlets, meshletCount * sizeof(Meshlet));
	geo.meshletdata.assign(meshletdataCount, 0x0a0b0c0du);
	geo.meshletvtx0.assign(meshletvtx0Count, 0x1234);
	geo.meshes.resize(meshCount);
	memcpy(geo.meshes.data(), meshes, meshCount * sizeof(Mesh));
	geo.ommStates = ommStates;
	std::vector<Material> materials(materialCount);
	std::vector<MeshDraw> drawList(drawCount);
	memcpy(drawList.data(), draws, drawCount * sizeof(MeshDraw));
	std::vector<Light> lights(lightCount);
	std::vector<std::string> texturePaths(texturePathCount, std::string("textures/albedo.dds"));
	std::vector<Animation> animations(animationCount);
	std::vector<Keyframe> keyframes(keyframeCount);
	Camera camera;
	camera.position = { camera9[0], camera9[1], camera9[2] };
	camera.orientation = { camera9[3], camera9[4], camera9[5], camera9[6] };
	camera.fovY = camera9[7];
	camera.znear = camera9[8];
	vec3 sun = { sun3[0], sun3[1], sun3[2] };
	return saveSceneCache(path, geo, materials, drawList, lights, texturePaths, animations, keyframes, camera, sun, hashMeta, clrtMode != 0, false, false) ? 0 : 1;
}

// loadSceneCache (src/scenecache.cpp:273-370) into this module's globals; counts[0..2] = meshes, meshlets, draws
int ref_load_scene_cache(const char* path, uint64_t hashMeta, int clrtMode, int ommStates, uint32_t* counts3, float* camera9, float* sun3)
{
	if (!loadSceneCache(path, g_geometry, g_materials, g_draws, g_lights, g_texturePaths, g_animations, g_keyframes, g_camera, g_sun, hashMeta, clrtMode != 0,
	                    ommStates))
		return 1;
	counts3[0] = (uint32_t)g_geometry.meshes.size();
	counts3[1] = (uint32_t)g_geometry.meshlets.size();
	counts3[2] = (uint32_t)g_draws.size();
	const float cam[9] = { g_camera.position.x, g_camera.position.y, g_camera.position.z, g_camera.orientation.x, g_camera.orientation.y,
		                   g_camera.orientation.z, g_camera.orientation.w, g_camera.fovY, g_camera.znear };
	memcpy(camera9, cam, sizeof(cam));
	sun3[0] = g_sun.x, sun3[1] = g_sun.y, sun3[2] = g_sun.z;
	return 0;
}

void ref_loaded_arrays(void* meshes, void* meshlets, void* draws)
{
	memcpy(meshes, g_geometry.meshes.data(), g_geometry.meshes.size() * sizeof(Mesh));
	memcpy(meshlets, g_geometry.meshlets.data(), g_geometry.meshlets.size() * sizeof(Meshlet));
	memcpy(draws, g_draws.data(), g_draws.size() * sizeof(MeshDraw));
}

} // extern "C"
