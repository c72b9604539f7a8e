// oracle/ref_shims/glm/_pod.hpp — TEST INFRASTRUCTURE.  glm is an un-vendored submodule of the reference
// (.gitmodules:10-12, no pinned commit); the reference's scene-cache code (src/scenecache.cpp, src/scene.h) uses its
// vector types only as plain data inside structs that are written to disk raw.  These stand-ins give those structs the
// layout glm gives them under the reference's build flags (CMakeLists.txt:18: GLM_FORCE_XYZW_ONLY,
// GLM_FORCE_QUAT_DATA_XYZW): tightly packed floats, quaternion stored x, y, z, w.  Nothing here computes.
#pragma once
namespace glm
{
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct quat { float x, y, z, w; };
struct mat2 { vec2 c[2]; };
struct mat3 { vec3 c[3]; };
struct mat4 { vec4 c[4]; };
} // namespace glm
