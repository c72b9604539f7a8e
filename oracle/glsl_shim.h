/*
 * glsl_shim.h — the GLSL built-ins the reference's cull shaders use, as plain C++ (TEST INFRASTRUCTURE).
 *
 * oracle/ref_translate.py rewrites the reference's shader sources (read where they lie under
 * /root/reference/src/shaders) into C++ that includes this header; oracle/ref_runner.cpp then runs the
 * shader main() once per invocation, serially.  Together they form oracle/_ref/libniagara_ref.so, i.e. the
 * REFERENCE'S OWN SOURCE TEXT executing on the CPU.  Only what GLSL leaves to the implementation lives here:
 *   - vector/matrix operators: component-wise fp32, mat*vec = ((c0*x + c1*y) + c2*z) + c3*w, dot/length
 *     summed x->y->z, one IEEE operation per C operation (build with -ffp-contract=off);
 *   - log2/exp2/ceil/floor/sqrt = libm (so ceil(log2(x)) here is the NATURAL libm mapping, not the oracle's
 *     exact-exponent form; tests/test_oracle_vs_ref.py documents the measure-zero set where they differ);
 *   - the sampler: LINEAR + NEAREST-mip + CLAMP_TO_EDGE + MIN reduction (src/niagara.cpp:629,
 *     src/resources.cpp:294-325) = min over the non-zero-weight texels of the fp32 bilinear footprint;
 *   - atomics: serial read-modify-write.
 */
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace glsl
{

typedef unsigned int uint;

/* ---- fp16 storage type (GL_EXT_shader_16bit_storage) ---- */
struct float16_t
{
	uint16_t bits;
	operator float() const
	{
		uint32_t sign = (uint32_t)(bits & 0x8000u) << 16, e = (bits >> 10) & 31u, m = bits & 1023u, u;
		if (e == 0)
		{
			float v = ldexpf((float)m, -24);
			return sign ? -v : v;
		}
		u = e == 31 ? (sign | 0x7f800000u | (m << 13)) : (sign | ((e + 112u) << 23) | (m << 13));
		float f;
		memcpy(&f, &u, 4);
		return f;
	}
};

/* ---- vectors ---- */
struct uvec2
{
	uint x, y;
	uvec2() : x(0), y(0) {}
	uvec2(uint x_, uint y_) : x(x_), y(y_) {}
};
struct vec2
{
	float x, y;
	vec2() : x(0), y(0) {}
	explicit vec2(float s) : x(s), y(s) {}
	vec2(float x_, float y_) : x(x_), y(y_) {}
	explicit vec2(uvec2 v) : x((float)v.x), y((float)v.y) {}
};
struct ivec2
{
	int x, y;
	ivec2() : x(0), y(0) {}
	ivec2(int x_, int y_) : x(x_), y(y_) {}
	explicit ivec2(uvec2 v) : x((int)v.x), y((int)v.y) {}
};
struct uvec3
{
	uint x, y, z;
	uvec3() : x(0), y(0), z(0) {}
	uvec3(uint x_, uint y_, uint z_) : x(x_), y(y_), z(z_) {}
	uvec2 xy() const { return uvec2(x, y); }
};
struct bvec2
{
	bool x, y;
};

struct vec3
{
	float x, y, z;
	vec3() : x(0), y(0), z(0) {}
	explicit vec3(float s) : x(s), y(s), z(s) {}
	vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	vec3(vec2 v, float z_) : x(v.x), y(v.y), z(z_) {}
	vec2 xy() const { return vec2(x, y); }
};
struct vec4
{
	float x, y, z, w;
	vec4() : x(0), y(0), z(0), w(0) {}
	explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
	vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
	vec4(vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
	vec2 xy() const { return vec2(x, y); }
	vec2 zw() const { return vec2(z, w); }
	vec3 xyz() const { return vec3(x, y, z); }
	vec4 xwzy() const { return vec4(x, w, z, y); }
	explicit operator vec3() const { return vec3(x, y, z); }
};

inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator*(vec2 a, vec2 b) { return vec2(a.x * b.x, a.y * b.y); }
inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }

inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator*(vec4 a, vec4 b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator/(vec4 a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }

/* ---- matrices (column-major, like GLSL and glm) ---- */
struct mat4
{
	vec4 c[4];
	mat4() {}
	mat4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3, float c0, float c1, float c2, float c3,
	     float d0, float d1, float d2, float d3)
	{
		c[0] = vec4(a0, a1, a2, a3);
		c[1] = vec4(b0, b1, b2, b3);
		c[2] = vec4(c0, c1, c2, c3);
		c[3] = vec4(d0, d1, d2, d3);
	}
	vec4& operator[](int i) { return c[i]; }
	const vec4& operator[](int i) const { return c[i]; }
};
struct mat3
{
	vec3 c[3];
	mat3() {}
	explicit mat3(const mat4& m)
	{
		for (int i = 0; i < 3; ++i)
			c[i] = vec3(m.c[i].x, m.c[i].y, m.c[i].z);
	}
};
inline vec4 operator*(const mat4& m, vec4 v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }
inline vec3 operator*(const mat3& m, vec3 v) { return (m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z; }
inline mat4 transpose(const mat4& m)
{
	return mat4(m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x, m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y, m.c[0].z, m.c[1].z, m.c[2].z,
	            m.c[3].z, m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w);
}

/* ---- built-in functions ---- */
/* GLSL spec: cross(x,y) = (x[1]*y[2] - y[1]*x[2], x[2]*y[0] - y[2]*x[0], x[0]*y[1] - y[0]*x[1]) */
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
/* glm forms used by the host lines compiled verbatim (src/scene.cpp:207-220): component-wise, division is division */
inline float distance(vec3 a, vec3 b) { return length(vec3(a.x - b.x, a.y - b.y, a.z - b.z)); }
inline vec3& operator+=(vec3& a, vec3 b)
{
	a.x += b.x;
	a.y += b.y;
	a.z += b.z;
	return a;
}
inline vec3& operator/=(vec3& a, float s)
{
	a.x /= s;
	a.y /= s;
	a.z /= s;
	return a;
}
inline float sqrt(float x) { return sqrtf(x); }
inline float abs(float x) { return fabsf(x); }
inline float ceil(float x) { return ceilf(x); }
inline float log2(float x) { return log2f(x); }
inline float exp2(float x) { return exp2f(x); }
inline float fract(float x) { return x - floorf(x); }
inline vec2 fract(vec2 v) { return vec2(fract(v.x), fract(v.y)); }
/* GLSL spec: max(x,y) = x<y ? y : x ; min(x,y) = y<x ? y : x */
inline float max(float a, float b) { return a < b ? b : a; }
inline float min(float a, float b) { return b < a ? b : a; }
inline uint max(uint a, uint b) { return a < b ? b : a; }
inline uint min(uint a, uint b) { return b < a ? b : a; }
inline vec2 min(vec2 a, vec2 b) { return vec2(min(a.x, b.x), min(a.y, b.y)); }
inline vec2 max(vec2 a, vec2 b) { return vec2(max(a.x, b.x), max(a.y, b.y)); }
/* round(): GLSL leaves the direction of .5 to the implementation; defined as round-half-to-even (v_rndne_f32, SPIR-V RoundEven) */
inline float round(float x) { return rintf(x); }
/* shading attributes (normal / tangent unpacking, src/shaders/math.h:131-136) are outside the visibility path: the
 * translation keeps only math.h:2-49, and the mesh shader's call resolves to this stub */
inline void unpackTBN(uint, uint, vec3& normal, vec4& tangent)
{
	normal = vec3(0, 0, 1);
	tangent = vec4(1, 0, 0, 1);
}
inline bvec2 lessThanEqual(vec2 a, vec2 b)
{
	bvec2 r = { a.x <= b.x, a.y <= b.y };
	return r;
}
inline bool all(bvec2 b) { return b.x && b.y; }

/* ---- atomics (serial) ---- */
inline uint atomicAdd(uint& mem, uint v)
{
	uint old = mem;
	mem = old + v;
	return old;
}
inline int atomicAdd(int& mem, int v) /* shared int sharedCount (meshlet.task.glsl:50) */
{
	int old = mem;
	mem = old + v;
	return old;
}
inline uint atomicOr(uint& mem, uint v)
{
	uint old = mem;
	mem = old | v;
	return old;
}
inline uint atomicAnd(uint& mem, uint v)
{
	uint old = mem;
	mem = old & v;
	return old;
}

/* ---- images and the MIN-reduction sampler ---- */
struct texture2D
{
	const float* base;
	uint width, height, levels;
	const uint* mipOffset; /* in floats */
};
struct sampler
{
};
struct sampler2D
{
	texture2D t;
	sampler2D(texture2D t_, sampler) : t(t_) {}
};
struct image2D
{
	float* data;
	uint width, height;
};

inline void shim_axis(float t, uint size, int idx[2], bool use[2])
{
	float f0 = floorf(t), fr = t - f0, lim = (float)size;
	if (!(f0 >= -1.0f))
		f0 = -1.0f;
	if (f0 > lim)
		f0 = lim;
	int i0 = (int)f0, i1 = i0 + 1, hi = (int)size - 1;
	idx[0] = i0 < 0 ? 0 : (i0 > hi ? hi : i0);
	idx[1] = i1 < 0 ? 0 : (i1 > hi ? hi : i1);
	use[0] = (1.0f - fr) != 0.0f;
	use[1] = fr != 0.0f;
}

inline vec4 textureLod(sampler2D s, vec2 uv, float lod)
{
	int l = (int)lod;
	int top = (int)s.t.levels - 1;
	l = l < 0 ? 0 : (l > top ? top : l);
	uint w = s.t.width >> l, h = s.t.height >> l;
	w = w ? w : 1;
	h = h ? h : 1;
	const float* img = s.t.base + s.t.mipOffset[l];
	int xi[2], yi[2];
	bool xu[2], yu[2];
	shim_axis(uv.x * (float)w - 0.5f, w, xi, xu);
	shim_axis(uv.y * (float)h - 0.5f, h, yi, yu);
	float best = 0;
	bool have = false;
	for (int j = 0; j < 2; ++j)
		for (int i = 0; i < 2; ++i)
			if (xu[i] && yu[j])
			{
				float t = img[(size_t)yi[j] * w + xi[i]];
				best = have ? min(best, t) : t;
				have = true;
			}
	return vec4(best, 0, 0, 1);
}
inline vec4 texture(sampler2D s, vec2 uv) { return textureLod(s, uv, 0.0f); }
inline void imageStore(image2D img, ivec2 p, vec4 v) { img.data[(size_t)p.y * img.width + p.x] = v.x; }

/* invocation ids (gl_GlobalInvocationID etc.) are per-TU statics defined by oracle/ref_runner.cpp */

} // namespace glsl
