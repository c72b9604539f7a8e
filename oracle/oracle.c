/*
 * oracle.c — CPU restatement of niagara's visibility passes.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h for who may use it and for the parity-pinning status).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math [-fopenmp]  (oracle/Makefile)
 *
 * Every function cites the reference lines it follows (paths relative to the reference tree).
 * Arithmetic is written one IEEE fp32 operation per C operation, in the order the GLSL spells it.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TASK_WGSIZE 64u           /* src/config.h:2 */
#define TASK_WGLIMIT (1u << 22)   /* src/config.h:25 */
#define CLUSTER_LIMIT (1u << 24)  /* src/config.h:28 */
#define CLUSTER_TILE 16u          /* src/config.h:22 */

/* ------------------------------------------------------------------ bit helpers */

static inline uint32_t f2u(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	return u;
}
static inline float u2f(uint32_t u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}

/* IEEE binary16 -> binary32, exact (what float(float16_t) does in clustercull.comp.glsl:72-76) */
float orc_half_to_float(uint16_t h)
{
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1fu;
	uint32_t man = h & 0x3ffu;
	if (exp == 0)
	{
		if (man == 0)
			return u2f(sign);
		/* subnormal: man * 2^-24, exact in fp32 */
		float v = (float)man * u2f(0x33800000u); /* 2^-24 */
		return sign ? -v : v;
	}
	if (exp == 31)
		return u2f(sign | 0x7f800000u | (man << 13));
	return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

/* GLSL min/max: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x */
static inline float gl_minf(float x, float y) { return y < x ? y : x; }
static inline float gl_maxf(float x, float y) { return x < y ? y : x; }
static inline uint32_t minu(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* ------------------------------------------------------------------ math.h */

/* cross(a,b) per the GLSL spec: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y) */
static inline void cross3(const float a[3], const float b[3], float o[3])
{
	o[0] = a[1] * b[2] - b[1] * a[2];
	o[1] = a[2] * b[0] - b[2] * a[0];
	o[2] = a[0] * b[1] - b[0] * a[1];
}

static inline float dot3(const float a[3], const float b[3])
{
	return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static inline float length3(const float a[3])
{
	return sqrtf(dot3(a, a));
}

/* src/shaders/math.h:46-49  v + 2.0 * cross(q.xyz, cross(q.xyz, v) + q.w * v) */
void orc_rotate_quat(const float v[3], const float q[4], float out[3])
{
	float t[3], u[3];
	cross3(q, v, t);
	t[0] = t[0] + q[3] * v[0];
	t[1] = t[1] + q[3] * v[1];
	t[2] = t[2] + q[3] * v[2];
	cross3(q, t, u);
	out[0] = v[0] + 2.0f * u[0];
	out[1] = v[1] + 2.0f * u[1];
	out[2] = v[2] + 2.0f * u[2];
}

/* (view * vec4(p, 1)).xyz with mat*vec = ((col0*x + col1*y) + col2*z) + col3*w, w = 1 */
static inline void view_point(const float m[16], const float p[3], float o[3])
{
	for (int r = 0; r < 3; ++r)
		o[r] = ((m[r] * p[0] + m[4 + r] * p[1]) + m[8 + r] * p[2]) + m[12 + r];
}

/* mat3(view) * v = (col0*x + col1*y) + col2*z */
static inline void view_dir(const float m[16], const float v[3], float o[3])
{
	for (int r = 0; r < 3; ++r)
		o[r] = (m[r] * v[0] + m[4 + r] * v[1]) + m[8 + r] * v[2];
}

/* src/shaders/math.h:2-22 */
int orc_project_sphere(const float c[3], float r, float znear, float P00, float P11, float aabb[4])
{
	if (c[2] < r + znear)
		return 0;

	float crx = c[0] * r, cry = c[1] * r, crz = c[2] * r;
	float czr2 = c[2] * c[2] - r * r;

	float vx = sqrtf(c[0] * c[0] + czr2);
	float minx = (vx * c[0] - crz) / (vx * c[2] + crx);
	float maxx = (vx * c[0] + crz) / (vx * c[2] - crx);

	float vy = sqrtf(c[1] * c[1] + czr2);
	float miny = (vy * c[1] - crz) / (vy * c[2] + cry);
	float maxy = (vy * c[1] + crz) / (vy * c[2] - cry);

	/* aabb = vec4(minx*P00, miny*P11, maxx*P00, maxy*P11); aabb = aabb.xwzy * vec4(.5,-.5,.5,-.5) + vec4(.5) */
	float ax = minx * P00, ay = miny * P11, az = maxx * P00, aw = maxy * P11;
	aabb[0] = ax * 0.5f + 0.5f;
	aabb[1] = aw * -0.5f + 0.5f;
	aabb[2] = az * 0.5f + 0.5f;
	aabb[3] = ay * -0.5f + 0.5f;
	return 1;
}

/* exact ceil(log2(x)) for finite x > 0 (also correct for subnormals); +inf -> 129 */
static inline int ceil_log2_exact(float x)
{
	uint32_t u = f2u(x);
	uint32_t e = (u >> 23) & 0xffu;
	uint32_t m = u & 0x7fffffu;
	if (e == 0)
	{
		/* subnormal: x = m * 2^-149 */
		int hb = 31 - __builtin_clz(m);
		int pow2 = (m & (m - 1)) == 0;
		return (hb - 149) + (pow2 ? 0 : 1);
	}
	if (e == 255)
		return 129;
	return (int)e - 127 + (m != 0);
}

static inline float fractf(float x) { return x - floorf(x); }

/* src/shaders/math.h:24-39.  Returns an integer-valued level in [0, 32]; raw levels above 32 are
 * reported as 32 (every consumer clamps to levels-1 <= 15, sampler maxLod = 16). */
float orc_occlusion_mip(const float aabb[4], float pw, float ph)
{
	float sx = aabb[2] - aabb[0];
	float sy = aabb[3] - aabb[1];
	float m = gl_maxf(sx * pw, sy * ph);

	if (!(m > 0.0f))
		return 0.0f; /* log2 of 0/negative/NaN: max(level, 0) = 0 */

	int level = ceil_log2_exact(m);
	if (level <= 0)
		return 0.0f;
	if (level > 32)
		level = 32;

	/* fmipSize = pyramidSize * exp2(1 - level): exact power-of-two scale, 2^(1-level) in [2^-31, 1] */
	float scale = u2f((uint32_t)(127 + 1 - level) << 23);
	float fx = pw * scale, fy = ph * scale;
	int fits = (fractf(aabb[0] * fx) + sx * fx <= 2.0f) && (fractf(aabb[1] * fy) + sy * fy <= 2.0f);
	level -= fits;

	return (float)level;
}

/* src/shaders/math.h:41-44 with camera_position = 0 (clustercull.comp.glsl:102) */
int orc_cone_cull(const float c[3], float r, const float axis[3], float cutoff)
{
	return dot3(c, axis) >= cutoff * length3(c) + r;
}

/* ------------------------------------------------------------------ sampler */

/* One axis of the bilinear footprint: texel indices (clamped to edge) and which of them carry weight. */
static inline void footprint(float t, uint32_t size, int idx[2], int use[2])
{
	float f0 = floorf(t);
	float fr = t - f0;
	float lim = (float)size;
	if (!(f0 >= -1.0f))
		f0 = -1.0f; /* also catches NaN */
	if (f0 > lim)
		f0 = lim;
	int i0 = (int)f0, i1 = i0 + 1;
	int hi = (int)size - 1;
	idx[0] = i0 < 0 ? 0 : (i0 > hi ? hi : i0);
	idx[1] = i1 < 0 ? 0 : (i1 > hi ? hi : i1);
	use[0] = (1.0f - fr) != 0.0f;
	use[1] = fr != 0.0f;
}

/* texture()/textureLod() on one mip with VK_SAMPLER_REDUCTION_MODE_MIN, LINEAR filter, CLAMP_TO_EDGE
 * (src/niagara.cpp:629, src/resources.cpp:294-325) */
float orc_sample_min_image(const float* img, uint32_t w, uint32_t h, float u, float v)
{
	int xi[2], xu[2], yi[2], yu[2];
	footprint(u * (float)w - 0.5f, w, xi, xu);
	footprint(v * (float)h - 0.5f, h, yi, yu);

	float best = 0.0f;
	int have = 0;
	for (int j = 0; j < 2; ++j)
		for (int i = 0; i < 2; ++i)
			if (xu[i] && yu[j])
			{
				float t = img[(size_t)yi[j] * w + (size_t)xi[i]];
				best = have ? gl_minf(best, t) : t;
				have = 1;
			}
	return best;
}

static inline uint32_t mip_dim(uint32_t d, uint32_t level)
{
	uint32_t r = d >> level;
	return r ? r : 1u;
}

float orc_sample_min(const OrcPyramid* p, float u, float v, float level)
{
	int l = (int)level; /* integer-valued, nearest-mip */
	if (l < 0)
		l = 0;
	if (l > (int)p->levels - 1)
		l = (int)p->levels - 1;
	return orc_sample_min_image(p->base + p->mipOffset[l], mip_dim(p->width, (uint32_t)l), mip_dim(p->height, (uint32_t)l), u, v);
}

/* ------------------------------------------------------------------ host helpers */

/* src/niagara.cpp:439-447 */
uint32_t orc_previous_pow2(uint32_t v)
{
	uint32_t r = 1;
	while (r * 2 < v)
		r *= 2;
	return r;
}

/* src/resources.cpp:280-292 */
uint32_t orc_image_mip_levels(uint32_t w, uint32_t h)
{
	uint32_t n = 1;
	while (w > 1 || h > 1)
	{
		n++;
		w /= 2;
		h /= 2;
	}
	return n;
}

/* src/niagara.cpp:1340-1344 */
void orc_pyramid_init(OrcPyramid* p, uint32_t depthW, uint32_t depthH)
{
	p->width = orc_previous_pow2(depthW);
	p->height = orc_previous_pow2(depthH);
	p->levels = orc_image_mip_levels(p->width, p->height);
	uint32_t off = 0;
	for (uint32_t i = 0; i < 16; ++i)
	{
		p->mipOffset[i] = off;
		if (i < p->levels)
			off += mip_dim(p->width, i) * mip_dim(p->height, i);
	}
	p->totalTexels = off;
}

/* src/niagara.cpp:424-437 (projection, normalizePlane) and :1487-1516 (CullData).
 * glm is not vendored in the reference snapshot; the matrix steps are defined here as:
 * mat4_cast(q) per the usual quaternion->matrix formula, inverse of the rigid [R|t] as [R^T | -(R^T t)],
 * then scale(1,1,-1) * view negates row 2. */
void orc_build_cull_data(OrcCullData* out, const float camPos[3], const float q[4], float fovY, float znear,
                         float drawDistance, uint32_t vw, uint32_t vh, uint32_t pw, uint32_t ph, uint32_t drawCount,
                         int debugLodStep)
{
	memset(out, 0, sizeof(*out));

	float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
	float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
	float qxz = qx * qz, qxy = qx * qy, qyz = qy * qz;
	float qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;

	/* R[col][row] */
	float R[3][3];
	R[0][0] = 1.0f - 2.0f * (qyy + qzz);
	R[0][1] = 2.0f * (qxy + qwz);
	R[0][2] = 2.0f * (qxz - qwy);
	R[1][0] = 2.0f * (qxy - qwz);
	R[1][1] = 1.0f - 2.0f * (qxx + qzz);
	R[1][2] = 2.0f * (qyz + qwx);
	R[2][0] = 2.0f * (qxz + qwy);
	R[2][1] = 2.0f * (qyz - qwx);
	R[2][2] = 1.0f - 2.0f * (qxx + qyy);

	/* inverse: rotation = R^T, translation = -(R^T * t) */
	float V[16];
	memset(V, 0, sizeof(V));
	for (int c = 0; c < 3; ++c)
		for (int r = 0; r < 3; ++r)
			V[4 * c + r] = R[r][c];
	for (int r = 0; r < 3; ++r)
	{
		float d = (R[r][0] * camPos[0] + R[r][1] * camPos[1]) + R[r][2] * camPos[2];
		V[12 + r] = -d;
	}
	V[15] = 1.0f;
	/* scale(1,1,-1) * view: negate row 2 */
	for (int c = 0; c < 4; ++c)
		V[4 * c + 2] = -V[4 * c + 2];
	memcpy(out->view, V, sizeof(V));

	float f = 1.0f / tanf(fovY / 2.0f);
	float aspect = (float)vw / (float)vh;
	out->P00 = f / aspect;
	out->P11 = f;
	out->znear = znear;
	out->zfar = drawDistance;

	/* frustumX = normalizePlane(projT[3] + projT[0]) = (P00, 0, 1, 0) / |(P00,0,1)| */
	float lx = sqrtf((out->P00 * out->P00 + 0.0f * 0.0f) + 1.0f * 1.0f);
	float ly = sqrtf((0.0f * 0.0f + out->P11 * out->P11) + 1.0f * 1.0f);
	out->frustum[0] = out->P00 / lx;
	out->frustum[1] = 1.0f / lx;
	out->frustum[2] = out->P11 / ly;
	out->frustum[3] = 1.0f / ly;

	out->lodTarget = (2.0f / out->P11) * (1.0f / (float)vh) * (float)(1 << debugLodStep);
	out->pyramidWidth = (float)pw;
	out->pyramidHeight = (float)ph;
	out->drawCount = drawCount;
}

/* src/niagara.cpp:1002-1020 */
void orc_assign_visibility_offsets(OrcMeshDraw* draws, uint32_t n, const OrcMesh* meshes, uint32_t* slots, uint32_t* postMask)
{
	uint32_t count = 0, mask = 0;
	for (uint32_t i = 0; i < n; ++i)
	{
		const OrcMesh* mesh = &meshes[draws[i].meshIndex];
		draws[i].meshletVisibilityOffset = count;
		uint32_t mc = 0;
		for (uint32_t l = 0; l < mesh->lodCount; ++l)
			mc = mc > mesh->lods[l].meshletCount ? mc : mesh->lods[l].meshletCount;
		count += mc;
		mask |= 1u << draws[i].postPass;
	}
	*slots = count;
	*postMask = mask;
}

/* src/niagara.cpp:460-469 (pcg32_random_r) */
uint32_t orc_pcg32(uint64_t* state, uint64_t inc)
{
	uint64_t old = *state;
	*state = old * 6364136223846793005ULL + (inc | 1);
	uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
}

/* src/niagara.cpp:969-998.  The three rand01() calls inside vec3(...) are evaluated x, y, z. */
void orc_synth_draws(OrcMeshDraw* draws, uint32_t n, uint32_t meshCount, float sceneRadius)
{
	uint64_t state = 0x42;
	const uint64_t inc = 0xda3e39cb94b95bdbULL;
#define RAND01() ((double)orc_pcg32(&state, inc) / (double)(1ull << 32))
	for (uint32_t i = 0; i < n; ++i)
	{
		OrcMeshDraw* d = &draws[i];
		memset(d, 0, sizeof(*d));
		uint32_t meshIndex = orc_pcg32(&state, inc) % meshCount;
		d->position[0] = (float)RAND01() * sceneRadius * 2 - sceneRadius;
		d->position[1] = (float)RAND01() * sceneRadius * 2 - sceneRadius;
		d->position[2] = (float)RAND01() * sceneRadius * 2 - sceneRadius;
		d->scale = (float)RAND01() + 1;
		d->scale *= 2;
		float ax = (float)RAND01() * 2 - 1;
		float ay = (float)RAND01() * 2 - 1;
		float az = (float)RAND01() * 2 - 1;
		float inv = 1.0f / sqrtf((ax * ax + ay * ay) + az * az); /* glm::normalize = v * inversesqrt(dot(v,v)) */
		ax *= inv;
		ay *= inv;
		az *= inv;
		float angle = ((float)RAND01() * 90.f) * 0.01745329251994329576923690768489f; /* glm::radians */
		float s = sinf(angle * 0.5f);
		d->orientation[0] = ax * s;
		d->orientation[1] = ay * s;
		d->orientation[2] = az * s;
		d->orientation[3] = cosf(angle * 0.5f);
		d->meshIndex = meshIndex;
	}
#undef RAND01
}

/* ------------------------------------------------------------------ shared per-invocation math */

/* frustum + near/far (drawcull.comp.glsl:77-82, clustercull.comp.glsl:103-108) */
static inline int frustum_test(const OrcCullData* cd, const float c[3], float r)
{
	int vis = 1;
	vis = vis && c[2] * cd->frustum[1] - fabsf(c[0]) * cd->frustum[0] > -r;
	vis = vis && c[2] * cd->frustum[3] - fabsf(c[1]) * cd->frustum[2] > -r;
	vis = vis && c[2] + r > cd->znear && c[2] - r < cd->zfar;
	return vis;
}

/* HiZ test (drawcull.comp.glsl:86-99, clustercull.comp.glsl:110-123); returns updated visibility */
static inline int hiz_test(const OrcCullData* cd, const OrcPyramid* pyr, const float c[3], float r)
{
	float aabb[4];
	if (orc_project_sphere(c, r, cd->znear, cd->P00, cd->P11, aabb))
	{
		float level = orc_occlusion_mip(aabb, cd->pyramidWidth, cd->pyramidHeight);
		float depth = orc_sample_min(pyr, (aabb[0] + aabb[2]) * 0.5f, (aabb[1] + aabb[3]) * 0.5f, level);
		float depthSphere = cd->znear / (c[2] - r);
		return depthSphere > depth;
	}
	return 1;
}

/* ------------------------------------------------------------------ drawcull */

typedef struct
{
	uint8_t emit;
	uint8_t lod;
	uint8_t oldVis;
} DrawDecision;

/* drawcull.comp.glsl:54-118 up to the LOD choice, plus the LATE visibility write (:154-155) */
static inline DrawDecision drawcull_decide(const OrcCullData* cd, int late, uint32_t di, const OrcMeshDraw* draws,
                                           const OrcMesh* meshes, uint32_t* dvb, const OrcPyramid* pyr)
{
	DrawDecision dec = { 0, 0, 0 };
	const OrcMeshDraw* d = &draws[di];

	if (d->postPass != cd->postPass)
		return dec;
	if (!late && dvb[di] == 0)
		return dec;

	const OrcMesh* mesh = &meshes[d->meshIndex];

	float rc[3], wc[3], c[3];
	orc_rotate_quat(mesh->center, d->orientation, rc);
	wc[0] = rc[0] * d->scale + d->position[0];
	wc[1] = rc[1] * d->scale + d->position[1];
	wc[2] = rc[2] * d->scale + d->position[2];
	view_point(cd->view, wc, c);
	float radius = mesh->radius * d->scale;

	int visible = frustum_test(cd, c, radius);
	visible = visible || cd->cullingEnabled == 0;

	if (late && visible && cd->occlusionEnabled == 1)
		visible = visible && hiz_test(cd, pyr, c, radius);

	uint32_t oldVis = dvb[di];
	dec.oldVis = (uint8_t)(oldVis != 0 ? (oldVis == 1 ? 1 : 2) : 0);

	/* TASK_CULL == 1 (src/config.h:8) */
	if (visible && (!late || cd->clusterOcclusionEnabled == 1 || oldVis == 0 || cd->postPass != 0))
	{
		uint32_t lodIndex = 0;
		if (cd->lodEnabled == 1)
		{
			float distance = gl_maxf(length3(c) - radius, 0.0f);
			float threshold = distance * cd->lodTarget / d->scale;
			for (uint32_t i = 1; i < mesh->lodCount; ++i)
				if (mesh->lods[i].error < threshold)
					lodIndex = i;
		}
		dec.emit = 1;
		dec.lod = (uint8_t)lodIndex;
	}

	if (late)
		dvb[di] = visible ? 1u : 0u;
	return dec;
}

/* drawcull.comp.glsl:120-151: append at dci (the value the atomicAdd returned) */
static inline uint32_t drawcull_emit(int task, uint32_t di, uint32_t dci, uint32_t oldVisWord, const OrcMeshDraw* draws,
                                     const OrcMesh* meshes, uint32_t lodIndex, void* commands)
{
	const OrcMeshDraw* d = &draws[di];
	const OrcMesh* mesh = &meshes[d->meshIndex];
	const OrcMeshLod* lod = &mesh->lods[lodIndex];
	if (task)
	{
		OrcMeshTaskCommand* tc = (OrcMeshTaskCommand*)commands;
		uint32_t groups = (lod->meshletCount + TASK_WGSIZE - 1) / TASK_WGSIZE;
		if (dci + groups <= TASK_WGLIMIT)
			for (uint32_t i = 0; i < groups; ++i)
			{
				tc[dci + i].drawId = di;
				tc[dci + i].taskOffset = lod->meshletOffset + i * TASK_WGSIZE;
				tc[dci + i].taskCount = minu(TASK_WGSIZE, lod->meshletCount - i * TASK_WGSIZE);
				tc[dci + i].lateDrawVisibility = oldVisWord;
				tc[dci + i].meshletVisibilityOffset = d->meshletVisibilityOffset + i * TASK_WGSIZE;
			}
		return groups;
	}
	OrcMeshDrawCommand* dc = (OrcMeshDrawCommand*)commands;
	dc[dci].drawId = di;
	dc[dci].indexCount = lod->indexCount;
	dc[dci].instanceCount = 1;
	dc[dci].firstIndex = lod->indexOffset;
	dc[dci].vertexOffset = mesh->vertexOffset;
	dc[dci].firstInstance = 0;
	return 1;
}

static inline uint32_t drawcull_groups(int task, uint32_t di, const OrcMeshDraw* draws, const OrcMesh* meshes, uint32_t lodIndex)
{
	if (!task)
		return 1;
	const OrcMeshLod* lod = &meshes[draws[di].meshIndex].lods[lodIndex];
	return (lod->meshletCount + TASK_WGSIZE - 1) / TASK_WGSIZE;
}

/* drawcull.comp.glsl:54-156, invocations serialised in ascending gl_GlobalInvocationID.x */
void orc_drawcull(const OrcCullData* cd, int late, int task, const OrcMeshDraw* draws, const OrcMesh* meshes, void* commands,
                  uint32_t* count4, uint32_t* dvb, const OrcPyramid* pyr)
{
	for (uint32_t di = 0; di < cd->drawCount; ++di)
	{
		uint32_t oldWord = dvb[di];
		DrawDecision dec = drawcull_decide(cd, late, di, draws, meshes, dvb, pyr);
		if (dec.emit)
		{
			uint32_t dci = count4[0];
			count4[0] += drawcull_groups(task, di, draws, meshes, dec.lod);
			drawcull_emit(task, di, dci, oldWord, draws, meshes, dec.lod, commands);
		}
	}
}

/* tasksubmit.comp.glsl:27-47 */
void orc_tasksubmit(uint32_t* count4, OrcMeshTaskCommand* commands)
{
	uint32_t count = minu(count4[0], TASK_WGLIMIT);
	count4[1] = minu((count + 63) / 64, 65535);
	count4[2] = 64;
	count4[3] = 1;
	uint32_t boundary = (count + 63) & ~63u;
	for (uint32_t tid = 0; tid < 64; ++tid)
		if (count + tid < boundary)
			memset(&commands[count + tid], 0, sizeof(OrcMeshTaskCommand));
}

/* ------------------------------------------------------------------ clustercull */

typedef struct
{
	float c[3];
	float r;
	float axis[3];
	float cutoff;
} ClusterBounds;

/* clustercull.comp.glsl:72-80 */
static inline void cluster_bounds(const OrcCullData* cd, const OrcMeshDraw* d, const OrcMeshlet* m, ClusterBounds* b)
{
	float lc[3] = { orc_half_to_float(m->center[0]), orc_half_to_float(m->center[1]), orc_half_to_float(m->center[2]) };
	float rc[3], wc[3];
	orc_rotate_quat(lc, d->orientation, rc);
	wc[0] = rc[0] * d->scale + d->position[0];
	wc[1] = rc[1] * d->scale + d->position[1];
	wc[2] = rc[2] * d->scale + d->position[2];
	view_point(cd->view, wc, b->c);
	b->r = orc_half_to_float(m->radius) * d->scale;

	float la[3] = { (float)(int)m->cone_axis[0] / 127.0f, (float)(int)m->cone_axis[1] / 127.0f, (float)(int)m->cone_axis[2] / 127.0f };
	float ra[3];
	orc_rotate_quat(la, d->orientation, ra);
	view_dir(cd->view, ra, b->axis);
	b->cutoff = (float)(int)m->cone_cutoff / 127.0f;
}

/* one invocation of clustercull.comp.glsl:56-133 / meshlet.task.glsl:53-133; returns 1 if the lane appends.
 * bit updates go through __atomic so the _mt form can share words between threads. */
static inline int cluster_lane(const OrcCullData* cd, int late, const OrcMeshTaskCommand* cmd, const OrcMeshDraw* d,
                               const OrcMeshlet* meshlets, uint32_t mgi, uint32_t* mvb, const OrcPyramid* pyr)
{
	uint32_t mi = mgi + cmd->taskOffset;
	uint32_t mvi = mgi + cmd->meshletVisibilityOffset;

	ClusterBounds b;
	cluster_bounds(cd, d, &meshlets[mi], &b);

	int valid = mgi < cmd->taskCount;
	int visible = valid;
	int skip = 0;

	if (cd->clusterOcclusionEnabled == 1 && cd->postPass == 0)
	{
		uint32_t bit = __atomic_load_n(&mvb[mvi >> 5], __ATOMIC_RELAXED) & (1u << (mvi & 31));
		if (!late && bit == 0)
			visible = 0;
		if (late && cmd->lateDrawVisibility == 1 && bit != 0)
			skip = 1;
	}

	visible = visible && (cd->clusterBackfaceEnabled == 0 || !orc_cone_cull(b.c, b.r, b.axis, b.cutoff));
	visible = visible && frustum_test(cd, b.c, b.r);

	if (late && cd->clusterOcclusionEnabled == 1 && visible)
		visible = visible && hiz_test(cd, pyr, b.c, b.r);

	if (late && cd->clusterOcclusionEnabled == 1 && valid)
	{
		if (visible)
			__atomic_fetch_or(&mvb[mvi >> 5], 1u << (mvi & 31), __ATOMIC_RELAXED);
		else
			__atomic_fetch_and(&mvb[mvi >> 5], ~(1u << (mvi & 31)), __ATOMIC_RELAXED);
	}

	return visible && !skip;
}

/* number of commands an indirect dispatch of (groupCountX, 64, 1) covers: tasksubmit.comp.glsl:33-38,
 * clustercull.comp.glsl:59 */
static inline uint32_t indirect_commands(const uint32_t* count4)
{
	return count4[1] * 64u;
}

/* clustercull.comp.glsl:56-149, workgroups serialised by commandId, lanes by gl_LocalInvocationID.x */
void orc_clustercull(const OrcCullData* cd, int late, const OrcMeshTaskCommand* commands, const uint32_t* count4,
                     const OrcMeshDraw* draws, const OrcMeshlet* meshlets, uint32_t* mvb, const OrcPyramid* pyr,
                     uint32_t* clusterIndices, uint32_t* clusterCount4)
{
	uint32_t ncmd = indirect_commands(count4);
	for (uint32_t commandId = 0; commandId < ncmd; ++commandId)
	{
		const OrcMeshTaskCommand* cmd = &commands[commandId];
		const OrcMeshDraw* d = &draws[cmd->drawId];
		for (uint32_t mgi = 0; mgi < 64; ++mgi)
			if (cluster_lane(cd, late, cmd, d, meshlets, mgi, mvb, pyr))
			{
				uint32_t index = clusterCount4[0]++;
				if (index < CLUSTER_LIMIT)
					clusterIndices[index] = commandId | (mgi << 24);
			}
	}
}

/* clustersubmit.comp.glsl:25-45 */
void orc_clustersubmit(uint32_t* cc4, uint32_t* clusterIndices)
{
	uint32_t count = minu(cc4[0], CLUSTER_LIMIT);
	cc4[1] = CLUSTER_TILE;
	cc4[2] = minu((count + 255) / 256, 65535);
	cc4[3] = 256 / CLUSTER_TILE;
	uint32_t boundary = (count + 255) & ~255u;
	for (uint32_t tid = 0; tid < 256; ++tid)
		if (count + tid < boundary)
			clusterIndices[count + tid] = ~0u;
}

/* meshlet.task.glsl:53-149 (cull half): per-workgroup payload + EmitMeshTasksEXT count */
void orc_taskcull(const OrcCullData* cd, int late, const OrcMeshTaskCommand* commands, const uint32_t* count4,
                  const OrcMeshDraw* draws, const OrcMeshlet* meshlets, uint32_t* mvb, const OrcPyramid* pyr, uint32_t* payloads,
                  uint32_t* payloadCounts)
{
	uint32_t ncmd = indirect_commands(count4);
	for (uint32_t commandId = 0; commandId < ncmd; ++commandId)
	{
		const OrcMeshTaskCommand* cmd = &commands[commandId];
		const OrcMeshDraw* d = &draws[cmd->drawId];
		uint32_t shared = 0;
		for (uint32_t mgi = 0; mgi < 64; ++mgi)
			if (cluster_lane(cd, late, cmd, d, meshlets, mgi, mvb, pyr))
				payloads[(size_t)commandId * 64 + shared++] = commandId | (mgi << 24);
		payloadCounts[commandId] = shared;
	}
}

/* src/shaders/meshlet.mesh.glsl:91-116; index = gl_WorkGroupID.x + gl_WorkGroupID.y * 256 + gl_WorkGroupID.z * CLUSTER_TILE over the
 * grid clustersubmit wrote */
void orc_cluster_expand(const OrcMeshTaskCommand* commands, const OrcMeshlet* meshlets, const uint32_t* clusterIndices, const uint32_t* cc4,
                        uint32_t* records8, uint32_t capacity, uint64_t* totals3)
{
	for (uint32_t y = 0; y < cc4[2]; ++y)
		for (uint32_t z = 0; z < cc4[3]; ++z)
			for (uint32_t x = 0; x < cc4[1]; ++x)
			{
				uint32_t index = x + y * 256 + z * CLUSTER_TILE;
				uint32_t ci = clusterIndices[index];
				uint32_t rec[8] = { ~0u, 0, 0, 0, 0, 0, 0, 0 };
				if (ci != ~0u)
				{
					const OrcMeshTaskCommand* command = &commands[ci & 0xffffff];
					uint32_t mi = command->taskOffset + (ci >> 24);
					uint32_t vertexCount = meshlets[mi].vertexCount, triangleCount = meshlets[mi].triangleCount;
					uint32_t dataOffset = meshlets[mi].dataOffset;
					int shortRefs = meshlets[mi].shortRefs == 1;
					rec[0] = command->drawId;
					rec[1] = mi;
					rec[2] = vertexCount;
					rec[3] = triangleCount;
					rec[4] = dataOffset;
					rec[5] = dataOffset + (shortRefs ? (vertexCount + 1) / 2 : vertexCount);
					rec[6] = meshlets[mi].baseVertex;
					rec[7] = (uint32_t)shortRefs;
					totals3[0] += 1;
					totals3[1] += vertexCount;
					totals3[2] += triangleCount;
				}
				if (index < capacity)
					memcpy(records8 + (size_t)index * 8, rec, sizeof(rec));
			}
}

/* src/shaders/meshlet.mesh.glsl:91-198, CULL = 1.  One workgroup per grid slot; vertex phase (:121-160), barrier, triangle
 * phase (:166-205). */
void orc_trianglecull(const OrcGlobals* globals, const OrcMeshTaskCommand* commands, const OrcMeshDraw* draws, const OrcMeshlet* meshlets,
                      const uint32_t* meshletData, const OrcVertex* vertices, const uint32_t* clusterIndices, const uint32_t* cc4,
                      uint32_t* masks4, uint32_t capacity, uint64_t* totals3)
{
	const uint16_t* meshletData16 = (const uint16_t*)meshletData;
	const uint8_t* meshletData8 = (const uint8_t*)meshletData;
	const float* P = globals->projection;
	const float* V = globals->cullData.view;
	for (uint32_t y = 0; y < cc4[2]; ++y)
		for (uint32_t z = 0; z < cc4[3]; ++z)
			for (uint32_t x = 0; x < cc4[1]; ++x)
			{
				uint32_t index = x + y * 256 + z * CLUSTER_TILE;
				uint32_t ci = clusterIndices[index];
				uint32_t out[4] = { 0, 0, 0, 0 };
				if (ci != ~0u)
				{
					const OrcMeshTaskCommand* command = &commands[ci & 0xffffff];
					uint32_t mi = command->taskOffset + (ci >> 24);
					const OrcMeshDraw* meshDraw = &draws[command->drawId];
					uint32_t vertexCount = meshlets[mi].vertexCount, triangleCount = meshlets[mi].triangleCount;
					uint32_t dataOffset = meshlets[mi].dataOffset, baseVertex = meshlets[mi].baseVertex;
					int shortRefs = meshlets[mi].shortRefs == 1;
					uint32_t vertexOffset = dataOffset;
					uint32_t indexOffset = dataOffset + (shortRefs ? (vertexCount + 1) / 2 : vertexCount);
					float vertexClip[64][3];
					memset(vertexClip, 0, sizeof(vertexClip));
					for (uint32_t i = 0; i < vertexCount && i < 64; ++i)
					{
						uint32_t vi = shortRefs ? (uint32_t)meshletData16[vertexOffset * 2 + i] + baseVertex : meshletData[vertexOffset + i] + baseVertex;
						float position[3] = { orc_half_to_float(vertices[vi].vx), orc_half_to_float(vertices[vi].vy), orc_half_to_float(vertices[vi].vz) };
						float rot[3], wpos[3], v4[4], clip[4];
						orc_rotate_quat(position, meshDraw->orientation, rot);
						for (int k = 0; k < 3; ++k)
							wpos[k] = rot[k] * meshDraw->scale + meshDraw->position[k];
						/* clip = projection * (view * vec4(wpos, 1)); mat * vec = ((c0*x + c1*y) + c2*z) + c3*w */
						for (int r = 0; r < 4; ++r)
							v4[r] = ((V[r] * wpos[0] + V[4 + r] * wpos[1]) + V[8 + r] * wpos[2]) + V[12 + r] * 1.0f;
						for (int r = 0; r < 4; ++r)
							clip[r] = ((P[r] * v4[0] + P[4 + r] * v4[1]) + P[8 + r] * v4[2]) + P[12 + r] * v4[3];
						/* vertexClip[i] = vec3((clip.xy / clip.w * 0.5 + vec2(0.5)) * screen, clip.w) */
						vertexClip[i][0] = ((clip[0] / clip[3]) * 0.5f + 0.5f) * globals->screenWidth;
						vertexClip[i][1] = ((clip[1] / clip[3]) * 0.5f + 0.5f) * globals->screenHeight;
						vertexClip[i][2] = clip[3];
					}
					uint32_t kept = 0;
					for (uint32_t i = 0; i < triangleCount && i < 96; ++i)
					{
						uint32_t offset = indexOffset * 4 + i * 3;
						uint32_t a = meshletData8[offset], b = meshletData8[offset + 1], c = meshletData8[offset + 2];
						const float *pa = vertexClip[a & 63], *pb = vertexClip[b & 63], *pc = vertexClip[c & 63];
						int culled = 0;
						float ebx = pb[0] - pa[0], eby = pb[1] - pa[1];
						float ecx = pc[0] - pa[0], ecy = pc[1] - pa[1];
						culled = culled || (ebx * ecy <= eby * ecx);
						float bminx = gl_minf(pa[0], gl_minf(pb[0], pc[0])), bminy = gl_minf(pa[1], gl_minf(pb[1], pc[1]));
						float bmaxx = gl_maxf(pa[0], gl_maxf(pb[0], pc[0])), bmaxy = gl_maxf(pa[1], gl_maxf(pb[1], pc[1]));
						float sbprec = 1.0f / 256.0f;
						culled = culled || (rintf(bminx - sbprec) == rintf(bmaxx) || rintf(bminy) == rintf(bmaxy + sbprec));
						culled = culled && (pa[2] > 0 && pb[2] > 0 && pc[2] > 0);
						if (!culled)
						{
							out[i >> 5] |= 1u << (i & 31);
							kept++;
						}
					}
					out[3] = (triangleCount & 0xffu) | (vertexCount & 0xffu) << 8 | kept << 16;
					totals3[0] += 1;
					totals3[1] += triangleCount;
					totals3[2] += kept;
				}
				if (index < capacity)
					memcpy(masks4 + (size_t)index * 4, out, sizeof(out));
			}
}

void orc_probe_cluster_scalars(const OrcCullData* cd, const OrcMeshTaskCommand* commands, uint32_t commandCount,
                               const OrcMeshDraw* draws, const OrcMeshlet* meshlets, const OrcPyramid* pyr, float* out16)
{
	for (uint32_t commandId = 0; commandId < commandCount; ++commandId)
	{
		const OrcMeshTaskCommand* cmd = &commands[commandId];
		const OrcMeshDraw* d = &draws[cmd->drawId];
		for (uint32_t mgi = 0; mgi < 64; ++mgi)
		{
			float* o = out16 + ((size_t)commandId * 64 + mgi) * 16;
			memset(o, 0, 16 * sizeof(float));
			ClusterBounds b;
			cluster_bounds(cd, d, &meshlets[mgi + cmd->taskOffset], &b);
			o[0] = b.c[0];
			o[1] = b.c[1];
			o[2] = b.c[2];
			o[3] = b.r;
			o[4] = dot3(b.c, b.axis);
			o[5] = b.cutoff * length3(b.c) + b.r;
			float aabb[4];
			int proj = orc_project_sphere(b.c, b.r, cd->znear, cd->P00, cd->P11, aabb);
			if (proj)
			{
				o[6] = aabb[0];
				o[7] = aabb[1];
				o[8] = aabb[2];
				o[9] = aabb[3];
				o[10] = orc_occlusion_mip(aabb, cd->pyramidWidth, cd->pyramidHeight);
				if (pyr && pyr->base)
					o[11] = orc_sample_min(pyr, (aabb[0] + aabb[2]) * 0.5f, (aabb[1] + aabb[3]) * 0.5f, o[10]);
				o[12] = cd->znear / (b.c[2] - b.r);
			}
			o[13] = proj ? 1.0f : 0.0f;
			o[14] = frustum_test(cd, b.c, b.r) ? 1.0f : 0.0f;
			o[15] = orc_cone_cull(b.c, b.r, b.axis, b.cutoff) ? 1.0f : 0.0f;
		}
	}
}

/* ------------------------------------------------------------------ depth pyramid */

/* depthreduce.comp.glsl:14-22 for one level; host loop src/niagara.cpp:1713-1728 */
void orc_depthreduce(const float* depth, uint32_t w, uint32_t h, const OrcPyramid* pyr)
{
	const float* src = depth;
	uint32_t sw = w, sh = h;
	for (uint32_t i = 0; i < pyr->levels; ++i)
	{
		uint32_t lw = mip_dim(pyr->width, i), lh = mip_dim(pyr->height, i);
		float* dst = pyr->base + pyr->mipOffset[i];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (lw * lh > 65536)
#endif
		for (uint32_t y = 0; y < lh; ++y)
			for (uint32_t x = 0; x < lw; ++x)
			{
				float u = ((float)x + 0.5f) / (float)lw;
				float v = ((float)y + 0.5f) / (float)lh;
				dst[(size_t)y * lw + x] = orc_sample_min_image(src, sw, sh, u, v);
			}
		src = dst;
		sw = lw;
		sh = lh;
	}
}

/* ------------------------------------------------------------------ OpenMP baseline forms */

int orc_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* static chunks over commands, per-thread lists concatenated in chunk order: output identical to orc_clustercull */
void orc_clustercull_mt(const OrcCullData* cd, int late, const OrcMeshTaskCommand* commands, const uint32_t* count4,
                        const OrcMeshDraw* draws, const OrcMeshlet* meshlets, uint32_t* mvb, const OrcPyramid* pyr,
                        uint32_t* clusterIndices, uint32_t* clusterCount4, int threads)
{
	uint32_t ncmd = indirect_commands(count4);
	if (threads < 1)
		threads = 1;
	uint32_t** lists = (uint32_t**)calloc((size_t)threads, sizeof(uint32_t*));
	uint32_t* counts = (uint32_t*)calloc((size_t)threads, sizeof(uint32_t));
	uint32_t chunk = (ncmd + (uint32_t)threads - 1) / (uint32_t)threads;

#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(threads)
#endif
	for (int t = 0; t < threads; ++t)
	{
		uint32_t begin = minu((uint32_t)t * chunk, ncmd), end = minu(begin + chunk, ncmd);
		uint32_t* list = (uint32_t*)malloc((size_t)(end - begin) * 64 * sizeof(uint32_t) + 4);
		uint32_t n = 0;
		for (uint32_t commandId = begin; commandId < end; ++commandId)
		{
			const OrcMeshTaskCommand* cmd = &commands[commandId];
			const OrcMeshDraw* d = &draws[cmd->drawId];
			for (uint32_t mgi = 0; mgi < 64; ++mgi)
				if (cluster_lane(cd, late, cmd, d, meshlets, mgi, mvb, pyr))
					list[n++] = commandId | (mgi << 24);
		}
		lists[t] = list;
		counts[t] = n;
	}

	for (int t = 0; t < threads; ++t)
	{
		for (uint32_t i = 0; i < counts[t]; ++i)
		{
			uint32_t index = clusterCount4[0]++;
			if (index < CLUSTER_LIMIT)
				clusterIndices[index] = lists[t][i];
		}
		free(lists[t]);
	}
	free(lists);
	free(counts);
}

void orc_drawcull_mt(const OrcCullData* cd, int late, int task, const OrcMeshDraw* draws, const OrcMesh* meshes, void* commands,
                     uint32_t* count4, uint32_t* dvb, const OrcPyramid* pyr, int threads)
{
	uint32_t n = cd->drawCount;
	DrawDecision* dec = (DrawDecision*)malloc((size_t)n * sizeof(DrawDecision) + 1);
	uint32_t* oldWord = (uint32_t*)malloc((size_t)n * sizeof(uint32_t) + 4);
	if (threads < 1)
		threads = 1;

#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
	for (uint32_t di = 0; di < n; ++di)
	{
		oldWord[di] = dvb[di];
		dec[di] = drawcull_decide(cd, late, di, draws, meshes, dvb, pyr);
	}

	for (uint32_t di = 0; di < n; ++di)
		if (dec[di].emit)
		{
			uint32_t dci = count4[0];
			count4[0] += drawcull_groups(task, di, draws, meshes, dec[di].lod);
			drawcull_emit(task, di, dci, oldWord[di], draws, meshes, dec[di].lod, commands);
		}

	free(dec);
	free(oldWord);
}

/* ---------------------------------------------------------------------------------------------------------------
 * SURVEY.md §8(f) N2 — meshlet bounds: bounding sphere, normal cone and their fp16 / s8 quantisation, as niagara stores
 * them in a Meshlet (src/scene.cpp:69-85: meshopt_computeMeshletBounds, meshopt_quantizeHalf, cone_axis_s8 /
 * cone_cutoff_s8).
 *
 * PARITY UNPINNED.  The arithmetic is meshoptimizer's (https://github.com/zeux/meshoptimizer, .gitmodules:7-9, an
 * un-vendored submodule with no pinned commit): nothing under /root/reference computes or stores a meshlet bound, so
 * there is no golden vector to hold this against.  What follows restates the library's PUBLISHED algorithm
 * (clusterizer.cpp: meshopt_computeClusterBounds + computeBoundingSphere in its three-axis form; meshoptimizer.h:
 * meshopt_quantizeHalf, meshopt_quantizeSnorm) with DEFINED semantics — fp32, one IEEE operation per C operation, no
 * contraction, sums left to right, points visited in triangle order — and the tests pin what can be pinned without the
 * library: the HIP kernel equals this function bit for bit, every vertex lies inside the sphere and every triangle normal
 * inside the cone (up to the fp16 / s8 rounding that niagara's own call sites accept), and culling real geometry with the
 * generated bounds never removes a meshlet whose triangles a per-triangle test keeps (tests/test_meshlet_bounds.py).
 * Input positions are the fp16 vertex positions, dequantised, exactly like src/scene.cpp:193-198 feeds the library. */
static uint16_t orc_quantize_half(float v)
{
	/* meshopt_quantizeHalf: round to nearest (ties away in the mantissa sum), flush below 2^-14, saturate to inf */
	union { float f; uint32_t ui; } u = { v };
	uint32_t ui = u.ui;
	int s = (int)((ui >> 16) & 0x8000);
	int em = (int)(ui & 0x7fffffff);
	int h = (em - (112 << 23) + (1 << 12)) >> 13;
	h = (em < (113 << 23)) ? 0 : h;
	h = (em >= (143 << 23)) ? 0x7c00 : h;
	h = (em > (255 << 23)) ? 0x7e00 : h;
	return (uint16_t)(s | h);
}

static int orc_quantize_snorm8(float v)
{
	/* meshopt_quantizeSnorm(v, 8) */
	const float scale = 127.0f;
	float round = (v >= 0 ? 0.5f : -0.5f);
	v = (v >= -1) ? v : -1;
	v = (v <= +1) ? v : +1;
	return (int)(v * scale + round);
}

/* computeBoundingSphere (three-axis Ritter): extreme points per axis, the longest of the three segments as the first
 * diameter, then one pass that grows the sphere over every point outside it */
static void bounding_sphere(float result[4], const float (*points)[3], uint32_t count)
{
	uint32_t pmin[3] = { 0, 0, 0 }, pmax[3] = { 0, 0, 0 };
	for (uint32_t i = 0; i < count; ++i)
		for (int axis = 0; axis < 3; ++axis)
		{
			pmin[axis] = (points[i][axis] < points[pmin[axis]][axis]) ? i : pmin[axis];
			pmax[axis] = (points[i][axis] > points[pmax[axis]][axis]) ? i : pmax[axis];
		}
	float paxisd2 = 0;
	int paxis = 0;
	for (int axis = 0; axis < 3; ++axis)
	{
		const float* p1 = points[pmin[axis]];
		const float* p2 = points[pmax[axis]];
		float dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = p2[2] - p1[2];
		float d2 = (dx * dx + dy * dy) + dz * dz;
		if (d2 > paxisd2)
		{
			paxisd2 = d2;
			paxis = axis;
		}
	}
	const float* p1 = points[pmin[paxis]];
	const float* p2 = points[pmax[paxis]];
	float center[3] = { (p1[0] + p2[0]) / 2, (p1[1] + p2[1]) / 2, (p1[2] + p2[2]) / 2 };
	float radius = sqrtf(paxisd2) / 2;
	for (uint32_t i = 0; i < count; ++i)
	{
		const float* p = points[i];
		float dx = p[0] - center[0], dy = p[1] - center[1], dz = p[2] - center[2];
		float d2 = (dx * dx + dy * dy) + dz * dz;
		if (d2 > radius * radius)
		{
			float d = sqrtf(d2);
			float k = 0.5f + (radius / d) / 2;
			center[0] = center[0] * k + p[0] * (1 - k);
			center[1] = center[1] * k + p[1] * (1 - k);
			center[2] = center[2] * k + p[2] * (1 - k);
			radius = (radius + d) / 2;
		}
	}
	result[0] = center[0], result[1] = center[1], result[2] = center[2], result[3] = radius;
}

/* fills center / radius / cone_axis / cone_cutoff of meshlets[0..count); out8 (optional) receives the unquantised
 * {center.xyz, radius, axis.xyz, cutoff} per meshlet for the tests */
void orc_meshlet_bounds(const OrcVertex* vertices, const uint32_t* meshletData, OrcMeshlet* meshlets, uint32_t count, float* out8)
{
	const uint16_t* data16 = (const uint16_t*)meshletData;
	const uint8_t* data8 = (const uint8_t*)meshletData;
	for (uint32_t mi = 0; mi < count; ++mi)
	{
		OrcMeshlet* m = &meshlets[mi];
		uint32_t vertexCount = m->vertexCount < 64 ? m->vertexCount : 64, triangleCount = m->triangleCount < 96 ? m->triangleCount : 96;
		int shortRefs = m->shortRefs == 1;
		uint32_t indexOffset = m->dataOffset + (shortRefs ? (vertexCount + 1) / 2 : vertexCount);
		float pos[64][3];
		memset(pos, 0, sizeof(pos));
		for (uint32_t i = 0; i < vertexCount; ++i)
		{
			uint32_t vi = (shortRefs ? (uint32_t)data16[m->dataOffset * 2 + i] : meshletData[m->dataOffset + i]) + m->baseVertex;
			pos[i][0] = orc_half_to_float(vertices[vi].vx);
			pos[i][1] = orc_half_to_float(vertices[vi].vy);
			pos[i][2] = orc_half_to_float(vertices[vi].vz);
		}
		/* triangle normals and corners, degenerate triangles dropped (meshopt_computeClusterBounds) */
		static _Thread_local float normals[96][3];
		static _Thread_local float corners[96 * 3][3];
		uint32_t triangles = 0;
		for (uint32_t i = 0; i < triangleCount; ++i)
		{
			uint32_t a = data8[indexOffset * 4 + i * 3 + 0] & 63, b = data8[indexOffset * 4 + i * 3 + 1] & 63, c = data8[indexOffset * 4 + i * 3 + 2] & 63;
			const float *p0 = pos[a], *p1 = pos[b], *p2 = pos[c];
			float p10[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };
			float p20[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
			float nx = p10[1] * p20[2] - p10[2] * p20[1];
			float ny = p10[2] * p20[0] - p10[0] * p20[2];
			float nz = p10[0] * p20[1] - p10[1] * p20[0];
			float area = sqrtf((nx * nx + ny * ny) + nz * nz);
			if (area == 0.0f)
				continue;
			normals[triangles][0] = nx / area, normals[triangles][1] = ny / area, normals[triangles][2] = nz / area;
			memcpy(corners[triangles * 3 + 0], p0, 12);
			memcpy(corners[triangles * 3 + 1], p1, 12);
			memcpy(corners[triangles * 3 + 2], p2, 12);
			triangles++;
		}
		float center[3] = { 0, 0, 0 }, radius = 0, axis[3] = { 0, 0, 0 }, cutoff = 0;
		int axis_s8[3] = { 0, 0, 0 }, cutoff_s8 = 0;
		if (triangles)
		{
			float psphere[4], nsphere[4];
			bounding_sphere(psphere, corners, triangles * 3);
			bounding_sphere(nsphere, normals, triangles);
			center[0] = psphere[0], center[1] = psphere[1], center[2] = psphere[2], radius = psphere[3];
			axis[0] = nsphere[0], axis[1] = nsphere[1], axis[2] = nsphere[2];
			float axislength = sqrtf((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
			float invaxislength = axislength == 0.0f ? 0.0f : 1.0f / axislength;
			axis[0] *= invaxislength, axis[1] *= invaxislength, axis[2] *= invaxislength;
			float mindp = 1.0f;
			for (uint32_t i = 0; i < triangles; ++i)
			{
				float dp = (normals[i][0] * axis[0] + normals[i][1] * axis[1]) + normals[i][2] * axis[2];
				mindp = (dp < mindp) ? dp : mindp;
			}
			if (mindp <= 0.1f)
			{
				/* normal cone wider than a hemisphere: no cone (cutoff 1 never culls) */
				axis[0] = axis[1] = axis[2] = 0;
				cutoff = 1;
				cutoff_s8 = 127;
			}
			else
			{
				cutoff = sqrtf(1 - mindp * mindp);
				float e = 0;
				for (int k = 0; k < 3; ++k)
				{
					axis_s8[k] = orc_quantize_snorm8(axis[k]);
					e += fabsf((float)axis_s8[k] / 127.0f - axis[k]);
				}
				/* rounded up so that the 8-bit test stays conservative */
				int c8 = (int)(127 * (cutoff + e) + 1);
				cutoff_s8 = c8 > 127 ? 127 : c8;
			}
		}
		m->center[0] = orc_quantize_half(center[0]);
		m->center[1] = orc_quantize_half(center[1]);
		m->center[2] = orc_quantize_half(center[2]);
		m->radius = orc_quantize_half(radius);
		m->cone_axis[0] = (int8_t)axis_s8[0], m->cone_axis[1] = (int8_t)axis_s8[1], m->cone_axis[2] = (int8_t)axis_s8[2];
		m->cone_cutoff = (int8_t)cutoff_s8;
		if (out8)
		{
			float* o = out8 + (size_t)mi * 8;
			o[0] = center[0], o[1] = center[1], o[2] = center[2], o[3] = radius, o[4] = axis[0], o[5] = axis[1], o[6] = axis[2], o[7] = cutoff;
		}
	}
}
