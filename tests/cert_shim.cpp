// cert_shim.cpp — host-side window onto niagara_amd/csrc/filtermath.h for tests/test_cert_margins.py: the SAME source the kernels compile
// (filter_make, filter_k, filter_view_norms and the constants), built with g++ -ffp-contract=off at test time.  Test infrastructure.
#include <cstddef>

#include "../niagara_amd/csrc/filtermath.h"

extern "C" {

// {K, u, slack, floor, radius_u, magnitude_max, scale_min, coeff_max, plane_max, coneK slope, coneK offset, 1/127}
void shim_constants(float out[12])
{
	const float c[12] = { nv::FILTER_K, nv::FILTER_U, nv::FILTER_SLACK, nv::FILTER_FLOOR, nv::FILTER_RADIUS_U, nv::FILTER_MAGNITUDE_MAX, nv::FILTER_SCALE_MIN,
		                  nv::FILTER_COEFF_MAX, nv::FILTER_PLANE_MAX, nv::CONE_K_SLOPE, nv::CONE_K_OFFSET, nv::INV_127 };
	for (int i = 0; i < 12; ++i)
		out[i] = c[i];
}

float shim_filter_k(const float frustum[4], float znear, float zfar) { return nv::filter_k(frustum, znear, zfar); }

int shim_fold_sound(const float frustum[4]) { return nv::filter_fold_sound(frustum) ? 1 : 0; }

void shim_view_norms(const float view[16], float out[3]) { nv::filter_view_norms(view, &out[0], &out[1], &out[2]); }

// draws: n x {position.xyz, scale, orientation.xyzw} (the first 32 bytes of a MeshDraw, stride in floats); out: n x 19 floats
// {m[9], b[3], aK, bK, aR, scale, coneK, is127, tK}
void shim_make_filters(const float view[16], const float* draws, unsigned stride, unsigned n, float filterK, float vmax3, float rmax, float* out)
{
	float Vn, V3n, sumV;
	nv::filter_view_norms(view, &Vn, &V3n, &sumV);
	for (unsigned i = 0; i < n; ++i)
	{
		const float* d = draws + (size_t)i * stride;
		const nv::FilterDraw f = nv::filter_make(view, d[4], d[5], d[6], d[7], d[3], d[0], d[1], d[2], filterK, Vn, V3n, sumV, vmax3, rmax);
		float* o = out + (size_t)i * 19;
		for (int k = 0; k < 9; ++k)
			o[k] = f.m[k];
		for (int k = 0; k < 3; ++k)
			o[9 + k] = f.b[k];
		o[12] = f.aK;
		o[13] = f.bK;
		o[14] = f.aR;
		o[15] = f.scale;
		o[16] = f.coneK;
		o[17] = f.is127;
		o[18] = f.tK;
	}
}
}
