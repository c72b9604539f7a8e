// launch_cost.hip — what a dependent kernel boundary costs in WALL time on this box.
//
// Round 3 priced a launch at "4.2 us" from the kernel-trace duration of a one-thread kernel; that is a begin -> end stamp of one
// dispatch, not what a second dependent launch adds to a stream.  This tool measures the latter: chains of K dependent launches on one
// stream, host wall clock around (enqueue K x R launches + one synchronize) divided by K x R, for
//   * trivial kernels (1 workgroup of 64 threads; 256 workgroups of 256 threads),
//   * the same behind a streaming kernel of ~20 us (a 128 MB read-reduce), i.e. stream = [big, small x K] repeated R times: the
//     per-boundary price is (wall(K) - wall(0)) / K per repetition,
//   * and two streaming kernels back to back (what a cull launch + scatter launch are).
// Prints one JSON line per case.  hipcc --offload-arch=gfx950 -O3 tools/launch_cost.hip -o tools/launch_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
	do                                                                                            \
	{                                                                                             \
		hipError_t e_ = (x);                                                                      \
		if (e_ != hipSuccess)                                                                     \
		{                                                                                         \
			fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(1);                                                                              \
		}                                                                                         \
	} while (0)

__global__ void trivial_kernel(unsigned* p)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
		p[0] += 1u; // dependent on the previous launch's store
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, size_t n16, unsigned* __restrict__ out)
{
	unsigned acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
	{
		const uint4 v = src[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u)
		out[1] = acc;
}

static double now_us()
{
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
	const int R = argc > 1 ? atoi(argv[1]) : 200;
	unsigned* d = nullptr;
	CHECK(hipMalloc(&d, 64));
	CHECK(hipMemset(d, 0, 64));
	const size_t copies = 4, bytes = 128u << 20; // four 128 MB sources rotated: cold reads (Infinity Cache = 256 MiB)
	std::vector<uint4*> src(copies);
	for (size_t c = 0; c < copies; ++c)
	{
		CHECK(hipMalloc(&src[c], bytes));
		CHECK(hipMemset(src[c], (int)c + 1, bytes));
	}
	hipStream_t s;
	CHECK(hipStreamCreate(&s));
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;

	auto run = [&](int big, int K, int wgs, int threads) -> double
	{
		// warm-up
		for (int r = 0; r < 5; ++r)
		{
			if (big)
				hipLaunchKernelGGL(stream_kernel, dim3(cus * 8), dim3(256), 0, s, src[r % copies], bytes / 16, d);
			for (int k = 0; k < K; ++k)
				hipLaunchKernelGGL(trivial_kernel, dim3(wgs), dim3(threads), 0, s, d);
		}
		CHECK(hipStreamSynchronize(s));
		double best = 1e30;
		for (int rep = 0; rep < 5; ++rep)
		{
			const double t0 = now_us();
			for (int r = 0; r < R; ++r)
			{
				for (int b = 0; b < big; ++b)
					hipLaunchKernelGGL(stream_kernel, dim3(cus * 8), dim3(256), 0, s, src[(r * big + b) % copies], bytes / 16, d);
				for (int k = 0; k < K; ++k)
					hipLaunchKernelGGL(trivial_kernel, dim3(wgs), dim3(threads), 0, s, d);
			}
			CHECK(hipStreamSynchronize(s));
			const double t = (now_us() - t0) / R;
			best = t < best ? t : best;
		}
		return best; // us per repetition
	};

	for (int shape = 0; shape < 2; ++shape)
	{
		const int wgs = shape ? 256 : 1, threads = shape ? 256 : 64;
		for (int K : { 1, 2, 4, 8, 16 })
		{
			const double t = run(0, K, wgs, threads);
			printf("{\"case\": \"chain of K trivial launches\", \"workgroups\": %d, \"threads\": %d, \"K\": %d, \"us_per_chain\": %.3f, \"us_per_launch\": %.3f}\n", wgs, threads, K, t, t / K);
		}
	}
	const double big1 = run(1, 0, 1, 64), big2 = run(2, 0, 1, 64);
	printf("{\"case\": \"streaming kernel alone (128 MB read, cold)\", \"us\": %.3f, \"TBps\": %.3f}\n", big1, bytes / big1 / 1e6);
	printf("{\"case\": \"two streaming kernels back to back\", \"us\": %.3f, \"us_second_minus_first\": %.3f}\n", big2, big2 - big1);
	for (int shape = 0; shape < 2; ++shape)
	{
		const int wgs = shape ? 256 : 1, threads = shape ? 256 : 64;
		for (int K : { 1, 2, 4, 8 })
		{
			const double t = run(1, K, wgs, threads);
			printf("{\"case\": \"streaming kernel + K dependent trivial launches\", \"workgroups\": %d, \"threads\": %d, \"K\": %d, \"us\": %.3f, \"us_per_added_launch\": %.3f}\n", wgs, threads, K,
			       t, (t - big1) / K);
		}
	}
	return 0;
}
