// Micro-benchmark: do v_mfma_i32_32x32x32_i8 and FP64 VALU work issued by DIFFERENT
// wavefronts of one SIMD overlap (gfx950)?  8 waves per SIMD; mode 0: all run the MFMA
// loop, mode 1: all run the FP64 loop, mode 2: odd waves MFMA, even waves FP64.
// Serialised pipes would give T2 ~ (T0 + T1) / 2, overlapped ones ~ max(T0, T1) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, int *out)
{
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1));
	v16i a0 = {}, a1 = {};
	v4i x = {(int)threadIdx.x, 1, 2, 3}, y = {3, 2, 1, (int)threadIdx.x};
	double d0 = threadIdx.x, d1 = 1.5, d2 = 2.5, d3 = 3.5;
	if (do_mfma) {
		for (int i = 0; i < iters; i++) {
			a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a0, 0, 0, 0);
			a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(y, x, a1, 0, 0, 0);
			a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a0, 0, 0, 0);
			a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(y, x, a1, 0, 0, 0);
		}
	} else {
		for (int i = 0; i < iters; i++) {
			asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
				"v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
				"v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
				"v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
				"v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
				"v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
				"v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
				: "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));
		}
	}
	int s = 0;
	for (int r = 0; r < 16; r++) s += a0[r] + a1[r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)(d0 + d1 + d2 + d3);
}

int main()
{
	int *d_out;
	(void)hipMalloc(&d_out, 256 * 4 * 512 * sizeof(int));
	const int iters = 4000;
	float t[3];
	for (int mode = 0; mode < 3; mode++) {
		hipEvent_t a, b;
		(void)hipEventCreate(&a); (void)hipEventCreate(&b);
		hipLaunchKernelGGL(k, dim3(256 * 4), dim3(512), 0, 0, mode, 10, d_out);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(a);
		hipLaunchKernelGGL(k, dim3(256 * 4), dim3(512), 0, 0, mode, iters, d_out);
		(void)hipEventRecord(b);
		(void)hipEventSynchronize(b);
		(void)hipEventElapsedTime(&t[mode], a, b);
	}
	// per SIMD: 8 waves; mode 0: 8 waves x iters x 4 MFMA; mode 1: 8 waves x iters x 28 FP64 ops
	printf("all MFMA  : %8.3f ms  -> %6.1f ns per MFMA per SIMD\n", t[0], t[0] * 1e6 / (8.0 * iters * 4));
	printf("all FP64  : %8.3f ms  -> %6.2f ns per FP64 op per SIMD\n", t[1], t[1] * 1e6 / (8.0 * iters * 28));
	printf("half/half : %8.3f ms   serialised would be %.3f, overlapped %.3f\n", t[2], (t[0] + t[1]) / 2, (t[0] > t[1] ? t[0] : t[1]) / 2);
	return 0;
}
