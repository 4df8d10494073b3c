// hibag_ubench.hip -- the instruction costs the issue floor of the prediction kernels is priced with, measured on the
// device the process runs on (bench.py calls this once per run instead of trusting numbers from another box or round:
// the shader clock under an FP64 / matrix load differs by up to 10 % between MI355X boards).
//
// Every kernel keeps 8 wavefronts on every SIMD of the chip busy with one kind of instruction and reports the time per
// wave64 instruction per SIMD:
//   v_mul_f64 / v_add_f64 (what `cell += prod * TAB[d]` of src/LibHLA.cpp:1786-1813 costs per haplotype pair and pass),
//   v_mfma_i32_32x32x32_i8 (the distance dot product of 31..32-SNP classifiers), v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1
//   operands (up to 30 SNPs), and a half / half mix of FP64 and int8 MFMA wavefronts, which shows whether the matrix pipe
//   hides behind the FP64 work (it does not on gfx950: the mix takes ~88 % of the serial sum).
// The stand-alone tools/ubench_*.hip print the same numbers with more context.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hibag_hip.h"

int hibag_fail(int code, const char *fmt, ...);

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// mode 0: int8 MFMA, 1: FP64 mul / add, 2: odd wavefronts MFMA and even ones FP64, 3: FP4 MFMA
__global__ __launch_bounds__(512) void k_issue(int mode, int iters, int *out)
{
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1));
	v16i a0 = {}, a1 = {};
	v16f f0 = {}, f1 = {};
	v4i x = {(int)threadIdx.x, 1, 2, 3}, y = {3, 2, 1, (int)threadIdx.x};
	double d0 = threadIdx.x, d1 = 1.5, d2 = 2.5, d3 = 3.5;
	if (mode == 3) {
		const v8i x8 = {x[0] & 0x22222222, 0x22222222, 0x02020202, 0x20202020, 0, 0, 0, 0}, y8 = {0x22222222, y[3] & 0x22222222, 0x20202020, 0x02020202, 0, 0, 0, 0};
		for (int i = 0; i < iters; i++) {
			f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x8, y8, f0, 4, 4, 0, 100, 0, 100);
			f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(y8, x8, f1, 4, 4, 0, 100, 0, 100);
			f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x8, y8, f0, 4, 4, 0, 100, 0, 100);
			f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(y8, x8, f1, 4, 4, 0, 100, 0, 100);
		}
	} else if (do_mfma) {
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
	for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + (int)(f0[r] + f1[r]);
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)(d0 + d1 + d2 + d3);
}

} // namespace

extern "C" int hibag_hip_measure_issue_costs(double *fp64_op_ns, double *mfma_i8_ns, double *mfma_fp4_ns, double *mix_frac_of_serial)
{
	int dev = 0, cus = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
		return hibag_fail(HIBAG_HIP_ENODEV, "no HIP device to measure on");
	int *d_out = nullptr;
	const int blocks = cus * 4;                         // 512 threads = 8 wavefronts each: 8 wavefronts on every SIMD
	if (hipMalloc((void **)&d_out, (size_t)blocks * 512 * sizeof(int)) != hipSuccess) return hibag_fail(HIBAG_HIP_ENOMEM, "out of device memory");
	hipEvent_t a = nullptr, b = nullptr;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	const int iters = 1500;
	float t[4] = {0, 0, 0, 0};
	int rc = 0;
	for (int mode = 0; mode < 4 && !rc; mode++) {
		hipLaunchKernelGGL(k_issue, dim3(blocks), dim3(512), 0, 0, mode, 10, d_out);          // warm-up (code object, clocks)
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(a, 0);
		hipLaunchKernelGGL(k_issue, dim3(blocks), dim3(512), 0, 0, mode, iters, d_out);
		(void)hipEventRecord(b, 0);
		if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t[mode], a, b) != hipSuccess)
			rc = hibag_fail(HIBAG_HIP_ENODEV, "the issue-cost kernels did not run: %s", hipGetErrorString(hipGetLastError()));
	}
	(void)hipEventDestroy(a); (void)hipEventDestroy(b);
	(void)hipFree(d_out);
	if (rc) return rc;
	// per SIMD: 8 wavefronts x iters x (4 MFMA | 28 FP64 operations)
	if (mfma_i8_ns) *mfma_i8_ns = t[0] * 1e6 / (8.0 * iters * 4);
	if (fp64_op_ns) *fp64_op_ns = t[1] * 1e6 / (8.0 * iters * 28);
	if (mfma_fp4_ns) *mfma_fp4_ns = t[3] * 1e6 / (8.0 * iters * 4);
	if (mix_frac_of_serial) *mix_frac_of_serial = t[2] / ((t[0] + t[1]) / 2);
	return 0;
}
