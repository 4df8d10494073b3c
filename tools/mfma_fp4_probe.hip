#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// mode 0: correctness / layout, mode 1: timing
__global__ void k2(const uint32_t *A, const uint32_t *B, float *D)
{
	const int l = threadIdx.x;
	v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
	for (int i = 0; i < 4; i++) { a[i] = (int)A[l * 4 + i]; b[i] = (int)B[l * 4 + i]; }
	v16f c = {};
	const int sb = l < 32 ? 54 : 55;      // per-lane scale: the upper K half of B counts twice
	c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 54, 0, sb);
	for (int r = 0; r < 16; r++) D[l * 16 + r] = c[r];
}

// two K steps chained through the accumulator: does a denormal C operand survive?
__global__ void k3(const uint32_t *A, const uint32_t *B, float *D)
{
	const int l = threadIdx.x;
	v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
	for (int i = 0; i < 4; i++) { a[i] = (int)A[l * 4 + i]; b[i] = (int)B[l * 4 + i]; }
	v16f c = {};
	const int sb = l < 32 ? 54 : 55;
	c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 54, 0, sb);
	c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 54, 0, sb);
	c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 54, 0, sb);
	for (int r = 0; r < 16; r++) D[l * 16 + r] = c[r];
}

__global__ void k(const uint32_t *A, const uint32_t *B, float *D, int sa, int sb)
{
	const int l = threadIdx.x;
	v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
	for (int i = 0; i < 4; i++) { a[i] = (int)A[l * 4 + i]; b[i] = (int)B[l * 4 + i]; }
	v16f c = {};
	c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, sa, 0, sb);
	for (int r = 0; r < 16; r++) D[l * 16 + r] = c[r];
}

__global__ __launch_bounds__(256) void kt(int iters, float *out)
{
	v8i a = {(int)threadIdx.x, 2, 3, 4, 0, 0, 0, 0}, b = {5, 6, (int)threadIdx.x, 8, 0, 0, 0, 0};
	v16f c0 = {}, c1 = {};
	for (int i = 0; i < iters; i++) {
		c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
		c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 4, 4, 0, 127, 0, 127);
		c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
		c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 4, 4, 0, 127, 0, 127);
	}
	float s = 0;
	for (int r = 0; r < 16; r++) s += c0[r] + c1[r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
	// A[row][k]: fp4 code per element; lane l holds row l%32, K elements 32*(l/32) .. +31 (assumed), 8 nibbles per dword
	uint32_t hA[64 * 4], hB[64 * 4];
	float hD[64 * 16];
	// values: A[row][kk] = 1.0 (code 2) if kk <= row else 0  -> row sums = row+1 ; B[kk][col] = 1.0 for kk < 64 except col-dependent: B = 2.0 (code 4) if kk == col else 1.0
	for (int l = 0; l < 64; l++) {
		const int row = l % 32, kh = l / 32;
		for (int d = 0; d < 4; d++) {
			uint32_t wa = 0, wb = 0;
			for (int n = 0; n < 8; n++) {
				const int kk = 32 * kh + 8 * d + n;
				const uint32_t ca = kk <= row ? 2u : 0u;
				const uint32_t cb = kk == row ? 4u : 2u;      // here `row` plays the column for B's lane
				wa |= ca << (4 * n); wb |= cb << (4 * n);
			}
			hA[l * 4 + d] = wa; hB[l * 4 + d] = wb;
		}
	}
	uint32_t *dA, *dB; float *dD;
	hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
	hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
	for (int trial = 0; trial < 2; trial++) {
		const int sa = trial ? 54 : 127, sb = trial ? 54 : 127;
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
		hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
		// expected D[row][col] = sum_kk A[row][kk]*B[kk][col] = (row+1) + (col <= row ? 1 : 0)
		int bad = 0;
		for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
			const int col = l % 32, row = 8 * (r / 4) + 4 * (l / 32) + r % 4;
			const double want = (row + 1) + (col <= row ? 1 : 0);
			uint32_t bits; memcpy(&bits, &hD[l * 16 + r], 4);
			const double got = trial ? (double)bits / 8.0 : hD[l * 16 + r];      // scales 54 + 54: 2^-146 = 8 * 2^-149
			if (got != want) { if (bad < 5) printf("  mismatch lane %d reg %d: got %g (bits %08x) want %g\n", l, r, got, bits, want); bad++; }
		}
		printf("trial %d (scales %d,%d): %s (%d mismatches)%s\n", trial, sa, sb, bad ? "MISMATCH" : "OK", bad, trial ? "  [denormal results: bit pattern = 8 x the dot product]" : "");
	}
	{
		// trial 2: A[row][kk] in {0,1,2,4} codes {0,2,4,6}; B[kk][col] in {+1,-1,2,0} codes {2,0xA,4,0}; upper K half scaled x2
		static const uint32_t acode[4] = {0, 2, 4, 6}; static const double aval[4] = {0, 1, 2, 4};
		static const uint32_t bcode[4] = {2, 0xA, 4, 0}; static const double bval[4] = {1, -1, 2, 0};
		auto ai = [](int row, int kk) { return (row * 7 + kk * 3 + (kk >> 2)) & 3; };
		auto bi = [](int kk, int col) { return (col * 5 + kk + (kk >> 3)) & 3; };
		for (int l = 0; l < 64; l++) for (int d = 0; d < 4; d++) {
			uint32_t wa = 0, wb = 0;
			for (int n = 0; n < 8; n++) {
				const int kk = 32 * (l / 32) + 8 * d + n;
				wa |= acode[ai(l % 32, kk)] << (4 * n); wb |= bcode[bi(kk, l % 32)] << (4 * n);
			}
			hA[l * 4 + d] = wa; hB[l * 4 + d] = wb;
		}
		hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dA, dB, dD);
		hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
		int bad = 0, neg = 0;
		for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
			const int col = l % 32, row = 8 * (r / 4) + 4 * (l / 32) + r % 4;
			double want = 0;
			for (int kk = 0; kk < 64; kk++) want += aval[ai(row, kk)] * bval[bi(kk, col)] * (kk < 32 ? 1 : 2);
			uint32_t bits; memcpy(&bits, &hD[l * 16 + r], 4);
			if (want < 0) { neg++; if ((bits & 0x7FFFFFFFu) != (uint32_t)(-8 * want) || !(bits >> 31)) bad++; continue; }
			if (bits != (uint32_t)(8 * want)) { if (bad < 5) printf("  mismatch lane %d reg %d: bits %08x want %g\n", l, r, bits, 8 * want); bad++; }
		}
		printf("trial 2 (per-lane B scale 54/55, signed operands): %s (%d mismatches, %d negative sums)\n", bad ? "MISMATCH" : "OK", bad, neg);
		// trial 3: the same product three times through the C operand: bits = 3 x 8 x the dot product if denormal accumulators survive
		hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, 0, dA, dB, dD);
		hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
		int bad3 = 0;
		for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
			const int col = l % 32, row = 8 * (r / 4) + 4 * (l / 32) + r % 4;
			double want = 0;
			for (int kk = 0; kk < 64; kk++) want += aval[ai(row, kk)] * bval[bi(kk, col)] * (kk < 32 ? 1 : 2);
			uint32_t bits; memcpy(&bits, &hD[l * 16 + r], 4);
			const uint32_t w = (uint32_t)(24 * (want < 0 ? -want : want)) | (want < 0 ? 0x80000000u : 0u);
			if (bits != w) { if (bad3 < 5) printf("  chained: lane %d reg %d: bits %08x want %08x\n", l, r, bits, w); bad3++; }
		}
		printf("trial 3 (three K steps chained through a denormal accumulator): %s (%d mismatches)\n", bad3 ? "MISMATCH" : "OK", bad3);
	}
	float *dOut; hipMalloc(&dOut, 256 * 8 * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(kt, dim3(256 * 8), dim3(256), 0, 0, 10, dOut); hipDeviceSynchronize();
	hipEventRecord(e0);
	const int iters = 4000;
	hipLaunchKernelGGL(kt, dim3(256 * 8), dim3(256), 0, 0, iters, dOut);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("fp4 32x32x64: %.3f ms -> %.1f ns per MFMA per SIMD (8 waves/SIMD)\n", ms, ms * 1e6 / (8.0 * iters * 4));
	return 0;
}
