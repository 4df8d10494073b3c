// Probe: operand/result layout of v_mfma_i32_32x32x32_i8 and v_permlane32_swap on gfx950.
// A[i][k] = (i == probe_row && k == probe_k), B[k][j] = j + 1 if k == probe_k  => D[probe_row][j] = j + 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const v4i *A, const v4i *B, int *D, int *S)
{
	const int l = threadIdx.x;
	v16i acc = {0};
	acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[l], B[l], acc, 0, 0, 0);
	for (int r = 0; r < 16; r++) D[r * 64 + l] = acc[r];
	// permlane32_swap probe: v0 = lane, v1 = 100 + lane
	int v0 = l, v1 = 100 + l;
	auto sw = __builtin_amdgcn_permlane32_swap(v0, v1, false, false);
	S[l] = sw[0]; S[64 + l] = sw[1];
}

int main()
{
	std::vector<signed char> A(64 * 16, 0), B(64 * 16, 0);
	// hypothesis: lane l supplies row/col l%32, k = 16*(l/32) + byte
	auto setA = [&](int i, int k, int v) { A[(size_t)((k / 16) * 32 + i) * 16 + (k % 16)] = (signed char)v; };
	auto setB = [&](int k, int j, int v) { B[(size_t)((k / 16) * 32 + j) * 16 + (k % 16)] = (signed char)v; };
	// A = rows i with value (i+1) at k = i (so D[i][j] = (i+1) * B[i][j]); B[k][j] = (j + 2 if k < 32)
	for (int i = 0; i < 32; i++) setA(i, i, 1);
	for (int k = 0; k < 32; k++) for (int j = 0; j < 32; j++) setB(k, j, ((k * 3 + j) % 7) - 3);
	v4i *dA, *dB; int *dD, *dS;
	hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 16 * 64 * 4); hipMalloc(&dS, 128 * 4);
	hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
	std::vector<int> D(16 * 64), S(128);
	hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(S.data(), dS, 512, hipMemcpyDeviceToHost);
	// expected D[i][j] = B[i][j] (A = identity on k = i)
	int bad = 0;
	for (int r = 0; r < 16; r++) for (int l = 0; l < 64; l++) {
		const int row = 8 * (r / 4) + 4 * (l / 32) + (r % 4), col = l % 32;
		const int want = ((row * 3 + col) % 7) - 3;
		if (D[r * 64 + l] != want) { if (bad < 8) printf("mismatch r=%d lane=%d got %d want %d\n", r, l, D[r * 64 + l], want); bad++; }
	}
	printf("mfma layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
	printf("permlane32_swap: out0 lanes 0,31,32,63 = %d %d %d %d ; out1 = %d %d %d %d\n", S[0], S[31], S[32], S[63], S[64], S[95], S[96], S[127]);
	return 0;
}
