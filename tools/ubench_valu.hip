// Micro-benchmark: issue rate of the VALU instructions the pair loop is made of
// (gfx950).  Every wave runs `iters` x 32 independent copies of one instruction;
// all SIMDs are loaded with 8 waves each.  Prints wave-instructions per SIMD
// per microsecond and the ratio to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out)
{
	unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 0x55555555u ^ threadIdx.x;
	unsigned r0 = a, r1 = b, r2 = c, r3 = a + 1;
	double d0 = a + 1.5, d1 = b + 0.25, d2 = 1.0000001, d3 = 0.5;
	float f0 = a, f1 = 1.0001f, f2 = 0.5f, f3 = 2.0f;
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3" : "+v"(f0), "+v"(f1) : "v"(f2), "v"(f3), "v"(f2), "v"(f3));) }
		if (KIND == 1) { REP8(asm volatile("v_bcnt_u32_b32 %0, %2, %0\n v_bcnt_u32_b32 %1, %3, %1\n v_bcnt_u32_b32 %4, %2, %4\n v_bcnt_u32_b32 %5, %3, %5" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3));) }
		if (KIND == 2) { REP8(asm volatile("v_bitop3_b32 %0, %2, %3, %0 bitop3:0x48\n v_bitop3_b32 %1, %3, %2, %1 bitop3:0x48\n v_bitop3_b32 %4, %2, %3, %4 bitop3:0x48\n v_bitop3_b32 %5, %3, %2, %5 bitop3:0x48" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3));) }
		if (KIND == 3) { REP8(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %3, %3, %2\n v_mul_f64 %4, %4, %2" : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3), "v"(d3));) }
		if (KIND == 4) { REP8(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %3, %3, %2\n v_add_f64 %4, %4, %2" : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3), "v"(d3));) }
		if (KIND == 5) { REP8(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_and_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b));) }
		if (KIND == 6) { REP8(asm volatile("v_bitop3_b32 %0, s4, %3, %0 bitop3:0x48\n v_bitop3_b32 %1, s5, %2, %1 bitop3:0x48\n v_bitop3_b32 %4, s6, %3, %4 bitop3:0x48\n v_bitop3_b32 %5, s7, %2, %5 bitop3:0x48" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3) : "s4", "s5", "s6", "s7");) }
		if (KIND == 7) { REP8(asm volatile("v_mul_f64 %0, %0, s[4:5]\n v_mul_f64 %1, %1, s[6:7]\n v_mul_f64 %2, %2, s[4:5]\n v_mul_f64 %3, %3, s[6:7]" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : : "s4", "s5", "s6", "s7");) }
		if (KIND == 8) { REP8(asm volatile("v_mov_b64_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %4, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3), "v"(d3));) }
		if (KIND == 9) { REP8(asm volatile("v_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %3 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3));) }
		if (KIND == 10) { REP8(asm volatile("v_mul_u32_u24 %0, %2, %0\n v_mul_u32_u24 %1, %3, %1\n v_mul_u32_u24 %4, %2, %4\n v_mul_u32_u24 %5, %3, %5" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3));) }
		if (KIND == 11) { REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
		if (KIND == 12) { REP8(asm volatile("v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n v_mov_b64 %3, %2\n v_mov_b64 %4, %2" : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3), "v"(d3));) }
		if (KIND == 13) { REP8(asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));) }
		if (KIND == 14) { REP8(asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
		if (KIND == 15) { REP8(asm volatile("v_bfe_u32 %0, %2, 4, 4\n v_bfe_u32 %1, %3, 8, 4\n v_bfe_u32 %4, %2, 12, 4\n v_bfe_u32 %5, %3, 16, 4" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3));) }
		if (KIND == 16) { REP8(asm volatile("v_cndmask_b32 %0, %2, %0, vcc\n v_cndmask_b32 %1, %3, %1, vcc\n v_cndmask_b32 %4, %2, %4, vcc\n v_cndmask_b32 %5, %3, %5, vcc" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3) : "vcc");) }
		if (KIND == 17) { REP8(asm volatile("v_pk_add_u16 %0, %2, %0\n v_pk_add_u16 %1, %3, %1\n v_pk_add_u16 %4, %2, %4\n v_pk_add_u16 %5, %3, %5" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2), "v"(r3));) }
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + (unsigned)(d0 + d1 + d2 + d3) + (unsigned)(f0 + f1);
}

template <int KIND>
double run(const char *name, unsigned *d_out, double ref)
{
	const int blocks = 256 * 8, iters = 4000;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, 100, d_out);
	hipDeviceSynchronize();
	hipEventRecord(a);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, iters, d_out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	const double wave_instr = (double)blocks * 4 * iters * 32;       // 4 waves per block, 32 instr per iteration
	const double per_simd_per_us = wave_instr / 1024.0 / (ms * 1e3);
	printf("%-28s %8.3f ms  %8.1f wave-instr/SIMD/us%s", name, ms, per_simd_per_us, ref > 0 ? "" : "\n");
	if (ref > 0) printf("   = %.2fx the cost of v_fma_f32\n", ref / per_simd_per_us);
	return per_simd_per_us;
}

int main()
{
	unsigned *d_out;
	hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned));
	const double ref = run<0>("v_fma_f32", d_out, 0);
	run<1>("v_bcnt_u32_b32", d_out, ref);
	run<2>("v_bitop3_b32 (vgpr)", d_out, ref);
	run<6>("v_bitop3_b32 (sgpr src0)", d_out, ref);
	run<5>("v_lshl/and/xor", d_out, ref);
	run<3>("v_mul_f64", d_out, ref);
	run<7>("v_mul_f64 (sgpr src)", d_out, ref);
	run<4>("v_add_f64", d_out, ref);
	run<12>("v_mov_b64", d_out, ref);
	run<8>("v_mov_b64_dpp row_newbcast", d_out, ref);
	run<13>("v_fmac_f64_dpp row_newbcast", d_out, ref);
	run<9>("v_mov_b32_dpp row_newbcast", d_out, ref);
	run<10>("v_mul_u32_u24", d_out, ref);
	run<11>("v_permlane32_swap_b32", d_out, ref);
	run<14>("v_cvt_u32_f32", d_out, ref);
	run<15>("v_bfe_u32", d_out, ref);
	run<16>("v_cndmask_b32", d_out, ref);
	run<17>("v_pk_add_u16", d_out, ref);
	return 0;
}
