// Micro-benchmark: throughput of the data-movement instructions the accumulation
// phase is made of (gfx950): per-lane LDS gathers of the mutation table,
// wave-uniform (broadcast) LDS reads of the frequency factors, v_readlane and
// L1-hit global loads as alternatives.  All CUs loaded with 8 waves per SIMD.
// Prints cycles per wave-instruction per CU (at the measured shader clock
// implied by v_fma_f32 = 4 cycles... reported raw as ns too).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, const double *gtab, unsigned *out)
{
	__shared__ double tab[512];
	__shared__ double rows[4][8][64];                                   // (KIND 10..12: accumulator rows, conflict-free: lane * 8 bytes)
	for (int i = threadIdx.x; i < 512; i += 256) tab[i] = gtab[i & 63] + i;
	__syncthreads();
	const unsigned lane = threadIdx.x & 63;
	// addresses (bytes)
	unsigned a_gather = ((lane * 7 + (lane >> 3)) % 61) * 8;            // 61 distinct doubles, like TAB[d]
	unsigned a_rand = ((lane * 37 + 11) & 255) * 8;                     // 2 KB table, like the A expansion
	unsigned a_bcast = 64;
	unsigned a_b128 = lane * 16;
	const char *gp = (const char *)gtab;
	double acc = 0;
	unsigned u = lane, v = lane * 3;
	for (int i = 0; i < iters; i++) {
		double t0, t1, t2, t3, t4, t5, t6, t7;
		if (KIND == 0 || KIND == 1 || KIND == 4) {
			const unsigned a = KIND == 0 ? a_gather : (KIND == 1 ? a_bcast : a_rand);
			asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
				"ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)"
				: "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(a));
			acc += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
		}
		if (KIND == 2 || KIND == 3) {
			typedef double d2 __attribute__((ext_vector_type(2)));
			d2 p0, p1, p2, p3, p4, p5, p6, p7;
			const unsigned a = KIND == 2 ? a_bcast : a_b128;
			asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
				"ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
				: "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7) : "v"(a));
			acc += p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
		}
		if (KIND == 5) {     // 8 x (two v_readlane + use as 64-bit SGPR operand)
			asm volatile(
				"v_readlane_b32 s4, %1, 0\n v_readlane_b32 s5, %2, 0\n v_readlane_b32 s6, %1, 1\n v_readlane_b32 s7, %2, 1\n"
				"v_readlane_b32 s8, %1, 2\n v_readlane_b32 s9, %2, 2\n v_readlane_b32 s10, %1, 3\n v_readlane_b32 s11, %2, 3\n"
				"v_readlane_b32 s4, %1, 4\n v_readlane_b32 s5, %2, 4\n v_readlane_b32 s6, %1, 5\n v_readlane_b32 s7, %2, 5\n"
				"v_readlane_b32 s8, %1, 6\n v_readlane_b32 s9, %2, 6\n v_readlane_b32 s10, %1, 7\n v_readlane_b32 s11, %2, 7\n"
				"s_nop 0\n v_add_u32 %0, s4, %0\n v_add_u32 %0, s9, %0"
				: "+v"(u) : "v"(v), "v"(lane) : "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11");
		}
		if (KIND == 12) {    // 8 x v_permlane32_swap (what gives a lane its own sample's column of a 32x32 matrix result)
			asm volatile(
				"v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
				"v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1"
				: "+v"(u), "+v"(v));
		}
		if (KIND == 13) {    // 8 x v_mul_f64 / v_add_f64 on independent registers
			asm volatile(
				"v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n"
				"v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %2"
				: "+v"(acc), "+v"(t0) : "v"(1.0000001));
		}
		if (KIND == 6 || KIND == 7) {
			const char *p = gp + (KIND == 6 ? a_bcast : a_gather);
			asm volatile("global_load_dwordx2 %0, %8, off\n global_load_dwordx2 %1, %8, off offset:8\n global_load_dwordx2 %2, %8, off offset:16\n global_load_dwordx2 %3, %8, off offset:24\n"
				"global_load_dwordx2 %4, %8, off offset:32\n global_load_dwordx2 %5, %8, off offset:40\n global_load_dwordx2 %6, %8, off offset:48\n global_load_dwordx2 %7, %8, off offset:56\n s_waitcnt vmcnt(0)"
				: "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(p));
			acc += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
		}
		if (KIND == 8) {
			unsigned x0, x1, x2, x3, x4, x5, x6, x7;
			const unsigned a = a_gather >> 1;   // byte address = 4 * source lane
			asm volatile("ds_bpermute_b32 %0, %8, %9\n ds_bpermute_b32 %1, %8, %9\n ds_bpermute_b32 %2, %8, %9\n ds_bpermute_b32 %3, %8, %9\n"
				"ds_bpermute_b32 %4, %8, %9\n ds_bpermute_b32 %5, %8, %9\n ds_bpermute_b32 %6, %8, %9\n ds_bpermute_b32 %7, %8, %9\n s_waitcnt lgkmcnt(0)"
				: "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(a), "v"(v));
			u += x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
		}
		if (KIND == 10 || KIND == 11) {   // 8 x ds_add_f64 (no return) / ds_write_b64 into eight 512-byte rows of this wavefront
			const unsigned a = (unsigned)(size_t)&rows[threadIdx.x >> 6][0][lane];
			const double x = 1.0 + lane;
			if (KIND == 10)
				asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:512\n ds_add_f64 %0, %1 offset:1024\n ds_add_f64 %0, %1 offset:1536\n"
					"ds_add_f64 %0, %1 offset:2048\n ds_add_f64 %0, %1 offset:2560\n ds_add_f64 %0, %1 offset:3072\n ds_add_f64 %0, %1 offset:3584\n s_waitcnt lgkmcnt(0)"
					:: "v"(a), "v"(x) : "memory");
			else
				asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n"
					"ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n s_waitcnt lgkmcnt(0)"
					:: "v"(a), "v"(x) : "memory");
		}
		if (KIND == 9) {     // reference: 8 x v_fma_f64 (FP64 FMA = the VALU's full-rate FP64 op)
			asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n"
				"v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1" : "+v"(acc) : "v"(1.0000001));
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = u + (unsigned)acc;
}

template <int KIND>
void run(const char *name, const double *gtab, unsigned *d_out, int per_iter)
{
	const int blocks = 256 * 8, iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, 50, gtab, d_out);
	hipDeviceSynchronize();
	hipEventRecord(a);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, iters, gtab, d_out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	const double wave_instr_per_cu = (double)blocks * 4 * iters * per_iter / 256.0;
	const double ns = ms * 1e6 / wave_instr_per_cu;
	printf("%-44s %8.3f ms   %6.2f ns per wave-instr per CU  = %5.1f clk @2.4GHz\n", name, ms, ns, ns * 2.4);
}

int main()
{
	unsigned *d_out;
	double *gtab;
	hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned));
	hipMalloc(&gtab, 4096);
	hipMemset(gtab, 0, 4096);
	run<9>("v_fma_f64 (per CU: 4 SIMDs)", gtab, d_out, 8);
	run<0>("ds_read_b64 gather, 61 doubles (TAB[d])", gtab, d_out, 8);
	run<1>("ds_read_b64 broadcast", gtab, d_out, 8);
	run<4>("ds_read_b64 gather, 2 KB table (A expand)", gtab, d_out, 8);
	run<2>("ds_read_b128 broadcast (factors)", gtab, d_out, 8);
	run<3>("ds_read_b128 lane*16", gtab, d_out, 8);
	run<8>("ds_bpermute_b32", gtab, d_out, 8);
	run<10>("ds_add_f64 lane*8 (accumulator rows)", gtab, d_out, 8);
	run<11>("ds_write_b64 lane*8", gtab, d_out, 8);
	run<5>("v_readlane_b32 (x2 = one factor)", gtab, d_out, 16);
	run<12>("v_permlane32_swap_b32", gtab, d_out, 8);
	run<13>("v_mul_f64 / v_add_f64", gtab, d_out, 8);
	run<6>("global_load_dwordx2 broadcast (L1 hit)", gtab, d_out, 8);
	run<7>("global_load_dwordx2 gather 61 doubles (L1)", gtab, d_out, 8);
	return 0;
}
