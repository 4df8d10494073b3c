// hibag_sample.hip -- the per-sample route of the HIBAG plugin table: predict_init / predict_avg_prob / predict_done of
// TypeGPUExtProc (inst/include/LibHLA_ext.h:358-388), which an unmodified HIBAG calls ONCE PER SAMPLE
// (src/LibHLA.cpp:2433-2441).  The batched kernels map lane = sample; driven with one sample they keep one lane of 64
// alive and let a single wavefront walk a whole classifier (0.74 ms per call, profiles/r03_bench.json).  For one sample
// the parallelism has to come from the model instead:
//
//   k_one_cells   thread = one non-empty allele-pair cell of one classifier: the cell's haplotype pairs in the
//                 reference's order, d = hamm_d on the packed words, sum += (2 f1 f2) * TAB[d]   (src/LibHLA.cpp:1776-1821)
//   k_one_total   wavefront = classifier: its cells added in posterior order by lane 0 -> total, 1/total      (:1823-1829)
//   k_one_accum   thread = posterior cell p: S[p] += (cell * (1/total)) * w over the classifiers in order, normalised by
//                 the sum of weights (:1497-1518, :2448-2480); thread 0 also forms the matching value
//
// Every sum is formed by one thread in the reference's order: results are bit-identical to the CPU kernels, like the
// batched route's.  ~60,000 cells, 505,000 pairs for the benchmark model: a few tens of microseconds of device time.

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hibag_device.h"
#include "hibag_plugin.h"

int hibag_selected_device();      // hibag_api.hip

namespace {

struct OneCell { int c, a0, a1, b0, b1, k, p, pad; };  // classifier, haplotype ranges (absolute indices) of the two alleles, SNPs of the classifier, posterior cell

struct OneView {
	int n_hla, n_cell, n_classifier, n_cells_total;
	const uint64_t *bits;        // [n_haplo_total][2] packed haplotypes, bits >= the classifier's SNP count cleared
	const double *freq;          // [n_haplo_total]
	const OneCell *cells;        // [n_cells_total] the non-empty cells, classifier after classifier, posterior order inside
	const int *cell_off;         // [C + 1] first cell of each classifier
	const int *hap_off;          // [C + 1] first haplotype of each classifier
	const double *tab;           // [257]
	// per call
	const uint64_t *geno;        // [C][6] TGenotype: S1[2], S2[2], 16 bytes of book-keeping
	const double *weight;        // [C]
	double *cellv;               // [n_cells_total] the cell sums, classifier after classifier (k_one_total adds them in order)
	double *dense;               // [C][P] the same sums by posterior cell; the structurally empty cells stay +0.0 for ever (k_one_accum)
	double *tot, *inv;           // [C]
	double *out;                 // [P + 1]: the averaged posterior, then the matching value
};

// hamm_d, src/LibHLA.cpp:747-819: per SNP |g - h1 - h2| for a called genotype, 0 for a missing one
__device__ __forceinline__ int hamm_d(uint64_t h1a, uint64_t h1b, uint64_t h2a, uint64_t h2b,
	uint64_t s1a, uint64_t s1b, uint64_t s2a, uint64_t s2b)
{
	const uint64_t ma = s2a & ~s1a, mb = s2b & ~s1b;                       // missing
	const uint64_t ka = ((h1a ^ s2a) | (h2a ^ s1a)) & ~ma, kb = ((h1b ^ s2b) | (h2b ^ s1b)) & ~mb;
	return __popcll((h1a ^ s1a) & ka) + __popcll((h2a ^ s2a) & ka) + __popcll((h1b ^ s1b) & kb) + __popcll((h2b ^ s2b) & kb);
}

// The haplotypes a workgroup's cells refer to (its 256 consecutive cells belong to one or two classifiers, a few more for
// tiny ones) are staged in LDS: a cell is summed by ONE thread, pair after pair, so the round trip of every haplotype
// look-up is on the critical path of the longest cell (378 pairs in the benchmark model) -- from L2 that made the kernel
// 70 us, most of a predict_avg_prob call.  More haplotypes than the staging area holds: straight from memory.
#define ONE_LDS_HAPLO 2048

template <bool STAGED>
__device__ __forceinline__ double one_cell(const OneCell &q, const uint64_t *bits, const double *freq, int base,
	uint64_t s1a, uint64_t s1b, uint64_t s2a, uint64_t s2b, const double *tab_s)
{
	const bool diagonal = q.a0 == q.b0;
	double cell = 0;
	for (int a = q.a0 - base; a < q.a1 - base; a++) {
		const uint64_t h1a = bits[2 * (size_t)a], h1b = bits[2 * (size_t)a + 1];
		const double fa = freq[a];
		int b = q.b0 - base;
		const int b1 = q.b1 - base;
		if (diagonal) {                                                     // :1786 -- the pair (a, a) with f * f first
			cell += (fa * fa) * tab_s[hamm_d(h1a, h1b, h1a, h1b, s1a, s1b, s2a, s2b)];
			b = a + 1;
		}
		const double ff = 2 * fa;                                           // :1789-1793, :1808-1812
		// eight pairs at a time -- look-ups, distances, table values in flight together, the last batch of a row with
		// clamped indices -- and only the additions in order
		for (; b < b1; b += 8) {
			uint64_t ha[8], hb[8];
			double fb[8], t[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const size_t i = (size_t)(b + j < b1 ? b + j : b1 - 1);
				ha[j] = bits[2 * i]; hb[j] = bits[2 * i + 1]; fb[j] = freq[i];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) t[j] = tab_s[hamm_d(h1a, h1b, ha[j], hb[j], s1a, s1b, s2a, s2b)];
#pragma unroll
			for (int j = 0; j < 8; j++)
				if (b + j < b1) cell += (ff * fb[j]) * t[j];
		}
	}
	return cell;
}

// the same sums for a classifier of at most 32 SNPs -- nearly all of them: one 32-bit word per haplotype, a fifth of the
// instructions (the cell's thread is alone in its wavefront for most of a long cell: instruction count is its time)
__device__ __forceinline__ int hamm_d32(uint32_t h1, uint32_t h2, uint32_t s1, uint32_t s2, uint32_t notm)
{
	const uint32_t k = ((h1 ^ s2) | (h2 ^ s1)) & notm;
	return __popc((h1 ^ s1) & k) + __popc((h2 ^ s2) & k);
}

__device__ __forceinline__ double one_cell32(const OneCell &q, const uint64_t *bits, const double *freq, int base,
	uint32_t s1, uint32_t s2, const double *tab_s)
{
	const uint32_t notm = ~(s2 & ~s1);
	const bool diagonal = q.a0 == q.b0;
	double cell = 0;
	for (int a = q.a0 - base; a < q.a1 - base; a++) {
		const uint32_t h1 = (uint32_t)bits[2 * (size_t)a];
		const double fa = freq[a];
		int b = q.b0 - base;
		const int b1 = q.b1 - base;
		if (diagonal) {
			cell += (fa * fa) * tab_s[hamm_d32(h1, h1, s1, s2, notm)];
			b = a + 1;
		}
		const double ff = 2 * fa;
		for (; b < b1; b += 8) {
			uint32_t hb[8];
			double fb[8], t[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const size_t i = (size_t)(b + j < b1 ? b + j : b1 - 1);
				hb[j] = (uint32_t)bits[2 * i]; fb[j] = freq[i];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) t[j] = tab_s[hamm_d32(h1, hb[j], s1, s2, notm)];
#pragma unroll
			for (int j = 0; j < 8; j++)
				if (b + j < b1) cell += (ff * fb[j]) * t[j];
		}
	}
	return cell;
}

__global__ __launch_bounds__(256) void k_one_cells(OneView V)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	__shared__ uint64_t bits_s[2 * ONE_LDS_HAPLO];
	__shared__ double freq_s[ONE_LDS_HAPLO];
	for (int i = threadIdx.x; i < HIBAG_TAB_N; i += blockDim.x) tab_s[i] = V.tab[i];
	// the classifiers of this workgroup's cells and their haplotypes
	const int t0 = blockIdx.x * blockDim.x, t1 = min(t0 + (int)blockDim.x, V.n_cells_total) - 1;
	const int c_lo = V.cells[t0].c, c_hi = V.cells[t1].c;
	const int base = V.hap_off[c_lo], n_h = V.hap_off[c_hi + 1] - base;
	const bool staged = n_h <= ONE_LDS_HAPLO;
	if (staged)
		for (int i = threadIdx.x; i < n_h; i += blockDim.x) {
			bits_s[2 * i] = V.bits[2 * (size_t)(base + i)]; bits_s[2 * i + 1] = V.bits[2 * (size_t)(base + i) + 1];
			freq_s[i] = V.freq[base + i];
		}
	__syncthreads();
	const int t = t0 + threadIdx.x;
	if (t >= V.n_cells_total) return;
	const OneCell q = V.cells[t];
	if (!(V.weight[q.c] > 0)) return;                                      // the classifier is skipped (src/LibHLA.cpp:2451)
	const uint64_t *g = V.geno + (size_t)q.c * 6;
	// positions >= the classifier's SNP count are missing (S1 = 0, S2 = 1) whatever the host left there (TGenotype::IntToSNP
	// pre-fills them so, src/LibHLA.cpp:672-673)
	const uint64_t ka = q.k >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << q.k) - 1), kb = q.k >= 128 ? ~(uint64_t)0 : (q.k <= 64 ? 0 : (((uint64_t)1 << (q.k - 64)) - 1));
	const uint64_t s1a = g[0] & ka, s1b = g[1] & kb, s2a = g[2] | ~ka, s2b = g[3] | ~kb;
	double v;
	if (q.k <= 32)
		v = staged ? one_cell32(q, bits_s, freq_s, base, (uint32_t)s1a, (uint32_t)s2a, tab_s)
		           : one_cell32(q, V.bits, V.freq, 0, (uint32_t)s1a, (uint32_t)s2a, tab_s);
	else
		v = staged ? one_cell<true>(q, bits_s, freq_s, base, s1a, s1b, s2a, s2b, tab_s)
		           : one_cell<false>(q, V.bits, V.freq, 0, s1a, s1b, s2a, s2b, tab_s);
	V.cellv[t] = v;
	V.dense[(size_t)q.c * V.n_cell + q.p] = v;
}

// One wavefront per classifier: the lanes fetch its cell sums 1,024 at a time into LDS (coalesced), lane 0 adds them in cell
// order from there, sixteen values per LDS round trip -- a single thread reading memory directly took 16 us for the ~600 cells
// of a classifier of the benchmark model, most of it the latency of ten rounds of loads.
#define ONE_TOTAL_CHUNK 1024
__global__ __launch_bounds__(64) void k_one_total(OneView V)
{
	__shared__ double v_s[ONE_TOTAL_CHUNK];
	const int c = blockIdx.x;
	if (!(V.weight[c] > 0)) { if (threadIdx.x == 0) { V.tot[c] = 0; V.inv[c] = 0; } return; }
	const int i0 = V.cell_off[c], i1 = V.cell_off[c + 1];
	double total = 0;
	for (int base = i0; base < i1; base += ONE_TOTAL_CHUNK) {
		const int n = min(ONE_TOTAL_CHUNK, i1 - base);
		for (int i = threadIdx.x; i < ONE_TOTAL_CHUNK; i += 64) v_s[i] = i < n ? V.cellv[base + i] : 0.0;
		__syncthreads();
		if (threadIdx.x == 0) {
			int i = 0;
			for (; i + 16 <= n; i += 16) {
				double v[16];
#pragma unroll
				for (int j = 0; j < 16; j++) v[j] = v_s[i + j];
#pragma unroll
				for (int j = 0; j < 16; j++) total += v[j];
			}
			for (; i < n; i++) total += v_s[i];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		V.tot[c] = total;
		V.inv[c] = 1 / total;                             // src/LibHLA.cpp:1827 (inf when total == 0)
	}
}

__global__ __launch_bounds__(256) void k_one_accum(OneView V)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p > V.n_cell) return;
	const int C = V.n_classifier;
	if (p == V.n_cell) {                                  // one extra thread: the matching value (:2458-2459, :2480)
		double sum_m = 0, num_m = 0;
		for (int c = 0; c < C; c++) {
			const double w = V.weight[c];
			if (!(w > 0)) continue;
			sum_m += V.tot[c] * w;
			num_m += w;
		}
		V.out[V.n_cell] = sum_m / num_m;
		return;
	}
	// Thirty-two classifiers' values in flight (coalesced: consecutive threads, consecutive cells), added in classifier order.
	// A structurally empty cell reads +0.0 and adds (0 * inv) * w: +0.0 -- nothing -- for a finite 1/total, and NaN where the
	// total is 0 or so small that 1/total overflows, which is what the reference's loop produces there (0 * inf).
	double S = 0, sum_w = 0;
	constexpr int NB = 32;
	for (int c0 = 0; c0 < C; c0 += NB) {
		double w[NB], inv[NB], v[NB];
#pragma unroll
		for (int j = 0; j < NB; j++) {
			const int c = c0 + j < C ? c0 + j : C - 1;
			w[j] = c0 + j < C ? V.weight[c] : 0.0;
			inv[j] = V.inv[c];
			v[j] = V.dense[(size_t)c * V.n_cell + p];
		}
#pragma unroll
		for (int j = 0; j < NB; j++) {
			if (!(w[j] > 0)) continue;                    // AddProbToSum skips the classifier (:1497-1507)
			sum_w += w[j];
			S += (v[j] * inv[j]) * w[j];
		}
	}
	V.out[p] = sum_w > 0 ? S * (1.0 / sum_w) : S;         // NormalizeSumPostProb (:1509-1518)
}

struct OneState {
	bool active = false;
	int device = 0;
	OneView V{};
	int P = 0, C = 0;
	void *d_model = nullptr, *d_call = nullptr;           // one arena each
	void *h_pin = nullptr;                                // pinned staging: genotypes + weights in, posterior + matching out
	size_t call_in = 0, call_bytes = 0;
};
OneState g1;
thread_local char g1_msg[400];

[[noreturn]] void one_throw(const char *what, hipError_t e = hipSuccess)
{
	if (e != hipSuccess) snprintf(g1_msg, sizeof(g1_msg), "HIBAG HIP plugin: %s: %s", what, hipGetErrorString(e));
	else snprintf(g1_msg, sizeof(g1_msg), "HIBAG HIP plugin: %s", what);
	throw (const char *)g1_msg;
}
#define ONE_OK(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) one_throw(what, e_); } while (0)

} // namespace

// predict_done(): from a destructor (src/LibHLA.cpp:2312-2315, :2525-2531) -- must not throw
void hibag_sample_done()
{
	if (g1.d_model) (void)hipFree(g1.d_model);
	if (g1.d_call) (void)hipFree(g1.d_call);
	if (g1.h_pin) (void)hipHostFree(g1.h_pin);
	g1 = OneState();
}

// predict_init(nHLA, nClassifier, pHaplo[], nHaplo[], nSNP[]): src/LibHLA.cpp:2498-2523.  The lists are host-owned and only
// valid during the call: everything is copied.  aux.hla_allele was filled by SetHaploAux_GPU (:565-578).
void hibag_sample_init(int n_hla, int n_classifier, const PluginHaplotype *const p_haplo[], const int n_haplo[], const int n_snp[])
{
	hibag_sample_done();
	if (n_hla <= 0 || n_hla > 46340 || n_classifier < 0) one_throw("predict_init: invalid dimensions");
	ONE_OK(hipSetDevice(hibag_selected_device()), "hipSetDevice");
	const int C = n_classifier, P = n_hla * (n_hla + 1) / 2;
	std::vector<uint64_t> bits;
	std::vector<double> freq;
	std::vector<OneCell> cells;
	std::vector<int> cell_off(C + 1, 0), hap_off(C + 1, 0);
	for (int c = 0; c < C; c++) {
		const int H = n_haplo[c], k = n_snp[c];
		if (H < 0 || k < 0 || k > 128) one_throw("predict_init: invalid classifier");
		const int base = (int)freq.size();
		hap_off[c] = base;
		std::vector<int> st(n_hla + 1, 0);
		uint64_t mask[2];
		for (int w = 0; w < 2; w++) mask[w] = k >= 64 * w + 64 ? ~(uint64_t)0 : (k <= 64 * w ? 0 : (((uint64_t)1 << (k - 64 * w)) - 1));
		for (int i = 0; i < H; i++) {
			const PluginHaplotype &h = p_haplo[c][i];
			const int a = h.aux.hla_allele;
			if (a < 0 || a >= n_hla || (i > 0 && a < p_haplo[c][i - 1].aux.hla_allele)) one_throw("predict_init: haplotypes must be grouped by allele");
			st[a + 1]++;
			// bits >= nSNP are uninitialised in the reference (src/LibHLA.cpp:287-292) and harmless there because the genotype
			// marks them missing; cleared here all the same
			bits.push_back((uint64_t)h.packed[0] & mask[0]); bits.push_back((uint64_t)h.packed[1] & mask[1]);
			freq.push_back(h.freq);
		}
		for (int a = 0; a < n_hla; a++) st[a + 1] += st[a];
		cell_off[c] = (int)cells.size();
		int p = 0;
		for (int h1 = 0; h1 < n_hla; h1++)
			for (int h2 = h1; h2 < n_hla; h2++, p++) {
				if (st[h1] == st[h1 + 1] || st[h2] == st[h2 + 1]) continue;
				cells.push_back(OneCell{c, base + st[h1], base + st[h1 + 1], base + st[h2], base + st[h2 + 1], k, p, 0});
			}
	}
	cell_off[C] = (int)cells.size();
	hap_off[C] = (int)freq.size();
	double tab[HIBAG_TAB_N];
	for (int i = 0; i < HIBAG_TAB_N; i++) tab[i] = std::exp(i * std::log(1e-5));     // src/LibHLA.cpp:166-183
	tab[0] = 1;
	for (int i = 0; i < HIBAG_TAB_N; i++) if (!std::isfinite(tab[i])) tab[i] = 0;

	// model arena
	size_t o = 0;
	auto take = [&](size_t bytes) { const size_t at = o; o = (o + std::max<size_t>(bytes, 8) + 63) & ~(size_t)63; return at; };
	const size_t o_bits = take(bits.size() * 8), o_freq = take(freq.size() * 8), o_cells = take(cells.size() * sizeof(OneCell)),
		o_off = take(cell_off.size() * 4), o_hoff = take(hap_off.size() * 4), o_tab = take(sizeof(tab));
	ONE_OK(hipMalloc(&g1.d_model, o), "hipMalloc(model)");
	char *d = (char *)g1.d_model;
	auto put = [&](size_t at, const void *src, size_t bytes) { if (bytes) ONE_OK(hipMemcpy(d + at, src, bytes, hipMemcpyHostToDevice), "copy model"); };
	put(o_bits, bits.data(), bits.size() * 8); put(o_freq, freq.data(), freq.size() * 8);
	put(o_cells, cells.data(), cells.size() * sizeof(OneCell)); put(o_off, cell_off.data(), cell_off.size() * 4);
	put(o_hoff, hap_off.data(), hap_off.size() * 4);
	put(o_tab, tab, sizeof(tab));
	// per-call arena: [genotypes | weights] in, [posterior, matching] out, then scratch
	size_t oc = 0;
	auto takec = [&](size_t bytes) { const size_t at = oc; oc = (oc + std::max<size_t>(bytes, 8) + 63) & ~(size_t)63; return at; };
	const size_t c_geno = takec((size_t)C * sizeof(PluginGenotype)), c_w = takec((size_t)C * 8);
	g1.call_in = oc;
	const size_t c_out = takec((size_t)(P + 1) * 8), c_cellv = takec(cells.size() * 8), c_tot = takec((size_t)C * 8), c_inv = takec((size_t)C * 8),
		c_dense = takec((size_t)C * P * 8);            // (zeroed once, below: the empty cells are never written)
	g1.call_bytes = oc;
	ONE_OK(hipMalloc(&g1.d_call, oc), "hipMalloc(call)");
	ONE_OK(hipMemset(g1.d_call, 0, oc), "hipMemset(call)");
	ONE_OK(hipHostMalloc(&g1.h_pin, g1.call_in + (size_t)(P + 1) * 8 + 64, hipHostMallocDefault), "hipHostMalloc");
	char *dc = (char *)g1.d_call;
	OneView &V = g1.V;
	V.n_hla = n_hla; V.n_cell = P; V.n_classifier = C; V.n_cells_total = (int)cells.size();
	V.bits = (const uint64_t *)(d + o_bits); V.freq = (const double *)(d + o_freq); V.cells = (const OneCell *)(d + o_cells);
	V.cell_off = (const int *)(d + o_off); V.hap_off = (const int *)(d + o_hoff); V.tab = (const double *)(d + o_tab);
	V.geno = (const uint64_t *)(dc + c_geno); V.weight = (const double *)(dc + c_w);
	V.out = (double *)(dc + c_out); V.cellv = (double *)(dc + c_cellv); V.tot = (double *)(dc + c_tot); V.inv = (double *)(dc + c_inv);
	V.dense = (double *)(dc + c_dense);
	g1.P = P; g1.C = C; g1.device = hibag_selected_device();
	g1.active = true;
}

// predict_avg_prob(geno[nClassifier], weight[nClassifier], out_prob[P], out_match[1]): src/LibHLA.cpp:2433-2441
void hibag_sample_avg_prob(const PluginGenotype geno[], const double weight[], double out_prob[], double out_match[])
{
	if (!g1.active) one_throw("predict_avg_prob: predict_init was not called");
	ONE_OK(hipSetDevice(g1.device), "hipSetDevice");
	const int C = g1.C, P = g1.P;
	char *h = (char *)g1.h_pin;
	const size_t w_at = ((size_t)C * sizeof(PluginGenotype) + 63) & ~(size_t)63;     // (the arena's layout: genotypes, then weights)
	memcpy(h, geno, (size_t)C * sizeof(PluginGenotype));
	memcpy(h + w_at, weight, (size_t)C * 8);
	hipStream_t st = 0;
	if (C > 0) ONE_OK(hipMemcpyAsync(g1.d_call, h, g1.call_in, hipMemcpyHostToDevice, st), "copy genotypes");
	const OneView &V = g1.V;
	if (V.n_cells_total > 0) hipLaunchKernelGGL(k_one_cells, dim3((V.n_cells_total + 255) / 256), dim3(256), 0, st, V);
	if (C > 0) hipLaunchKernelGGL(k_one_total, dim3(C), dim3(64), 0, st, V);
	hipLaunchKernelGGL(k_one_accum, dim3((P + 1 + 255) / 256), dim3(256), 0, st, V);
	ONE_OK(hipGetLastError(), "launch");
	double *h_out = (double *)(h + g1.call_in);
	ONE_OK(hipMemcpyAsync(h_out, V.out, (size_t)(P + 1) * 8, hipMemcpyDeviceToHost, st), "read posterior");
	ONE_OK(hipStreamSynchronize(st), "predict_avg_prob");
	memcpy(out_prob, h_out, (size_t)P * 8);
	out_match[0] = h_out[P];
}
