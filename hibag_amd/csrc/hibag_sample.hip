// hibag_sample.hip -- the per-sample route of the HIBAG plugin table: predict_init / predict_avg_prob / predict_done of
// TypeGPUExtProc (inst/include/LibHLA_ext.h:358-388), which an unmodified HIBAG calls ONCE PER SAMPLE
// (src/LibHLA.cpp:2433-2441).  The batched kernels map lane = sample; driven with one sample they keep one lane of 64
// alive and let a single wavefront walk a whole classifier (0.74 ms per call).  For one sample the parallelism has to come
// from the model, and a call is latency: what counts is the number of launches, copies and dependent steps.  So a call is
// ONE kernel, k_one, workgroup = classifier (1,024 threads):
//
//   pairs    thread = haplotype pair of the classifier's flat pair list (host-built at predict_init: index pair and the
//            rounded factor (2 f1) f2 / f1 f1 of src/LibHLA.cpp:1786-1813): d = hamm_d on the packed words (:747-819),
//            x = factor * TAB[d] into LDS, 4,096 pairs per round
//   cells    thread = allele-pair cell: its x values added IN ORDER from LDS (:1776-1821) -- a 378-pair cell is 378
//            dependent additions by one lane, not 378 dependent distance computations as in round 3
//   total    lane 0: the round's completed cells added in cell order -> total, 1 / total (:1823-1829)
//   ensemble the LAST workgroup to finish (a device counter behind a release fence; nobody spins) does
//            S[p] += (cell * (1 / total)) * w over the classifiers in order, thread = posterior cell, normalises by the sum of
//            weights (:1497-1518, :2448-2480) and forms the matching value
//
// Genotypes and weights are read from, and the posterior is written to, host-mapped pinned memory: no copy commands.  The
// host learns of the end of a call from a sequence number the last workgroup stores behind a system-scope fence, which it
// polls -- a few microseconds sooner than the runtime's completion signal.  Every sum is formed by one thread in the
// reference's order: results are bit-identical to the CPU kernels (tests/test_hip_parity.py, every width and edge).

#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <immintrin.h>

#include "hibag_device.h"
#include "hibag_plugin.h"

int hibag_selected_device();      // hibag_api.hip

namespace {

#define ONE_THREADS 1024
#define ONE_ROUND 8192              // pairs per round (their products in LDS): one round for a classifier of up to 127 haplotypes
#define ONE_CELLS 4096              // cells a round may complete (their sums in LDS for the in-order total); more: a shorter round
#define ONE_LDS_HAPLO 2048          // haplotypes staged in LDS per classifier (more: straight from memory)
#define ONE_CHUNK 512               // classifiers whose weights / reciprocals / totals the ensemble phase holds in LDS at a time

struct OneCls {                     // one classifier of the model
	long long pair0;                // its first pair in OneView::pair / fac
	int n_pair, cell0, n_cell;      // cells: entries of cell_off (n_cell + 1 of them, classifier-local pair offsets) and cell_p from cell0 on
	int hap0, n_hap, k, round0;     // haplotypes from hap0 on; SNPs; its rounds (entries of OneView::round: n_round + 1 of them) from round0 on
	int n_round, r0_hi, r0_cell1, pad;   // the first round's end -- pair and cell -- here, so that it starts without another dependent load
};

struct OneRound { int pair0, cell0; };   // a round begins at this pair (classifier-local) inside this cell

struct OneView {
	int n_hla, n_cell, n_classifier;
	int stamps_on;                  // diagnostic (HIBAG_ONE_STAMPS): workgroup 0 stamps its phases
	int n_group;                    // workgroups of a launch: min(classifiers, what the device holds at once) -- they meet at a barrier
	unsigned spin_limit;            // polls at that barrier before a workgroup gives up and reports the call as not done
	const OneCls *cls;              // [C]
	const uint64_t *bits;           // [n_haplo_total][2] packed haplotypes, bits >= the classifier's SNP count cleared
	const uint32_t *pair;           // per pair: first | second << 16 (classifier-local haplotype indices)
	const double *fac;              // per pair: (2 f1) f2, or f1 f1 for the leading pair (i, i) of a diagonal cell -- rounded as the reference rounds it
	const int *cell_off;            // per classifier n_cell + 1 pair offsets of its non-empty cells (posterior order)
	const int *cell_p;              // posterior index of each of them
	const OneRound *round;          // per classifier its rounds: at most ONE_ROUND pairs and ONE_CELLS completed cells each
	const double *tab;              // [257]
	// per call, host-mapped
	const uint64_t *geno;           // [C][6] TGenotype: S1[2], S2[2], 16 bytes of book-keeping
	const double *weight;           // [C]
	double *out;                    // [P + 1]: the averaged posterior, then the matching value
	volatile uint32_t *flag;        // sequence number of the last finished call (0x80000000 | it: the barrier timed out)
	// per call, device
	double *dense;                  // [C][P] cell sums by posterior cell; the structurally empty cells stay +0.0 for ever
	double *tot, *inv, *wdev;       // [C]
	unsigned long long *arrived;    // workgroups that have reached the barrier / have written their outputs, over all calls so far
	unsigned long long *finished;
	unsigned long long *stamps;     // host-mapped, 16 entries: wall-clock stamps (100 MHz) of workgroup 0 (HIBAG_ONE_STAMPS=1 prints them)
};

// hamm_d, src/LibHLA.cpp:747-819: per SNP |g - h1 - h2| for a called genotype, 0 for a missing one
__device__ __forceinline__ int hamm_d(uint64_t h1a, uint64_t h1b, uint64_t h2a, uint64_t h2b,
	uint64_t s1a, uint64_t s1b, uint64_t s2a, uint64_t s2b)
{
	const uint64_t ma = s2a & ~s1a, mb = s2b & ~s1b;                       // missing
	const uint64_t ka = ((h1a ^ s2a) | (h2a ^ s1a)) & ~ma, kb = ((h1b ^ s2b) | (h2b ^ s1b)) & ~mb;
	return __popcll((h1a ^ s1a) & ka) + __popcll((h2a ^ s2a) & ka) + __popcll((h1b ^ s1b) & kb) + __popcll((h2b ^ s2b) & kb);
}

// the same for a classifier of at most 32 SNPs -- nearly all of them: one 32-bit word per haplotype
__device__ __forceinline__ int hamm_d32(uint32_t h1, uint32_t h2, uint32_t s1, uint32_t s2, uint32_t notm)
{
	const uint32_t k = ((h1 ^ s2) | (h2 ^ s1)) & notm;
	return __popc((h1 ^ s1) & k) + __popc((h2 ^ s2) & k);
}

// n values of v added to s in order, sixteen LDS reads in flight
__device__ __forceinline__ double add_in_order(double s, const double *v, int n)
{
	int i = 0;
	for (; i + 16 <= n; i += 16) {
		double t[16];
#pragma unroll
		for (int j = 0; j < 16; j++) t[j] = v[i + j];
#pragma unroll
		for (int j = 0; j < 16; j++) s += t[j];
	}
	for (; i < n; i++) s += v[i];
	return s;
}

// What one workgroup leaves for the others goes PAST its XCD's L2 (device-scope stores), so that announcing it needs no
// cache write-back -- a release fence costs 12 us here, a third of a call -- only the stores' completion; the readers fetch
// it past their own L2 likewise.  (System scope for what the host reads.)
__device__ __forceinline__ void put_dev(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double get_dev(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void put_host(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#define STAMP(i) do { if (V.stamps_on && blockIdx.x == 0 && threadIdx.x == 0) V.stamps[i] = wall_clock64(); } while (0)

// `call`: 1, 2, 3 ... -- the barrier counters only ever grow (call * n_group arrivals end call number `call`), nothing is reset
__global__ __launch_bounds__(ONE_THREADS) void k_one(OneView V, uint32_t seq, unsigned long long call)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	__shared__ double xs[ONE_ROUND];                  // the round's products factor * TAB[d], pair order; ensemble phase: cell sums
	__shared__ double cs[ONE_CELLS];                  // the sums of the cells the round completed, cell order
	__shared__ uint64_t bits_s[2 * ONE_LDS_HAPLO];
	__shared__ double carry_s[2], total_s;
	__shared__ int ok_s;
	const int tid = threadIdx.x, C = V.n_classifier, P = V.n_cell, G = V.n_group;
	STAMP(0);
	for (int i = tid; i < HIBAG_TAB_N; i += ONE_THREADS) tab_s[i] = V.tab[i];
	for (int c = blockIdx.x; c < C; c += G) {
		// (the host-mapped reads -- a PCIe round trip -- are requested first and looked at last)
		const double w = V.weight[c];
		const uint64_t *g = V.geno + (size_t)c * 6;
		const uint64_t g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
		const OneCls K = V.cls[c];
		const bool staged = K.n_hap <= ONE_LDS_HAPLO;
		__syncthreads();                                              // (the staging areas are free again)
		if (staged)
			for (int i = tid; i < 2 * K.n_hap; i += ONE_THREADS) bits_s[i] = V.bits[2 * (size_t)K.hap0 + i];
		if (tid == 0) { carry_s[0] = carry_s[1] = 0; total_s = 0; }
		__syncthreads();
		if (c == 0) STAMP(1);
		if (w > 0 && K.n_pair > 0) {
			// positions >= the classifier's SNP count are missing (S1 = 0, S2 = 1) whatever the host left there (TGenotype::IntToSNP
			// pre-fills them so, src/LibHLA.cpp:672-673)
			const int k = K.k;
			const uint64_t ka = k >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << k) - 1), kb = k >= 128 ? ~(uint64_t)0 : (k <= 64 ? 0 : (((uint64_t)1 << (k - 64)) - 1));
			const uint64_t s1a = g0 & ka, s1b = g1 & kb, s2a = g2 | ~ka, s2b = g3 | ~kb;
			const uint32_t s1 = (uint32_t)s1a, s2 = (uint32_t)s2a, notm = ~(s2 & ~s1);
			const uint64_t *hb = staged ? bits_s : V.bits + 2 * (size_t)K.hap0;
			const int *off = V.cell_off + K.cell0 + c;                // (n_cell + 1 entries per classifier)
			const int *cp = V.cell_p + K.cell0;
			const OneRound *rd = V.round + K.round0;
			double *dense = V.dense + (size_t)c * P;
			for (int r = 0; r < K.n_round; r++) {
				const int lo = r ? rd[r].pair0 : 0, hi = r ? rd[r + 1].pair0 : K.r0_hi, n = hi - lo, j0 = r ? rd[r].cell0 : 0;
				const int j1 = r ? rd[r + 1].cell0 : K.r0_cell1;         // the cell that holds the next round's first pair
				// the bounds of this thread's first cell of the round: requested now, needed behind the pairs
				const int jf = j0 + tid;
				int o0f = hi, o1f = hi, pjf = 0;
				if (jf < K.n_cell) { o0f = off[jf]; o1f = off[jf + 1]; pjf = cp[jf]; }
				// ---- pairs: x = factor * TAB[d]; eight pairs' loads in flight per thread (a whole round in one memory round trip)
				for (int i0 = tid; i0 < n; i0 += 8 * ONE_THREADS) {
					uint32_t pr[8];
					double f[8];
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const int i = min(i0 + u * ONE_THREADS, n - 1);
						pr[u] = V.pair[K.pair0 + lo + i];
						f[u] = V.fac[K.pair0 + lo + i];
					}
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const int i = i0 + u * ONE_THREADS;
						if (i >= n) break;
						const uint32_t a = pr[u] & 0xFFFFu, b = pr[u] >> 16;
						const int d = k <= 32 ? hamm_d32((uint32_t)hb[2 * a], (uint32_t)hb[2 * b], s1, s2, notm)
						                      : hamm_d(hb[2 * a], hb[2 * a + 1], hb[2 * b], hb[2 * b + 1], s1a, s1b, s2a, s2b);
						xs[i] = f[u] * tab_s[d];
					}
				}
				__syncthreads();
				if (c == 0 && r == 0) STAMP(2);
				// ---- cells: thread = cell; the round's first cell may continue one begun in the round before (carry in),
				// its last one may go on into the next round (carry out); every other cell lies inside the round
				for (int j = jf; j < K.n_cell; j += ONE_THREADS) {
					const int o0 = j == jf ? o0f : off[j];
					if (o0 >= hi) break;
					const int o1 = j == jf ? o1f : off[j + 1];
					const int a = max(o0, lo) - lo, b = min(o1, hi) - lo;
					const double s = add_in_order(o0 < lo ? carry_s[r & 1] : 0.0, xs + a, b - a);
					if (o1 > hi) carry_s[(r + 1) & 1] = s;
					else { cs[j - j0] = s; put_dev(&dense[j == jf ? pjf : cp[j]], s); }
				}
				__syncthreads();
				if (c == 0 && r == 0) STAMP(3);
				// ---- total: the completed cells in cell order (lane 0).  (The cell that holds the next round's first pair is
				// either unfinished or begins there: not this round's.)
				if (tid == 0) total_s = add_in_order(total_s, cs, j1 - j0);
				__syncthreads();
				if (c == 0 && r == 0) STAMP(4);
			}
			if (tid == 0) {
				put_dev(&V.tot[c], total_s);
				put_dev(&V.inv[c], 1 / total_s);                      // src/LibHLA.cpp:1827 (inf when total == 0)
				put_dev(&V.wdev[c], w);
			}
		} else if (tid == 0) {
			// skipped (src/LibHLA.cpp:2451), or without haplotypes: total 0 -- the ensemble loop below decides by the weight
			put_dev(&V.tot[c], 0.0); put_dev(&V.inv[c], w > 0 ? 1 / 0.0 : 0.0); put_dev(&V.wdev[c], w);
		}
	}
	STAMP(5);
	if (V.stamps_on && tid == 0 && blockIdx.x < 120) V.stamps[16 + blockIdx.x] = wall_clock64();
	// ---- barrier over the launch's workgroups (all resident: n_group is what the device holds at once).  This one's stores
	// have completed past the L2 when it arrives; it waits, polling past the L2, for everybody else's.
	stores_done();
	__syncthreads();
	if (tid == 0) {
		__hip_atomic_fetch_add(V.arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned long long want = call * (unsigned long long)G;
		unsigned spins = 0;
		int ok = 1;
		while (__hip_atomic_load(V.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
			__builtin_amdgcn_s_sleep(2);
			if (++spins > V.spin_limit) { ok = 0; break; }                // (tenths of a second: something else holds the device's CUs --
			                                                              //  the host then runs the call again on ONE workgroup, which needs nobody)
		}
		ok_s = ok;
	}
	__syncthreads();
	STAMP(6);
	if (!ok_s) {                                                      // never hand the host numbers the kernel cannot vouch for
		if (tid == 0) __hip_atomic_store((uint32_t *)V.flag, 0x80000000u | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	// ---- ensemble: this workgroup's share of the posterior cells, [p_lo, p_hi).  For a chunk of classifiers their weights,
	// reciprocals and totals, and for a tile of cells the chunk's cell sums, go to LDS in ONE batch of loads (a latency, not a
	// hundred); then thread = cell: S[p] += (cell * (1 / total)) * w over the classifiers IN ORDER (:1497-1507).  A structurally
	// empty cell reads +0.0 and adds (0 * inv) * w: +0.0 -- nothing -- for a finite 1 / total, and NaN where the total is 0 or so
	// small that 1 / total overflows, which is what the reference's loop produces there (0 * inf).
	const int p_lo = (int)((long long)P * blockIdx.x / G), p_hi = (int)((long long)P * (blockIdx.x + 1) / G);
	const int CC = min(C, ONE_CHUNK), PT = max(1, min(ONE_THREADS - 2, ONE_ROUND / max(CC, 1) - 2));      // cells per tile (+ two scalar columns)
	// Phase A, every thread: the terms t[q][j] = (cell * (1 / total)) * w of classifier q and cell j, or +0.0 where the
	// classifier is skipped (w <= 0: AddProbToSum leaves the sum alone, and S + (+0.0) is S bit for bit -- S is never -0.0);
	// column np holds the weight itself (-> the sum of weights, :1505), column np + 1 total * w (-> the matching value's
	// numerator, :2458; its denominator is the sum of weights again, the same additions in the same order).
	// Phase B, lane = column: the column's terms added IN classifier order, sixteen LDS reads in flight -- the only serial part.
	const bool match_wg = (int)blockIdx.x == G - 1;
	double *res_s = cs;                                               // the two scalar columns' sums, for every thread to see
	int pt = p_lo;
	do {                                                              // (at least once: the matching value needs a pass even without cells)
		const int np = max(0, min(PT, p_hi - pt)), W = np + 2;
		const int wx_log = 32 - __clz(W - 1), wx = 1 << wx_log, qpb = ONE_THREADS >> wx_log;
		double acc = 0;
		for (int cb = 0; cb < C; cb += CC) {
			const int nc = min(CC, C - cb);
			__syncthreads();
			// every load of the chunk requested before the first is looked at (one memory round trip): thread = (classifier of
			// a batch, column), the columns rounded up to a power of two so that the split costs a shift, four batches in flight
			{
				const int j = tid & (wx - 1), qt = tid >> wx_log, jc = min(j, max(np, 1) - 1);
				for (int qb = 0; qb < nc; qb += 4 * qpb) {
					double v[4], wv[4], iv[4], tv[4];
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int q = cb + min(qb + u * qpb + qt, nc - 1);
						wv[u] = get_dev(&V.wdev[q]); iv[u] = get_dev(&V.inv[q]); tv[u] = get_dev(&V.tot[q]);
						v[u] = np > 0 ? get_dev(&V.dense[(size_t)q * P + pt + jc]) : 0.0;
					}
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int q = qb + u * qpb + qt;
						if (j >= W || q >= nc) continue;
						const bool on = wv[u] > 0;                    // AddProbToSum skips the classifier otherwise (:1497-1507)
						const double t = j < np ? (v[u] * iv[u]) * wv[u] : j == np ? wv[u] : tv[u] * wv[u];
						xs[q * W + j] = on ? t : 0.0;
					}
				}
			}
			__syncthreads();
			if (cb == 0 && pt == p_lo) STAMP(8);
			if (tid < W) {
				int q = 0;
				for (; q + 16 <= nc; q += 16) {
					double t[16];
#pragma unroll
					for (int u = 0; u < 16; u++) t[u] = xs[(q + u) * W + tid];
#pragma unroll
					for (int u = 0; u < 16; u++) acc += t[u];
				}
				for (; q < nc; q++) acc += xs[q * W + tid];
			}
		}
		if (tid >= np && tid < W) res_s[tid - np] = acc;
		__syncthreads();
		if (pt == p_lo) STAMP(9);
		const double sum_w = res_s[0];
		if (tid < np) put_host(&V.out[pt + tid], sum_w > 0 ? acc * (1.0 / sum_w) : acc);   // NormalizeSumPostProb (:1509-1518)
		if (match_wg && pt == p_lo && tid == 0) put_host(&V.out[P], res_s[1] / sum_w);     // the matching value (:2480)
		pt += PT;
	} while (pt < p_hi);
	STAMP(7);
	// ---- the last workgroup to have its outputs in host memory reports the call.  The chain the host relies on is a formal one:
	// every workgroup's output stores happen-before its release increment of `finished` (the workgroup barrier in between makes
	// the other wavefronts' stores thread 0's to release), the last workgroup's increment acquires all of them (a release sequence
	// on the counter), and its release store of the sequence number at system scope pairs with the host's acquire load of the flag.
	// (The stores themselves already went past the caches -- system-scope write-through -- so the releases have next to nothing to
	// write back.)
	stores_done();
	__syncthreads();
	STAMP(10);
	if (tid == 0) {
		const unsigned long long n = __hip_atomic_fetch_add(V.finished, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (n + 1 == call * (unsigned long long)G) __hip_atomic_store((uint32_t *)V.flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

struct OneState {
	bool active = false;
	int device = 0;
	OneView V{};
	int P = 0, C = 0;
	void *d_model = nullptr, *d_call = nullptr;           // one arena each
	void *h_pin = nullptr;                                // host-mapped: genotypes + weights in, posterior + matching + sequence number out
	size_t o_w = 0, o_out = 0, o_flag = 0, o_stamps = 0;
	hipStream_t st = nullptr;
	uint32_t seq = 0;
	unsigned long long calls = 0;                         // launches so far (the barrier counters count n_group arrivals per launch)
	int n_group_full = 1;                                 // workgroups of a launch while the device is this process's alone
	bool counters_stale = false;                          // a launch failed or timed out: the device counters are zeroed before the next one
	int last_groups = 0;                                  // workgroups of the last launch (the counters count per launch of THAT width)
	long long degraded_calls = 0;                         // calls that had to be repeated on one workgroup (hibag_sample_degraded_calls)
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
	if (g1.st) { (void)hipStreamSynchronize(g1.st); (void)hipStreamDestroy(g1.st); }
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
	std::vector<uint32_t> pair;
	std::vector<double> fac;
	std::vector<int> cell_off, cell_p;
	std::vector<OneRound> rounds;
	std::vector<OneCls> cls(std::max(C, 1));
	size_t n_hap_total = 0;
	for (int c = 0; c < C; c++) {
		const int H = n_haplo[c], k = n_snp[c];
		if (H < 0 || H > 65535 || k < 0 || k > 128) one_throw("predict_init: invalid classifier");
		OneCls &K = cls[c];
		K.hap0 = (int)n_hap_total; K.n_hap = H; K.k = k;
		K.pair0 = (long long)pair.size();
		K.cell0 = (int)cell_p.size();
		K.round0 = (int)rounds.size();
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
		}
		n_hap_total += (size_t)H;
		for (int a = 0; a < n_hla; a++) st[a + 1] += st[a];
		// the flat pair list: cells in posterior order, inside a cell the reference's order (src/LibHLA.cpp:1776-1821: first
		// haplotype ascending, then the second; on a diagonal cell the pair (i, i) with f * f first, then (i, j > i))
		size_t n_pair = 0;
		int p = 0;
		for (int h1 = 0; h1 < n_hla; h1++)
			for (int h2 = h1; h2 < n_hla; h2++, p++) {
				if (st[h1] == st[h1 + 1] || st[h2] == st[h2 + 1]) continue;
				cell_off.push_back((int)n_pair);
				cell_p.push_back(p);
				for (int a = st[h1]; a < st[h1 + 1]; a++) {
					const double fa = p_haplo[c][a].freq;
					int b = st[h2];
					if (h1 == h2) { pair.push_back((uint32_t)a | ((uint32_t)a << 16)); fac.push_back(fa * fa); b = a + 1; }      // :1786
					const double ff = 2 * fa;                                                                                  // :1789-1793, :1808-1812
					for (; b < st[h2 + 1]; b++) { pair.push_back((uint32_t)a | ((uint32_t)b << 16)); fac.push_back(ff * p_haplo[c][b].freq); }
				}
				n_pair = pair.size() - (size_t)K.pair0;
				if (n_pair > 0x7FFFFFF0u) one_throw("predict_init: a classifier has too many haplotype pairs");
			}
		cell_off.push_back((int)n_pair);
		K.n_pair = (int)n_pair;
		K.n_cell = (int)cell_p.size() - K.cell0;
		// rounds: at most ONE_ROUND pairs, and at most ONE_CELLS cells completed, each; a round begins at pair `lo` inside cell `j`
		const int *off = cell_off.data() + K.cell0 + c;
		int lo = 0, j = 0;
		K.n_round = 0;
		while (lo < K.n_pair) {
			rounds.push_back(OneRound{lo, j});
			K.n_round++;
			int hi = std::min(K.n_pair, lo + ONE_ROUND);
			if (j + ONE_CELLS < K.n_cell && off[j + ONE_CELLS] < hi) hi = off[j + ONE_CELLS];      // (thousands of one-pair cells)
			while (j < K.n_cell && off[j + 1] <= hi) j++;           // the cell that holds pair `hi` (n_cell at the end)
			lo = hi;
		}
		rounds.push_back(OneRound{K.n_pair, K.n_cell});
		K.r0_hi = rounds[(size_t)K.round0 + (K.n_round ? 1 : 0)].pair0;
		K.r0_cell1 = rounds[(size_t)K.round0 + (K.n_round ? 1 : 0)].cell0;
	}
	double tab[HIBAG_TAB_N];
	for (int i = 0; i < HIBAG_TAB_N; i++) tab[i] = std::exp(i * std::log(1e-5));     // src/LibHLA.cpp:166-183
	tab[0] = 1;
	for (int i = 0; i < HIBAG_TAB_N; i++) if (!std::isfinite(tab[i])) tab[i] = 0;

	// model arena
	size_t o = 0;
	auto take = [&](size_t bytes) { const size_t at = o; o = (o + std::max<size_t>(bytes, 8) + 63) & ~(size_t)63; return at; };
	const size_t o_cls = take(cls.size() * sizeof(OneCls)), o_bits = take(bits.size() * 8), o_pair = take(pair.size() * 4), o_fac = take(fac.size() * 8),
		o_off = take(cell_off.size() * 4), o_cp = take(cell_p.size() * 4), o_rc = take(rounds.size() * sizeof(OneRound)), o_tab = take(sizeof(tab));
	ONE_OK(hipMalloc(&g1.d_model, o), "hipMalloc(model)");
	char *d = (char *)g1.d_model;
	auto put = [&](size_t at, const void *src, size_t bytes) { if (bytes) ONE_OK(hipMemcpy(d + at, src, bytes, hipMemcpyHostToDevice), "copy model"); };
	put(o_cls, cls.data(), cls.size() * sizeof(OneCls)); put(o_bits, bits.data(), bits.size() * 8);
	put(o_pair, pair.data(), pair.size() * 4); put(o_fac, fac.data(), fac.size() * 8);
	put(o_off, cell_off.data(), cell_off.size() * 4); put(o_cp, cell_p.data(), cell_p.size() * 4);
	put(o_rc, rounds.data(), rounds.size() * sizeof(OneRound));
	put(o_tab, tab, sizeof(tab));
	// per-call device arena: totals, reciprocals, weights, the finished-workgroup counter, the dense cell sums (zeroed once,
	// here: the structurally empty cells are never written)
	size_t oc = 0;
	auto takec = [&](size_t bytes) { const size_t at = oc; oc = (oc + std::max<size_t>(bytes, 8) + 63) & ~(size_t)63; return at; };
	const size_t c_tot = takec((size_t)C * 8), c_inv = takec((size_t)C * 8), c_w = takec((size_t)C * 8), c_cnt = takec(16),
		c_dense = takec((size_t)C * P * 8);
	ONE_OK(hipMalloc(&g1.d_call, oc), "hipMalloc(call)");
	ONE_OK(hipMemset(g1.d_call, 0, oc), "hipMemset(call)");
	// host-mapped: [genotypes | weights] in, [posterior, matching | sequence number] out
	size_t oh = 0;
	auto takeh = [&](size_t bytes) { const size_t at = oh; oh = (oh + std::max<size_t>(bytes, 8) + 63) & ~(size_t)63; return at; };
	const size_t h_geno = takeh((size_t)C * sizeof(PluginGenotype));
	g1.o_w = takeh((size_t)C * 8); g1.o_out = takeh((size_t)(P + 1) * 8); g1.o_flag = takeh(8); g1.o_stamps = takeh((16 + 120) * 8);
	ONE_OK(hipHostMalloc(&g1.h_pin, oh, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
	memset(g1.h_pin, 0, oh);
	char *dh = nullptr;
	ONE_OK(hipHostGetDevicePointer((void **)&dh, g1.h_pin, 0), "hipHostGetDevicePointer");
	ONE_OK(hipStreamCreateWithFlags(&g1.st, hipStreamNonBlocking), "hipStreamCreate");
	char *dc = (char *)g1.d_call;
	OneView &V = g1.V;
	V.n_hla = n_hla; V.n_cell = P; V.n_classifier = C;
	V.cls = (const OneCls *)(d + o_cls); V.bits = (const uint64_t *)(d + o_bits); V.pair = (const uint32_t *)(d + o_pair);
	V.fac = (const double *)(d + o_fac); V.cell_off = (const int *)(d + o_off); V.cell_p = (const int *)(d + o_cp);
	V.round = (const OneRound *)(d + o_rc); V.tab = (const double *)(d + o_tab);
	V.geno = (const uint64_t *)(dh + h_geno); V.weight = (const double *)(dh + g1.o_w);
	V.out = (double *)(dh + g1.o_out); V.flag = (volatile uint32_t *)(dh + g1.o_flag); V.stamps = (unsigned long long *)(dh + g1.o_stamps);
	V.tot = (double *)(dc + c_tot); V.inv = (double *)(dc + c_inv); V.wdev = (double *)(dc + c_w);
	V.arrived = (unsigned long long *)(dc + c_cnt); V.finished = V.arrived + 1;
	V.dense = (double *)(dc + c_dense);
	// the launch's workgroups meet at a barrier: no more of them than the device holds at once
	int per_cu = 0, cus = 0;
	ONE_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hibag_selected_device()), "hipDeviceGetAttribute");
	ONE_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_one, ONE_THREADS, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
	if (per_cu < 1 || cus < 1) one_throw("predict_init: the per-sample kernel does not fit the device");
	V.n_group = g1.n_group_full = std::max(1, std::min(C, per_cu * cus));
	// ~2^21 polls of ~0.1 us: a fifth of a second.  (HIBAG_ONE_SPIN: tests shorten it to provoke the fall-back.)
	V.spin_limit = getenv("HIBAG_ONE_SPIN") ? (unsigned)std::max(0, atoi(getenv("HIBAG_ONE_SPIN"))) : (1u << 21);
	V.stamps_on = getenv("HIBAG_ONE_STAMPS") != nullptr;
	g1.P = P; g1.C = C; g1.device = hibag_selected_device();
	g1.seq = 0; g1.calls = 0; g1.counters_stale = false; g1.degraded_calls = 0; g1.last_groups = 0;
	g1.active = true;
}

long long hibag_sample_degraded_calls() { return g1.degraded_calls; }

// predict_avg_prob(geno[nClassifier], weight[nClassifier], out_prob[P], out_match[1]): src/LibHLA.cpp:2433-2441
void hibag_sample_avg_prob(const PluginGenotype geno[], const double weight[], double out_prob[], double out_match[])
{
	if (!g1.active) one_throw("predict_avg_prob: predict_init was not called");
	const int C = g1.C, P = g1.P;
	char *h = (char *)g1.h_pin;
	if (C == 0) {                                                          // no classifier: the sums are all zero (src/LibHLA.cpp:1491-1495), 0 / 0 matching
		memset(out_prob, 0, (size_t)P * 8);
		out_match[0] = 0.0 / 0.0;
		return;
	}
	ONE_OK(hipSetDevice(g1.device), "hipSetDevice");
	memcpy(h, geno, (size_t)C * sizeof(PluginGenotype));
	memcpy(h + g1.o_w, weight, (size_t)C * 8);
	volatile uint32_t *flag = (volatile uint32_t *)(h + g1.o_flag);
	// One launch of `groups` workgroups and the wait for its report.  Returns false when the launch's workgroups did not all
	// become resident within the barrier's budget (the kernel's word, top bit of the flag); throws on anything the runtime
	// reports.  Whatever goes wrong leaves `counters_stale` set, so that the next launch -- of this call or of a later one --
	// starts from zeroed device counters: a failure never outlives the call it happened in.
	auto run = [&](int groups) -> bool {
		// (the barrier counters only ever grow, `calls * groups` arrivals ending call number `calls`: a launch of another width
		// than the last one starts them over, like a failed one)
		if (g1.counters_stale || (g1.last_groups != 0 && g1.last_groups != groups)) {
			ONE_OK(hipStreamSynchronize(g1.st), "predict_avg_prob (draining a failed launch)");     // its workgroups may still be giving up
			ONE_OK(hipMemsetAsync(g1.V.arrived, 0, 16, g1.st), "hipMemsetAsync(counters)");
			g1.calls = 0;
		}
		g1.counters_stale = true;                                          // until this launch has reported
		g1.last_groups = groups;
		const uint32_t seq = g1.seq = g1.seq % 0x7FFFFFFEu + 1;            // 1 .. 2^31 - 2: never the flag's initial value, top bit free
		OneView V = g1.V;
		V.n_group = groups;
		hipLaunchKernelGGL(k_one, dim3(groups), dim3(ONE_THREADS), 0, g1.st, V, seq, g1.calls + 1);
		ONE_OK(hipGetLastError(), "launch");
		g1.calls++;                                                        // (only a launch that happened counts)
		// the last workgroup to finish stores the call's sequence number in host memory: poll it (sooner than the runtime's
		// completion signal) with acquire loads -- they pair with the kernel's release store, so the posterior read below is
		// ordered behind it; after a second without it, ask the runtime what happened
		const auto t0 = std::chrono::steady_clock::now();
		uint32_t f;
		for (unsigned spins = 0; ((f = __atomic_load_n((const uint32_t *)flag, __ATOMIC_ACQUIRE)) & 0x7FFFFFFFu) != seq; spins++) {
			_mm_pause();
			if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) {
				ONE_OK(hipStreamSynchronize(g1.st), "predict_avg_prob");
				if (((f = __atomic_load_n((const uint32_t *)flag, __ATOMIC_ACQUIRE)) & 0x7FFFFFFFu) != seq)
					one_throw("predict_avg_prob: the kernel finished without reporting the call");
				break;
			}
		}
		if (f & 0x80000000u) return false;
		g1.counters_stale = false;
		return true;
	};
	// The full launch meets at a barrier over its workgroups, which therefore must all be resident at once.  The runtime
	// cannot promise that on a shared device (another stream or process may hold compute units), so a launch whose
	// workgroups gave up waiting is not an error: the call is run again on ONE workgroup -- slower (the classifiers one after
	// the other), but it waits for nobody -- and so are the next calls, until one in `ONE_RETRY_FULL` tries the full width again.
	constexpr long long ONE_RETRY_FULL = 64;
	const bool try_full = g1.degraded_calls == 0 || g1.degraded_calls % ONE_RETRY_FULL == 0;
	if (!(try_full && run(g1.n_group_full))) {
		g1.degraded_calls++;
		if (!run(1)) one_throw("predict_avg_prob: a single-workgroup launch reported a barrier time-out");      // (cannot happen: one arrival)
	} else if (g1.degraded_calls > 0) {
		g1.degraded_calls = 0;                                             // the device is ours again
	}
	const uint32_t seq = g1.seq;
	static const bool show = getenv("HIBAG_ONE_STAMPS") != nullptr;        // diagnostic: where a call's device time goes
	if (show && seq >= 100 && seq < 104) {
		const unsigned long long *t = (const unsigned long long *)(h + g1.o_stamps);
		auto us = [&](int a, int b) { return (double)(long long)(t[b] - t[a]) / 100.0; };
		fprintf(stderr, "[k_one] workgroup 0: table + staging %.1f, pairs %.1f, cells %.1f, total %.1f, further rounds / classifiers %.1f, "
			"barrier %.1f, ensemble %.1f (loads + terms %.1f, ordered sums %.1f, rest %.1f), its stores done %.1f us; start to end of its ensemble %.1f us\n",
			us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(6, 8), us(8, 9), us(9, 7), us(7, 10), us(0, 7));
		fprintf(stderr, "[k_one] arrival at the barrier after workgroup 0's start, us, by workgroup:");
		for (int b = 0; b < std::min(g1.V.n_group, 120); b++) fprintf(stderr, " %.1f", us(0, 16 + b));
		fprintf(stderr, "\n");
	}
	const double *h_out = (const double *)(h + g1.o_out);
	memcpy(out_prob, h_out, (size_t)P * 8);
	out_match[0] = h_out[P];
}
