// hibag_em.hip -- the EM haplotype-frequency fit of a growth step's candidate SNPs on the device.
//
// CAlg_EM::PrepareNewSNP + ExpectationMaximization (src/LibHLA.cpp:1127-1255) once per candidate SNP: the in-bag samples'
// haplotype pairs (CAlg_EM::PrepareHaplotypes, :1002-1125) index the DOUBLED haplotype list (DoubleHaplos, :416-442:
// entry 2i carries allele 0 of the new SNP, 2i + 1 allele 1), a pair is compatible with a sample's genotype g at the new
// SNP iff (h1 & 1) + (h2 & 1) == g (every pair where it is missing), and the iteration
//     E   G_j = (h1 != h2 ? (2 f[h1]) f[h2] : f[h1] f[h2]);  psum_i = sum_j G_j;  loglik += boot_i log(psum_i);  G_j *= boot_i / psum_i
//     M   f'[h] = (sum of the G_j of the pairs that contain h, in pair order, a homozygous pair twice) * (0.5 / n_samp)
// runs until |loglik - loglik_before| <= sqrt(eps) (|loglik_0| + sqrt(eps)), at most 500 times.
//
// Until round 4 the host fitted the candidates on a thread per CPU (hibag_train.hip fit_new_snp; it still does where this
// file says it must): with eight ranks on one node's cores -- two threads each on the pool's boxes -- the fit was 1.6 x
// slower and the longest part of a growth step.  Here: workgroup = candidate, the growth step's pair set and the
// candidate's state in LDS, and an iteration in barrier-separated phases (k_em_fit below): thread = pair for the products,
// thread = sample for its psum in list order (an incompatible pair's G is +0.0, and x + (+0.0) is x bit for bit -- no sum
// here can be -0.0 -- so nothing downstream needs the compatibility test again), thread = haplotype for the M step (the G of
// its pairs, a transposed list built once per growth step, in pair order), and one thread that does nothing but the
// samples' terms in sample order.  Every sum is formed by one thread in the host's order with the host's operations
// (-ffp-contract=off), so the fitted frequencies are bit-identical to the host's.
//
// What it buys (profiles/r04_cfg5_notes.txt): the fit no longer depends on the host's cores -- 0.7 ms per growth step
// whatever the rank's share of them, against 0.28 ms on 16 host threads and 0.8 ms on two.  It is a latency-bound kernel
// (the slowest candidate of a step takes 60 iterations of ~11 us: dependent FP64 additions at 12 cycles each and
// per-lane LDS gathers with bank conflicts), so the trainer uses it where a rank has two host threads or fewer
// (hibag_hip_trainer_set_em_mode) and the host threads otherwise.
//
// The ONE thing the device cannot reproduce bit for bit is log(): the host's is glibc's, the device's ocml's, both good to
// about an ulp and not identical.  loglik only ever decides WHEN the iteration stops, so the device decides with a margin:
// with B >= |loglik_device - loglik_host| (a bound from the logs' ulp errors and the length of the sum, computed beside the
// sum), the test |dL| <= tol is settled whenever |dL| is further than the accumulated bounds from tol -- then host and
// device agree for certain -- and where it is not (a band eight orders of magnitude narrower than tol) the candidate is
// handed back to the host, which fits it with its own log().  Stored models are reproduced bit for bit
// (tests/test_hip_train_driver.py, tests/test_hip_configs.py) with the host idle.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <mutex>
#include <vector>
#include "hibag_em.h"
#include "hibag_combine.h"

namespace {

#define EM_THREADS 1024
#define HIBAG_WAVE_EM 64
#define EM_MAX_ITER 500                     // src/LibHLA.cpp:98
#define EM_INIT_VAL_FRAC 0.001              // :100
#define EM_LDS_BYTES (156 * 1024)           // a growth step's pair set and a candidate's state live in LDS (a CU has 160 KB); larger steps go to the host

struct EmView {
	int n_ib, n_pair, n_hap, n_samp_total;  // in-bag samples, their pairs, doubled haplotypes, all samples (the M step's 0.5 / total)
	int n_ent;                              // entries of hent (padding included)
	int n_cand;                             // candidates of this growth step (= its workgroups in a fused launch)
	const uint32_t *pw;                     // [n_pair] h1 | h2 << 14
	const int *off;                         // [n_ib + 1]
	const int *boot;                        // [n_ib]
	const int *hoff;                        // [n_hap + 1] transposed lists: the pairs that contain haplotype h ...
	const uint32_t *hent;                   // ... in pair order, a homozygous pair twice: BYTE offsets into G, every list padded to a multiple of 4 entries with n_pair * 8 (a slot that holds +0.0); hoff counts entries incl. padding
	const uint16_t *pos;                    // or (round 6, where LDS has room: EmLayout::staged) the INVERSE of those lists: [2 n_pair] the two entries pair j has in them -- its scaled G is
	                                        // written there (phase B'), so a haplotype's G are contiguous and phase C reads them in order instead of gathering; null: hent is used
	const double *cur_freq;                 // [n_hap / 2]
	// per candidate
	const int8_t *geno;                     // [n_cand][n_ib] genotype of the in-bag samples at the candidate SNP: 0, 1, 2, 3 = missing
	const double *afreq;                    // [n_cand] allele frequency in the bag (DoubleHaplosInitFreq, :444-459)
	double *out_freq;                       // [n_cand][n_hap]
	int *status;                            // [n_cand] 1 = fitted, 2 = the host must fit it; [n_cand + c]: iterations
	long long *stamps;                      // measurement only (HIBAG_EM_STAMPS=1, else null): [n_cand][6] ticks of the 100 MHz clock candidate c spent in set-up, phases A, B, B', C and in all
};

// LDS layout of k_em_fit (bytes), shared by the kernel and the host's size check
struct EmLayout {
	size_t oldf, newf, G, lterm, rs, pw, off, bt, hoff, hent, ps, gt, total;
	// staged: the lists' G values have a (transposed) copy of their own, `gt`, and `hent` holds 16-bit positions instead of offsets
	__host__ __device__ EmLayout(int n_ib, int n_pair, int n_hap, bool staged)
	{
		size_t o = 0;
		auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 7) & ~(size_t)7; return at; };
		oldf = take((size_t)n_hap * 8); newf = take((size_t)n_hap * 8); G = take((size_t)(n_pair + 1) * 8 + 8); lterm = take((size_t)n_ib * 16); rs = take((size_t)n_ib * 8);
		pw = take((size_t)n_pair * 4); off = take((size_t)(n_ib + 1) * 4); bt = take((size_t)n_ib * 4); hoff = take((size_t)(n_hap + 1) * 4);
		const size_t n_ent_max = (size_t)2 * n_pair + 4 * (size_t)n_hap;
		hent = take(staged ? (size_t)4 * n_pair + 16 : n_ent_max * 4 + 16); ps = take((size_t)n_pair * 2);
		gt = take(staged ? n_ent_max * 8 + 16 : 0);
		total = o;
	}
};

// Workgroup = candidate.  An iteration is three barrier-separated phases of about a microsecond each --
//   A  thread = pair:       G_j = (h1 != h2 ? (2 f[h1]) f[h2] : f[h1] f[h2]), +0.0 for a pair the sample's genotype rules out
//   B  thread = sample:     psum = its G in list order;  term = boot log(psum);  its compatible G *= boot / psum
//   C  thread = haplotype:  f'[h] = (its pairs' G in pair order) * (0.5 / n)
// -- and the one long serial chain, loglik = the samples' terms in sample order (632 dependent additions: three
// microseconds), runs BESIDE them on a thread that does nothing else: a third of the chain in each phase of the NEXT
// iteration (the terms are double-buffered).  So the stopping test of iteration k is known at the end of iteration k + 1; a
// fit that stops at k returns the frequencies it had then -- oldf, which iteration k + 1 only read.
// A launch fits the candidates of SEVERAL growth steps -- one per trainer that runs beside others (hibag_combine.h) --: the
// workgroups of step j are M.first[j] .. M.first[j + 1] - 1, each with its own pair set, sizes and LDS layout.
__global__ __launch_bounds__(EM_THREADS) void k_em_fit(HibagMulti<EmView> M)
{
	const int owner = hibag_multi_owner(M, (int)blockIdx.x);
	const EmView V = M.v[owner];
	extern __shared__ __attribute__((aligned(16))) char lds[];
	__shared__ int verdict_s;                                         // of the iteration before: 0 = go on, 1 = converged, 2 = cannot tell (host)
	__shared__ double tabs_s[2][EM_THREADS / HIBAG_WAVE_EM];         // sum of |term| of an iteration, by parity and wavefront (only ever a BOUND: its own order of summation is free)
	const bool staged = V.pos != nullptr;
	const EmLayout L(V.n_ib, V.n_pair, V.n_hap, staged);
	double *oldf = (double *)(lds + L.oldf), *newf = (double *)(lds + L.newf), *G = (double *)(lds + L.G), *lterm = (double *)(lds + L.lterm);
	uint32_t *pw = (uint32_t *)(lds + L.pw);                          // h1 | h2 << 14 | genotype of the pair's sample << 28
	int *off = (int *)(lds + L.off), *bt = (int *)(lds + L.bt), *hoff = (int *)(lds + L.hoff);
	uint32_t *hent = (uint32_t *)((((uintptr_t)(lds + L.hent)) + 15) & ~(uintptr_t)15);
	uint16_t *pos = (uint16_t *)hent;                                 // (staged: the same bytes hold the pairs' positions)
	double *Gt = (double *)((((uintptr_t)(lds + L.gt)) + 15) & ~(uintptr_t)15);
	uint16_t *ps = (uint16_t *)(lds + L.ps);                          // the sample of each pair
	double *rs = (double *)(lds + L.rs);                              // boot_i / psum_i
	const int tid = threadIdx.x, c = (int)blockIdx.x - M.first[owner], n_ib = V.n_ib, n_pair = V.n_pair, n_hap = V.n_hap;
	// The loglik chain has a WAVEFRONT of its own (round 6): its first lane adds, the other 63 do nothing.  Until then the chain's
	// thread was the last lane of a wavefront whose other lanes were workers -- a wavefront runs both sides of a divergent branch
	// one after the other, so every phase lasted that wavefront's worker share PLUS its piece of the chain instead of the longer
	// of the two.  It also runs at raised priority: a chain of dependent FP64 additions that has to take turns with three busy
	// wavefronts on its SIMD waited 28-34 cycles per addition against 12 alone (profiles/r04_cfg5_notes.txt).
	// ... and (later in round 6) a SIMD of its own: raised priority did not get the chain more than its turn -- in-kernel stamps
	// (HIBAG_EM_STAMPS) showed every phase lasting exactly its piece of the chain at 29 cycles per addition, the rate of one
	// wavefront among four -- so the wavefronts that share the chain's SIMD (read from HW_ID; a workgroup's sixteen wavefronts
	// sit four to a SIMD) do no work at all.  The twelve others are the workers; no phase has more than one turn per thread
	// for them either (632 samples, ~600 haplotypes, 2,600 pairs four at a time).
	__shared__ int simd_s[EM_THREADS / HIBAG_WAVE_EM];
	const int wave = tid >> 6, last = EM_THREADS / HIBAG_WAVE_EM - 1;
	{
		uint32_t hw_id;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
		if ((tid & 63) == 0) simd_s[wave] = (int)((hw_id >> 4) & 3u);  // SIMD_ID
	}
	__syncthreads();
	int n_apart = 0, rank_apart = 0;
	for (int w = 0; w < last; w++) { const int apart = simd_s[w] != simd_s[last]; n_apart += apart; rank_apart += apart && w < wave; }
	const bool spread = n_apart > 0;                                  // (else: every wavefront on one SIMD -- all of them work, as before)
	const bool summer = tid == last * HIBAG_WAVE_EM;                  // the thread of the loglik chain (no other work)
	const bool worker = wave != last && (!spread || simd_s[wave] != simd_s[last]);
	const int nw = (spread ? n_apart : last) * HIBAG_WAVE_EM;         // worker threads
	const int wid = (spread ? rank_apart : wave) * HIBAG_WAVE_EM + (tid & 63);    // a worker's index among them
	if (wave == last) __builtin_amdgcn_s_setprio(3);
	// the growth step's pair set (the same for every candidate) and this candidate's genotypes: once into LDS
	if (staged) {
		for (int e = tid; e < 2 * V.n_pair; e += EM_THREADS) pos[e] = V.pos[e];
		for (int e = tid; e < V.n_ent; e += EM_THREADS) Gt[e] = 0.0;      // (the lists' padding entries stay +0.0: s + (+0.0) is s)
	} else
		for (int e = tid; e < V.n_ent; e += EM_THREADS) hent[e] = V.hent[e];
	if (tid == 0) G[n_pair] = 0.0;                                   // what the lists' padding entries point at
	for (int i = tid; i <= n_ib; i += EM_THREADS) off[i] = V.off[i];
	for (int i = tid; i < n_ib; i += EM_THREADS) bt[i] = V.boot[i];
	for (int h = tid; h <= n_hap; h += EM_THREADS) hoff[h] = V.hoff[h];
	const double p1 = V.afreq[c], p0 = 1 - p1;                        // :447-448
	for (int h = tid; h < n_hap; h += EM_THREADS)                     // DoubleHaplosInitFreq, :444-459
		newf[h] = ((h & 1) ? p1 : p0) * V.cur_freq[h >> 1] + EM_INIT_VAL_FRAC;
	for (int i = tid; i < n_ib; i += EM_THREADS) {
		const uint32_t g = (uint32_t)(V.geno[(size_t)c * n_ib + i] & 3);
		for (int j = V.off[i]; j < V.off[i + 1]; j++) { pw[j] = V.pw[j] | (g << 28); ps[j] = (uint16_t)i; }
	}
	const double em_reltol = 1.4901161193847656e-08;                  // sqrt(DBL_EPSILON), :102
	const double scale = 0.5 / V.n_samp_total;
	// the summer's state: the chain of iteration (iter - 1)'s terms, its loglik history, the tolerance.  The chain is cut in
	// four pieces, one per phase of the next iteration, about as long as the phases are (A 12 %, B 22 %, B' 6 %, C 60 %).
	double chain = 0, conv_tol = 0, tol_bound = 0, loglik_prev = -1e+30, bound_prev = 0;
	// (staged lists: phase C is a third of an iteration instead of three fifths)
	const int cut[5] = {0, (n_ib * (staged ? 19 : 12) + 99) / 100, (n_ib * (staged ? 54 : 34) + 99) / 100, (n_ib * (staged ? 67 : 40) + 99) / 100, n_ib};
	// terms [cut[part], cut[part + 1]) in order.  The additions are one dependent chain -- the longest thing in an iteration
	// (in-kernel stamps, HIBAG_EM_STAMPS: every phase lasted as long as its piece of the chain, 28 cycles per term) -- so
	// nothing else may sit in it: the NEXT eight terms are requested before the current eight are added (round 6; until then
	// each group of sixteen waited for its own loads first).
	auto sum_part = [&](const double *t, int part) {
		const int hi = min(n_ib, cut[part + 1]);
		int i = min(n_ib, cut[part]);
		if (i + 8 <= hi) {
			double a[8], b[8];
#pragma unroll
			for (int u = 0; u < 8; u++) a[u] = t[i + u];
			for (; i + 16 <= hi; i += 8) {
#pragma unroll
				for (int u = 0; u < 8; u++) b[u] = t[i + 8 + u];
#pragma unroll
				for (int u = 0; u < 8; u++) chain += a[u];
#pragma unroll
				for (int u = 0; u < 8; u++) a[u] = b[u];
			}
#pragma unroll
			for (int u = 0; u < 8; u++) chain += a[u];
			i += 8;
		}
		for (; i < hi; i++) chain += t[i];
	};
	int iter = 0, stopped = 0;
	if (tid == 0) verdict_s = 0;
	long long tk[6] = {0, 0, 0, 0, 0, 0}, t_last = V.stamps ? (long long)wall_clock64() : 0;
	const long long t_first = t_last;
	auto stamp = [&](int k) { if (V.stamps) { const long long t = (long long)wall_clock64(); tk[k] += t - t_last; t_last = t; } };
	// The frequencies alternate between two buffers (round 6: no copy, and four barriers per iteration instead of six): iteration
	// `iter` READS `cur` and phase C writes `nxt`; the first reads the initial values, which the set-up put into `newf`.
	const double *cur = newf;
	double *nxt = oldf;
	__syncthreads();
	stamp(0);

	for (; iter <= EM_MAX_ITER; iter++) {
		cur = (iter & 1) ? oldf : newf; nxt = (iter & 1) ? newf : oldf;
		const double *t_prev = lterm + (size_t)((iter + 1) & 1) * n_ib;   // the terms of iteration iter - 1
		double *t_cur = lterm + (size_t)(iter & 1) * n_ib;
		if (summer) chain = 0;
		// ---- A
		if (summer) { if (iter > 0) sum_part(t_prev, 0); }
		else if (worker)
			for (int j0 = wid; j0 < n_pair; j0 += 4 * nw) {            // four pairs per turn: their look-ups in flight together
				uint32_t w[4];
				double fa[4], fb[4];
#pragma unroll
				for (int u = 0; u < 4; u++) w[u] = pw[min(j0 + u * nw, n_pair - 1)];
#pragma unroll
				for (int u = 0; u < 4; u++) { fa[u] = cur[w[u] & 0x3FFFu]; fb[u] = cur[(w[u] >> 14) & 0x3FFFu]; }
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int j = j0 + u * nw;
					if (j >= n_pair) break;
					const int a = (int)(w[u] & 0x3FFFu), b = (int)((w[u] >> 14) & 0x3FFFu), g = (int)(w[u] >> 28);
					double x = 0;
					if (g > 2 || (a & 1) + (b & 1) == g) x = a != b ? (2 * fa[u]) * fb[u] : fa[u] * fb[u];
					G[j] = x;
				}
			}
		__syncthreads();
		stamp(1);
		// ---- B: a sample's psum in list order (sixteen reads in flight), its term, its scaling factor
		double my_abs = 0;
		if (summer) { if (iter > 0) sum_part(t_prev, 1); }
		else if (worker)
			for (int i = wid; i < n_ib; i += nw) {
				const int j0 = off[i], j1 = off[i + 1], b_i = bt[i];
				double psum = 0;
				int j = j0;
				for (; j + 16 <= j1; j += 16) {
					double v[16];
#pragma unroll
					for (int u = 0; u < 16; u++) v[u] = G[j + u];
#pragma unroll
					for (int u = 0; u < 16; u++) psum += v[u];
				}
				for (; j < j1; j++) psum += G[j];
				const double term = b_i * log(psum);
				t_cur[i] = term;
				rs[i] = b_i / psum;
				my_abs += fabs(term);                                  // (NaN / inf stay what they are: the verdict below then is "cannot tell")
			}
		for (int d = 32; d > 0; d >>= 1) my_abs += __shfl_xor(my_abs, d);
		if ((tid & 63) == 0) tabs_s[iter & 1][tid >> 6] = worker ? my_abs : 0.0;
		__syncthreads();
		stamp(2);
		// ---- B': thread = pair again: the compatible G *= boot / psum (an incompatible pair stays +0.0 whatever the factor is)
		if (summer) { if (iter > 0) sum_part(t_prev, 2); }
		else if (worker)
			for (int j0 = wid; j0 < n_pair; j0 += 4 * nw) {
				uint32_t w[4];
				double r[4], x[4];
#pragma unroll
				for (int u = 0; u < 4; u++) { const int j = min(j0 + u * nw, n_pair - 1); w[u] = pw[j]; r[u] = rs[ps[j]]; x[u] = G[j]; }
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int j = j0 + u * nw;
					if (j >= n_pair) break;
					const int g = (int)(w[u] >> 28);
					const bool compatible = g > 2 || (int)((w[u] & 1u) + ((w[u] >> 14) & 1u)) == g;
					if (staged) {
						const double y = compatible ? x[u] * r[u] : x[u];     // (an incompatible pair's G is the +0.0 phase A wrote)
						Gt[pos[2 * j]] = y; Gt[pos[2 * j + 1]] = y;
					} else if (compatible) G[j] = x[u] * r[u];
				}
			}
		__syncthreads();
		stamp(3);
		// ---- C
		if (summer) {
			if (iter > 0) {
				sum_part(t_prev, 3);
				// the stopping test of iteration iter - 1 (:1247-1254), with a margin for the device's log().  With logs within
				// 2 ulp of each other a term differs by at most 3 ulp (the product with the count rounds again), and two in-order
				// sums of n such terms by at most (2 (n - 1) u + 7 u) T, u = 2^-53, T = the sum of the terms' magnitudes
				// (Higham, Accuracy and Stability of Numerical Algorithms, 4.2): (2 n + 6) u T.  Taken four times as wide:
				// 8 (n + 4) u T.  (Until round 6 T was bounded by n times the largest term and the whole taken eight times as
				// wide: a band 6 times wider than this one, in which 1 % of all fits landed and went back to the host, a third of a
				// millisecond of its time each -- profiles/r06_notes.txt item 4g.)
				double tsum = 0;
				for (int w = 0; w < EM_THREADS / HIBAG_WAVE_EM; w++) tsum += tabs_s[(iter + 1) & 1][w];
				const double loglik = chain, bound = 8.0 * 1.1102230246251565e-16 * (double)(n_ib + 4) * tsum;
				int verdict = 0;
				if (iter > 1) {
					const double d = fabs(loglik - loglik_prev), slack = bound + bound_prev + tol_bound;
					if (d + slack <= conv_tol) verdict = 1;           // the host's test succeeds for certain
					else if (!(d - slack > conv_tol)) verdict = 2;    // too close to call (or not a number): the host's own log() decides
				} else {
					conv_tol = em_reltol * (fabs(loglik) + em_reltol);
					if (conv_tol < 0) conv_tol = 0;
					tol_bound = em_reltol * bound;
					if (!(fabs(loglik) <= DBL_MAX)) verdict = 2;
				}
				loglik_prev = loglik; bound_prev = bound;
				verdict_s = verdict;
			}
		} else if (worker && staged)
			for (int h = wid; h < n_hap; h += nw) {
				// the haplotype's G in pair order are Gt[e0 .. e1): sixteen reads in flight, the additions in order
				const int e0 = hoff[h], e1 = hoff[h + 1];                 // (multiples of 4)
				// (requesting the next values before the current ones are added, as the chain above does, was measured here: 4.4 us
				// per iteration instead of 2.9 -- every worker is busy in this phase, and the copies between the two register sets cost
				// more than the exposed look-ups)
				double s = 0;
				int e = e0;
				for (; e + 16 <= e1; e += 16) {
					double v[16];
#pragma unroll
					for (int u = 0; u < 16; u++) v[u] = Gt[e + u];
#pragma unroll
					for (int u = 0; u < 16; u++) s += v[u];
				}
				for (; e < e1; e += 4) {
					double v[4];
#pragma unroll
					for (int u = 0; u < 4; u++) v[u] = Gt[e + u];
#pragma unroll
					for (int u = 0; u < 4; u++) s += v[u];
				}
				nxt[h] = s * scale;
			}
		else if (worker)
			for (int h = wid; h < n_hap; h += nw) {
				const int e0 = hoff[h], e1 = hoff[h + 1];                 // (multiples of 4: the lists are padded with a +0.0 slot)
				const char *Gb = (const char *)G;
				const uint32_t pad_off = (uint32_t)n_pair * 8u;
				double s = 0;
				// batches of 8 entries, software-pipelined: while a batch is added (the only ordered part), the next batch's G
				// values and the one after's offsets are on their way
				auto idx = [&](int e, uint4 (&q)[2]) {
#pragma unroll
					for (int u = 0; u < 2; u++) q[u] = e + 4 * u < e1 ? *(const uint4 *)(hent + e + 4 * u) : uint4{pad_off, pad_off, pad_off, pad_off};
				};
				auto gat = [&](const uint4 (&q)[2], double (&t)[8]) {
#pragma unroll
					for (int u = 0; u < 2; u++) {
						t[4 * u] = *(const double *)(Gb + q[u].x); t[4 * u + 1] = *(const double *)(Gb + q[u].y);
						t[4 * u + 2] = *(const double *)(Gb + q[u].z); t[4 * u + 3] = *(const double *)(Gb + q[u].w);
					}
				};
				uint4 q0[2], q1[2];
				double ta[8], tb[8];
				idx(e0, q0); gat(q0, ta); idx(e0 + 8, q1);
				for (int e = e0; e < e1; e += 16) {
					gat(q1, tb); idx(e + 16, q0);
#pragma unroll
					for (int u = 0; u < 8; u++) s += ta[u];               // (entries behind the list's end are the +0.0 slot: s + 0.0 is s)
					if (e + 8 >= e1) break;
					gat(q0, ta); idx(e + 24, q1);
#pragma unroll
					for (int u = 0; u < 8; u++) s += tb[u];
				}
				nxt[h] = s * scale;
			}
		__syncthreads();
		stamp(4);
		if (verdict_s) { stopped = 1; break; }                        // iteration iter - 1 was the last: its frequencies are what this one read
	}
	__syncthreads();
	const double *res = stopped ? cur : nxt;
	for (int h = tid; h < n_hap; h += EM_THREADS) V.out_freq[(size_t)c * n_hap + h] = res[h];
	if (tid == 0 && V.stamps) {
		tk[5] = (long long)wall_clock64() - t_first;
		for (int k = 0; k < 6; k++) V.stamps[(size_t)c * 6 + k] = tk[k];
	}
	if (tid == 0) {
		V.status[c] = verdict_s == 2 ? 2 : 1;                         // (500 iterations without convergence end the host's loop too)
		V.status[V.n_cand + c] = stopped ? iter - 1 : iter;
	}
}

struct EmState {
	int device = -1;
	void *d = nullptr, *h = nullptr;        // one device arena, one pinned staging area (upload, then download behind it)
	size_t cap_d = 0, cap_h = 0;
};
thread_local EmState g_em;          // (per host thread, like the build state: concurrent trainers each have their own arena and stream)
thread_local char g_em_msg[300];

[[noreturn]] void em_throw(const char *what, hipError_t e)
{
	snprintf(g_em_msg, sizeof(g_em_msg), "HIBAG HIP trainer: %s: %s", what, hipGetErrorString(e));
	throw (const char *)g_em_msg;
}
#define EM_OK(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) em_throw(what, e_); } while (0)

double em_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

// The fused launch of the EM fits of n growth steps (hibag_combine.h): grid = their candidates back to back, dynamic LDS = the
// largest step's layout.
void em_launch(const HibagOp *const ops[], int n, hipStream_t st)
{
	static std::mutex lds_m;
	static bool lds_set[64] = {};
	int dev = 0;
	(void)hipGetDevice(&dev);
	{
		std::lock_guard<std::mutex> lk(lds_m);                        // k_em_fit may use EM_LDS_BYTES of dynamic LDS: once per device
		if (dev >= 0 && dev < 64 && !lds_set[dev]) {
			(void)hipFuncSetAttribute((const void *)k_em_fit, hipFuncAttributeMaxDynamicSharedMemorySize, EM_LDS_BYTES);
			lds_set[dev] = true;
		}
	}
	HibagMulti<EmView> M;
	M.n = n;
	size_t lds = 0;
	int at = 0;
	for (int j = 0; j < n; j++) {
		const EmView &V = *(const EmView *)ops[j]->view;
		M.v[j] = V;
		M.first[j] = at;
		at += V.n_cand;
		lds = std::max(lds, EmLayout(V.n_ib, V.n_pair, V.n_hap, V.pos != nullptr).total);
	}
	for (int j = n; j <= HIBAG_COMBINE_MAX; j++) M.first[j] = at;
	hipLaunchKernelGGL(k_em_fit, dim3(at), dim3(EM_THREADS), lds, st, M);
}
const bool g_em_registered = (hibag_combine_register(HIBAG_OP_EM, em_launch), true);

} // namespace

thread_local double g_em_prof[3] = {0, 0, 0};            // seconds in hibag_em_fit_batch: staging the upload, copy + kernel + copy back (waited for), total

void hibag_em_release()
{
	if (g_em.d) (void)hipFree(g_em.d);
	if (g_em.h) (void)hipHostFree(g_em.h);
	g_em = EmState();
}

// Fits `n_cand` candidates on the current device; see hibag_em.h.
void hibag_em_fit_batch(const HibagEmPairs &P, const int8_t *const geno[], const double afreq[], int n_cand, double *out_freq, int *status, int *iters)
{
	if (n_cand <= 0) return;
	const double t_in = em_now();
	int dev = 0;
	EM_OK(hipGetDevice(&dev), "hipGetDevice");
	if (g_em.device != dev) { hibag_em_release(); g_em.device = dev; }
	const size_t np = (size_t)P.n_pair, nib = (size_t)P.n_ib, nh = (size_t)P.n_hap, nc = (size_t)n_cand;
	// (the lists' own copy of G where the step leaves LDS room for it, else the gather through offsets)
	const bool staged = 2 * np + 4 * nh <= 65535 && EmLayout(P.n_ib, P.n_pair, P.n_hap, true).total + 64 <= EM_LDS_BYTES;
	// upload area (one copy), then the results (one copy back)
	size_t o = 0;
	auto take = [&](size_t bytes) { const size_t at = o; o = (o + std::max<size_t>(bytes, 8) + 63) & ~(size_t)63; return at; };
	const size_t o_pw = take(np * 4), o_off = take((nib + 1) * 4), o_boot = take(nib * 4), o_hoff = take((nh + 1) * 4),
		o_hent = take((2 * np + 4 * nh) * 4), o_cur = take(nh / 2 * 8), o_geno = take(nc * nib), o_af = take(nc * 8);
	const size_t up_bytes = o;
	static const bool stamps = getenv("HIBAG_EM_STAMPS") != nullptr;
	const size_t o_out = take(nc * nh * 8), o_stat = take(2 * nc * 4), o_stamp = take(stamps ? nc * 6 * 8 : 0);
	const size_t down_bytes = o - o_out;
	if (o > g_em.cap_d) {
		if (g_em.d) (void)hipFree(g_em.d);            // (nothing of this thread's is in flight: every operation is waited for)
		g_em.d = nullptr; g_em.cap_d = 0;
		EM_OK(hipMalloc(&g_em.d, o + o / 2), "hipMalloc(EM arena)");
		g_em.cap_d = o + o / 2;
	}
	if (o > g_em.cap_h) {
		if (g_em.h) (void)hipHostFree(g_em.h);
		g_em.h = nullptr; g_em.cap_h = 0;
		EM_OK(hipHostMalloc(&g_em.h, o + o / 2, hipHostMallocDefault), "hipHostMalloc(EM staging)");
		g_em.cap_h = o + o / 2;
	}
	char *h = (char *)g_em.h, *d = (char *)g_em.d;
	size_t n_ent = 0;
	{
		uint32_t *pw = (uint32_t *)(h + o_pw);
		for (size_t j = 0; j < np; j++) pw[j] = (uint32_t)P.h1[j] | ((uint32_t)P.h2[j] << 14);
		// the transposed lists as byte offsets into G, each padded to a multiple of four entries with the +0.0 slot behind G --
		// or, staged, their inverse: the two entries of every pair (a homozygous pair's are neighbours in one list)
		uint32_t *he = (uint32_t *)(h + o_hent);
		uint16_t *ps16 = (uint16_t *)(h + o_hent);
		int *ho = (int *)(h + o_hoff);
		size_t e = 0;
		static thread_local std::vector<uint8_t> seen;
		if (staged) seen.assign(np, 0);
		for (size_t q = 0; q < nh; q++) {
			ho[q] = (int)e;
			if (staged)
				for (int k = P.hoff[q]; k < P.hoff[q + 1]; k++) { const int j = P.hent[k]; ps16[2 * (size_t)j + seen[j]++] = (uint16_t)e++; }
			else {
				for (int k = P.hoff[q]; k < P.hoff[q + 1]; k++) he[e++] = (uint32_t)P.hent[k] * 8u;
				while (e & 3) he[e++] = (uint32_t)np * 8u;
			}
			e = (e + 3) & ~(size_t)3;
		}
		ho[nh] = (int)e;
		n_ent = e;
	}
	memcpy(h + o_off, P.off, (nib + 1) * 4); memcpy(h + o_boot, P.boot, nib * 4);
	memcpy(h + o_cur, P.cur_freq, nh / 2 * 8);
	for (size_t c = 0; c < nc; c++) memcpy(h + o_geno + c * nib, geno[c], nib);
	memcpy(h + o_af, afreq, nc * 8);
	const double t_up = em_now();
	EmView V;
	V.n_cand = n_cand;
	V.n_ib = P.n_ib; V.n_pair = P.n_pair; V.n_hap = P.n_hap; V.n_samp_total = P.n_samp_total;
	V.pw = (const uint32_t *)(d + o_pw); V.off = (const int *)(d + o_off); V.boot = (const int *)(d + o_boot);
	V.hoff = (const int *)(d + o_hoff); V.hent = staged ? nullptr : (const uint32_t *)(d + o_hent); V.pos = staged ? (const uint16_t *)(d + o_hent) : nullptr; V.n_ent = (int)n_ent; V.cur_freq = (const double *)(d + o_cur);
	V.geno = (const int8_t *)(d + o_geno); V.afreq = (const double *)(d + o_af);
	V.out_freq = (double *)(d + o_out); V.status = (int *)(d + o_stat);
	V.stamps = stamps ? (long long *)(d + o_stamp) : nullptr;
	// one operation: the step's upload, its candidates' workgroups (alone, or fused with the other trainers' of the moment:
	// hibag_combine.h), the results back -- returns when they are in the staging area
	HibagOp op;
	op.kind = HIBAG_OP_EM;
	op.view = &V;
	op.up.push_back(HibagCopy{d, h, up_bytes});
	op.down.push_back(HibagCopy{h + o_out, d + o_out, down_bytes});
	hibag_combine_run(op);
	const double t_done = em_now();
	memcpy(out_freq, h + o_out, nc * nh * 8);
	const int *st = (const int *)(h + o_stat);
	for (size_t c = 0; c < nc; c++) { status[c] = st[c]; if (iters) iters[c] = st[nc + c]; }
	if (stamps) {
		// per iteration of the step's slowest candidate: microseconds in the phases (100 MHz ticks)
		static thread_local double acc[7];
		static thread_local long n_steps;
		size_t w = 0;
		for (size_t c = 1; c < nc; c++) if (st[nc + c] > st[nc + w]) w = c;
		const long long *tk = (const long long *)(h + o_stamp) + w * 6;
		const double it = std::max(1, st[nc + w] + 1);
		for (int k = 0; k < 6; k++) acc[k] += 0.01 * tk[k] / (k == 0 || k == 5 ? 1.0 : it);
		acc[6] += it;
		if (++n_steps % 200 == 0)
			fprintf(stderr, "[hibag em stamps] slowest candidate of a step, mean of %ld steps: set-up %.2f us, per iteration A %.2f B %.2f B' %.2f C %.2f us, %.1f iterations, whole %.1f us\n",
				n_steps, acc[0] / n_steps, acc[1] / n_steps, acc[2] / n_steps, acc[3] / n_steps, acc[4] / n_steps, acc[6] / n_steps, acc[5] / n_steps);
	}
	g_em_prof[0] += t_up - t_in; g_em_prof[1] += t_done - t_up; g_em_prof[2] += em_now() - t_in;
}

// does a growth step of these dimensions fit the device kernel (its LDS, 16-bit indices)?
bool hibag_em_fits(int n_ib, int n_pair, int n_hap)
{
	return n_hap <= 16384 && n_pair <= 65535 && n_ib > 0 && n_ib <= 65535 && EmLayout(n_ib, n_pair, n_hap, false).total + 64 <= EM_LDS_BYTES;
}

