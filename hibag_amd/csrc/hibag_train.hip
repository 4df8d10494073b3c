// hibag_train.hip -- host driver of hlaAttrBagging()'s native side:
// CAttrBag_Model::BuildClassifiers (src/LibHLA.cpp:2268-2305) with
// NewClassifierBootstrap (:2220-2245), CVariableSelection::Search (:1981-2122) and
// CAlg_EM (:1000-1255), as reached from HIBAG_Training + HIBAG_NewClassifiers
// (src/HIBAG.cpp:516-634).
//
// Division of labour, as in the reference when a GPU plugin is installed: the
// haplotype-pair kernels run on the device -- candidate pair lists
// (build_haplomatch), out-of-bag calls (build_acc_oob) and in-bag posteriors
// (build_acc_ib), hibag_build.hip -- while bootstrap draws, the EM iterations
// (tiny, strictly ordered sums over a few pairs per sample) and the selection
// logic stay on the host.  Unlike the reference's GPU branch the pair lists are
// put into the order of its CPU branch (_PrepHaploMatch_def, :1569-1637), so a
// model trained here is bit-identical to one trained by the reference on the CPU
// from the same random stream (tests/test_hip_training.py reproduces the
// reference's inst/extdata/OutOfBag.RData).
//
// There is no CPU scoring path in this file: without a device the build entries
// throw and the C ABI returns HIBAG_HIP_ENODEV.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <thread>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sched.h>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/hibag_hip.h"
#include "hibag_plugin.h"
#include "hibag_pool.h"
#include "hibag_em.h"
#include "hibag_combine.h"

int hibag_fail(int code, const char *fmt, ...);       // hibag_api.hip: sets the thread's last error
int hibag_selected_device();                          // hibag_api.hip: the thread's hibag_hip_set_device() choice
extern thread_local double g_batch_prof[6];           // hibag_build.hip
extern thread_local double g_em_prof[3];              // hibag_em.hip

namespace {

// src/LibHLA.cpp:98-116
const int EM_MAX_ITER = 500;
const double EM_INIT_VAL_FRAC = 0.001;
const double MIN_RARE_FREQ = 1e-5;                    // inst/include/LibHLA_ext.h:230
const double FRACTION_HAPLO = 1.0 / 10;
const double STOP_RELTOL_LOGLIK_ADDSNP = 0.001;
const double PRUNE_RELTOL_LOGLIK = 0.1;
const int MAX_SNP = HIBAG_HIP_MAX_SNP_IN_CLASSIFIER;

// R's default generator (Mersenne-Twister + the scrambling of set.seed(), R src/main/RNG.c),
// for hosts that are not R; an R binding hands in R's own unif_rand instead.
struct RMersenne {
	uint32_t mt[624];
	int mti = 625;
	void set_seed(uint32_t seed)
	{
		for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
		for (int j = 0; j < 625; j++) {
			seed = 69069u * seed + 1u;
			if (j > 0) mt[j - 1] = seed;
		}
		mti = 624;
	}
	double unif()
	{
		const int N = 624, M = 397;
		uint32_t y;
		if (mti >= N) {
			if (mti == N + 1) set_seed(4357);          // never seeded
			int kk;
			for (kk = 0; kk < N - M; kk++) {
				y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
				mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			}
			for (; kk < N - 1; kk++) {
				y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
				mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			}
			y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
			mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			mti = 0;
		}
		y = mt[mti++];
		y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
		const double v = (double)y * 2.3283064365386963e-10, eps = 2.328306437080797e-10;
		if (v <= 0.0) return 0.5 * eps;
		if (1.0 - v <= 0.0) return 1.0 - 0.5 * eps;
		return v;
	}
};

// CHaplotypeList (src/LibHLA.h:85-140): haplotypes grouped by allele, in the layout the
// build entries take (THaplotype) plus the EM's previous frequency.
struct HapList {
	std::vector<PluginHaplotype> list;
	std::vector<double> old_freq;
	std::vector<size_t> len;          // [n_hla]
	int n_snp = 0;
};

struct HapPair { int h1, h2; bool flag; double gfreq; };                // CAlg_EM::THaploPair
// CAlg_EM::THaploPairList of every in-bag sample, flattened: sample i owns p[off[i] .. off[i+1])
struct PairSet {
	std::vector<HapPair> p;
	std::vector<int> off, boot, samp;
	size_t size() const { return boot.size(); }
};

struct OutClassifier {
	std::vector<int32_t> snpidx, samp_num, hla;
	std::vector<double> freq;
	std::vector<uint64_t> bits;       // [n_haplo][2]
	double acc = 0;
};

inline void set_allele(PluginHaplotype &h, int idx, int v)              // THaplotype::_SetAllele, :339-345
{
	const uint64_t bit = (uint64_t)1 << (idx & 63);
	uint64_t w = (uint64_t)h.packed[idx >> 6];
	w = v ? (w | bit) : (w & ~bit);
	h.packed[idx >> 6] = (int64_t)w;
}
inline int get_allele(const PluginHaplotype &h, int idx) { return (int)(((uint64_t)h.packed[idx >> 6] >> (idx & 63)) & 1); }

inline void geno_set(PluginGenotype &g, int idx, int val)               // TGenotype::_SetSNP, :609-622
{
	const uint64_t bit = (uint64_t)1 << (idx & 63);
	const int w = idx >> 6;
	const bool b1 = val == 1 || val == 2, b2 = !(val == 0 || val == 1);
	uint64_t s1 = (uint64_t)g.snp1[w], s2 = (uint64_t)g.snp2[w];
	s1 = b1 ? (s1 | bit) : (s1 & ~bit);
	s2 = b2 ? (s2 | bit) : (s2 & ~bit);
	g.snp1[w] = (int64_t)s1; g.snp2[w] = (int64_t)s2;
}

// CSamplingWithoutReplace, :930-993
struct Sampling {
	std::vector<int> a;
	int m_try = 0;
	void init(int m_total) { a.resize(m_total); for (int i = 0; i < m_total; i++) a[i] = i; m_try = 0; }
	int &at(int idx) { return a[a.size() - m_try + idx]; }
	void remove(int idx) { a.erase(a.begin() + (a.size() - m_try + idx)); }
	void remove_selection() { a.resize(a.size() - m_try); }
	void remove_flag()
	{
		const int n = (int)a.size();
		for (int i = n - 1; i >= n - m_try; i--)
			if (a[i] < 0) a.erase(a.begin() + i);
	}
};

} // namespace

struct hibag_hip_trainer {
	int device = 0;                             // HIP device of this trainer (hibag_hip_set_device at creation)
	int n_snp = 0, n_samp = 0, n_hla = 0;
	std::vector<int32_t> geno, h1, h2;          // geno [n_samp][n_snp]
	std::vector<int32_t> geno_t;                // the same SNP-major, [n_snp][n_samp]: a candidate SNP's genotypes are one contiguous row
	                                            // (the search reads 18 such columns per growth step, twice: out of the sample-major
	                                            // matrix every read was a cache miss)
	RMersenne rng;
	double (*unif_fn)(void *) = nullptr;
	void *unif_ctx = nullptr;
	std::vector<OutClassifier> out;
	std::mutex lock;

	// CVariableSelection state
	std::vector<PluginGenotype> g;
	int g_nsnp = 0;
	std::vector<int> inbag, outbag;
	PairSet pl;
	int n_threads = 1;                          // host threads that fit candidate SNPs concurrently
	int em_mode = 0;                            // where the EM fits run: 0 = by the thread count, 1 = host threads, 2 = device (hibag_em.hip)
	bool shared = false;                        // runs beside other trainers of the process: its device work goes through the combiners (hibag_combine.h)
	std::unique_ptr<Pool> pool;                 // n_threads - 1 helpers, created by the first training call

	double unif() { return unif_fn ? unif_fn(unif_ctx) : rng.unif(); }
	int random_num(int n)                                               // :120-126
	{
		int v = (int)(n * unif());
		if (v >= n) v = n - 1;
		return v;
	}
};

namespace {

typedef hibag_hip_trainer T;

// wall-clock split of a training call, printed when HIBAG_TRAIN_PROFILE is set
struct Profile {
	double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // the calling thread's CPU time in the same phases (what the phase costs the host, waits excluded)
	static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
	static double cpu() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
};
thread_local Profile g_prof;                          // (the calling thread's: trainers may run side by side)
thread_local long long g_em_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // candidates fitted on the device, of them handed back to the host, device iterations; growth steps' pairs (sum, max), longest transposed list
struct Tick {
	int k; double t0, c0;
	explicit Tick(int k_) : k(k_), t0(Profile::now()), c0(Profile::cpu()) {}
	~Tick() { g_prof.t[k] += Profile::now() - t0; g_prof.c[k] += Profile::cpu() - c0; }
};

void select(T &t, Sampling &s, int m_try)                               // RandomSelect, :949-962
{
	const int n = (int)s.a.size();
	if (m_try > n) m_try = n;
	if (m_try < n)
		for (int i = 0; i < m_try; i++) {
			const int I = t.random_num(n - i);
			std::swap(s.a[I], s.a[n - i - 1]);
		}
	s.m_try = m_try;
}

void init_selection(T &t, const std::vector<int> &boot)                 // InitSelection, :1843-1878
{
	t.inbag.clear(); t.outbag.clear();
	for (int i = 0; i < t.n_samp; i++) {
		PluginGenotype &g = t.g[i];
		g.bootstrap_count = boot[i];
		g.hla1 = t.h1[i]; g.hla2 = t.h2[i];
		if (g.hla2 < g.hla1) std::swap(g.hla1, g.hla2);
		(boot[i] > 0 ? t.inbag : t.outbag).push_back(i);
		g.snp1[0] = g.snp1[1] = 0;                                      // SetAllMissing, :883-891
		g.snp2[0] = g.snp2[1] = -1;
	}
	t.g_nsnp = 0;
}

void init_haplotype(T &t, HapList &h)                                   // _InitHaplotype, :1880-1911
{
	std::vector<int> tmp(t.n_hla, 0);
	int sum = 0;
	for (int k : t.inbag) {
		const PluginGenotype &g = t.g[k];
		tmp[g.hla1] += g.bootstrap_count; tmp[g.hla2] += g.bootstrap_count;
		sum += g.bootstrap_count;
	}
	h.len.assign(t.n_hla, 0);
	h.list.clear();
	h.n_snp = 0;
	const double scale = 0.5 / sum;
	for (int i = 0; i < t.n_hla; i++)
		if (tmp[i] > 0) {
			h.len[i] = 1;
			PluginHaplotype p{};
			p.freq = tmp[i] * scale;
			h.list.push_back(p);
		}
}

void set_aux(HapList &h)                                                // SetHaploAux_GPU, :565-578
{
	size_t i = 0;
	for (size_t a = 0; a < h.len.size(); a++)
		for (size_t m = h.len[a]; m > 0; m--, i++) {
			h.list[i].aux.freq_f32 = (float)h.list[i].freq;
			h.list[i].aux.hla_allele = (int)a;
		}
}

void add_snp(T &t, int snp)                                             // CGenotypeList::AddSNP, :860-874
{
	for (int i = 0; i < t.n_samp; i++) geno_set(t.g[i], t.g_nsnp, t.geno_t[(size_t)snp * t.n_samp + i]);
	t.g_nsnp++;
}

// CAlg_EM::PrepareHaplotypes (:1002-1125): the device lists, per in-bag sample, the pairs of
// CURRENT haplotypes of its two true alleles at distance 0 (or at the minimum distance);
// each becomes the 4 (3 on the diagonal) pairs of its doubled copies, ordered like the
// reference's CPU branch: ascending (first, second) index in the doubled list.
void prepare_haplotypes(T &t, const HapList &cur, HapList &next)
{
	next.n_snp = cur.n_snp + 1;                                         // DoubleHaplos, :416-442
	next.list.resize(cur.list.size() * 2);
	next.old_freq.assign(next.list.size(), 0.0);
	for (size_t i = 0; i < cur.list.size(); i++) {
		next.list[2 * i] = cur.list[i]; set_allele(next.list[2 * i], cur.n_snp, 0);
		next.list[2 * i + 1] = cur.list[i]; set_allele(next.list[2 * i + 1], cur.n_snp, 1);
	}
	next.len.resize(t.n_hla);
	std::vector<size_t> start(t.n_hla, 0);
	for (int h = 0, st = 0; h < t.n_hla; h++) {
		if (cur.len[h] > 65535) throw "There are too many HLA allele-specific haplotypes (# > 65535).";
		next.len[h] = cur.len[h] * 2;
		start[h] = st; st += (int)cur.len[h];
	}
	const size_t n_ib = t.inbag.size();
	PairSet &ps = t.pl;
	ps.p.clear();
	ps.boot.resize(n_ib); ps.samp.resize(n_ib);
	ps.off.assign(n_ib + 1, 0);
	for (size_t i = 0; i < n_ib; i++) { ps.boot[i] = t.g[t.inbag[i]].bootstrap_count; ps.samp[i] = t.inbag[i]; }

	size_t n_buf = 0;
	uint32_t *buf = hibag_build_haplomatch(cur.list.data(), cur.len.data(), cur.n_snp, t.g.data(), n_buf);
	// the device lists the pairs sample by sample (ascending in-bag index)
	size_t k_prev = 0;
	bool first = true;
	auto close_range = [&](size_t k) {                                   // [off[k], end) is complete: CPU-branch order
		std::sort(ps.p.begin() + ps.off[k], ps.p.end(), [](const HapPair &x, const HapPair &y) {
			return x.h1 != y.h1 ? x.h1 < y.h1 : x.h2 < y.h2; });
	};
	if (buf) {
		uint32_t n = buf[0] >> 1;
		for (const uint32_t *p = buf + 1; n > 0; n--, p += 2) {
			const size_t k = p[0];
			if (k >= n_ib || (!first && k < k_prev)) { free(buf); throw "build_haplomatch returned an invalid sample index"; }
			if (first || k != k_prev) {
				if (!first) close_range(k_prev);
				for (size_t i = first ? 0 : k_prev + 1; i <= k; i++) ps.off[i] = (int)ps.p.size();
				k_prev = k; first = false;
			}
			const PluginGenotype &g = t.g[t.inbag[k]];
			const int a = (int)(2 * (start[g.hla1] + (p[1] & 0xFFFF))), b = (int)(2 * (start[g.hla2] + (p[1] >> 16)));
			ps.p.push_back(HapPair{a, b, false, 0.0});
			ps.p.push_back(HapPair{a, b + 1, false, 0.0});
			if (a + 1 <= b) ps.p.push_back(HapPair{a + 1, b, false, 0.0});
			ps.p.push_back(HapPair{a + 1, b + 1, false, 0.0});
		}
		free(buf);
	}
	if (!first) close_range(k_prev);
	for (size_t i = first ? 0 : k_prev + 1; i <= n_ib; i++) ps.off[i] = (int)ps.p.size();
	for (size_t i = 0; i < n_ib; i++)
		if (ps.off[i + 1] == ps.off[i]) throw "PairList should not be empty in PrepareHaplotypes().";   // :1070-1071
}

// PrepareNewSNP, :1127-1183.  `pl` / `next` are the caller's working copies: candidates of one
// growth step are fitted concurrently, each on its own copy.
// One candidate SNP's fit: CAlg_EM::PrepareNewSNP (DoubleHaplosInitFreq, :444-459, and the pair flags, :1133-1183) followed by
// ExpectationMaximization (:1185-1255) on the in-bag samples' haplotype pairs `t.pl`, which index the doubled list `next`.
// The reference flags the pairs compatible with the new SNP's genotype and walks all pairs in every iteration; here the
// compatible ones are copied once into compact arrays (the indices, then the pair frequencies) and the frequencies live in
// two plain vectors -- the same additions and multiplications in the same order (per sample its pairs in list order, per
// haplotype the pairs in sample order), without a copy of the whole pair set per candidate, without a branch per pair and
// iteration, and with a third of the memory traffic.  Returns false where the reference skips the SNP (monomorphic in the bag).
struct FitScratch {
	std::vector<int> h1, h2, off;
	std::vector<double> gfreq, old_freq, new_freq, log_buf;
};
bool fit_new_snp(const T &t, int snp, const HapList &cur, HapList &next, FitScratch &S)
{
	int allele_cnt = 0, valid_cnt = 0;
	const int32_t *const gcol = &t.geno_t[(size_t)snp * t.n_samp];        // the SNP's genotypes, contiguous
	for (int i : t.inbag) {
		const int dup = t.g[i].bootstrap_count;
		const int g = gcol[i];
		if (0 <= g && g <= 2) { allele_cnt += g * dup; valid_cnt += 2 * dup; }
	}
	if (allele_cnt == 0 || allele_cnt == valid_cnt) return false;
	const double afreq = (double)allele_cnt / valid_cnt, p0 = 1 - afreq, p1 = afreq;   // DoubleHaplosInitFreq, :444-459
	const size_t nh = next.list.size();
	S.new_freq.resize(nh); S.old_freq.resize(nh);
	for (size_t i = 0; i < cur.list.size(); i++) {
		S.new_freq[2 * i] = p0 * cur.list[i].freq + EM_INIT_VAL_FRAC;
		S.new_freq[2 * i + 1] = p1 * cur.list[i].freq + EM_INIT_VAL_FRAC;
	}
	// the pairs compatible with the sample's genotype at the new SNP (every pair where it is missing)
	const PairSet &pls = t.pl;
	const size_t num = pls.size();
	const int idx_new = next.n_snp - 1;
	S.off.resize(num + 1);
	S.h1.clear(); S.h2.clear();
	for (size_t i = 0; i < num; i++) {
		S.off[i] = (int)S.h1.size();
		const int geno = gcol[pls.samp[i]];
		const bool typed = 0 <= geno && geno <= 2;
		for (int j = pls.off[i]; j < pls.off[i + 1]; j++) {
			const HapPair &p = pls.p[j];
			if (!typed || get_allele(next.list[p.h1], idx_new) + get_allele(next.list[p.h2], idx_new) == geno) {
				S.h1.push_back(p.h1); S.h2.push_back(p.h2);
			}
		}
	}
	S.off[num] = (int)S.h1.size();
	S.gfreq.resize(S.h1.size());
	if (S.log_buf.size() < num) S.log_buf.resize(num);
	const int *const H1 = S.h1.data(), *const H2 = S.h2.data();
	double *const G = S.gfreq.data(), *const oldf = S.old_freq.data(), *const newf = S.new_freq.data();

	const int total = t.n_samp;
	const double em_reltol = std::sqrt(DBL_EPSILON);                    // :102
	double conv_tol = 0, loglik = -1e+30;
	for (int iter = 0; iter <= EM_MAX_ITER; iter++) {
		const double old_loglik = loglik;
		for (size_t i = 0; i < nh; i++) { oldf[i] = newf[i]; newf[i] = 0; }
		for (size_t i = 0; i < num; i++) {
			const int j0 = S.off[i], j1 = S.off[i + 1];
			double psum = 0;
			for (int j = j0; j < j1; j++) {
				G[j] = (H1[j] != H2[j]) ? (2 * oldf[H1[j]] * oldf[H2[j]]) : (oldf[H1[j]] * oldf[H2[j]]);
				psum += G[j];
			}
			S.log_buf[i] = pls.boot[i] * std::log(psum);
			psum = pls.boot[i] / psum;
			for (int j = j0; j < j1; j++) G[j] *= psum;
		}
		loglik = 0;
		for (size_t i = 0; i < num; i++) {
			loglik += S.log_buf[i];
			for (int j = S.off[i]; j < S.off[i + 1]; j++) { newf[H1[j]] += G[j]; newf[H2[j]] += G[j]; }
		}
		const double scale = 0.5 / total;
		for (size_t i = 0; i < nh; i++) newf[i] *= scale;
		if (iter > 0) {
			if (std::fabs(loglik - old_loglik) <= conv_tol) break;
		} else {
			conv_tol = em_reltol * (std::fabs(loglik) + em_reltol);
			if (conv_tol < 0) conv_tol = 0;
		}
	}
	for (size_t i = 0; i < nh; i++) next.list[i].freq = newf[i];
	return true;
}

void erase_double_haplos(const HapList &in, double rare_prob, HapList &out)   // EraseDoubleHaplos, :461-515
{
	out.n_snp = in.n_snp;
	out.list.clear();
	out.len.assign(in.len.size(), 0);
	const PluginHaplotype *p = in.list.data();
	double sum = 0;
	for (size_t h = 0; h < in.len.size(); h++) {
		size_t num = 0;
		for (size_t n = in.len[h]; n > 0; n -= 2, p += 2) {
			const double sumfreq = p[0].freq + p[1].freq;
			if (p[0].freq < rare_prob || p[1].freq < rare_prob) {
				if (sumfreq >= MIN_RARE_FREQ) {
					out.list.push_back(p[0].freq >= p[1].freq ? p[0] : p[1]);
					out.list.back().freq = sumfreq;
					sum += sumfreq;
					num++;
				}
			} else {
				out.list.push_back(p[0]); out.list.push_back(p[1]);
				sum += sumfreq;
				num += 2;
			}
		}
		out.len[h] = num;
	}
	const double scale = 1 / sum;
	for (PluginHaplotype &h : out.list) h.freq *= scale;
}

// Host threads a trainer fits candidate SNPs on: the CPUs this process may use (affinity mask capped by the cgroup quota)
// divided by the ranks that share the host -- one process per GPU under torch.distributed.run, which exports
// LOCAL_WORLD_SIZE: eight trainers that each started one thread per CPU would oversubscribe the host eight times.
// HIBAG_TRAIN_THREADS overrides; hibag_hip_trainer_set_threads() sets it per trainer.
int usable_threads()
{
	if (const char *e = getenv("HIBAG_TRAIN_THREADS")) return std::max(1, std::min(atoi(e), 256));
	int n = (int)std::thread::hardware_concurrency();
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // cgroup v2
		char quota[32];
		long period = 0;
		if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
			n = std::min(n, std::max(1, (int)((atol(quota) + period / 2) / period)));
		fclose(f);
	} else {                                                                  // cgroup v1
		long quota = -1, period = 0;
		if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(q, "%ld", &quota) != 1) quota = -1; fclose(q); }
		if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(q, "%ld", &period) != 1) period = 0; fclose(q); }
		if (quota > 0 && period > 0) n = std::min(n, std::max(1, (int)((quota + period / 2) / period)));
	}
	int ranks = 1;
	if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
	return std::max(1, std::min(n / ranks, 64));
}

const char *date_text()
{
	static char buf[64];
	const time_t now = time(nullptr);
	strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", localtime(&now));
	return buf;
}

// CVariableSelection::Search, :1981-2122
void search(T &t, Sampling &vs, OutClassifier &o, int mtry, bool prune, bool verbose_detail)
{
	const double rare_prob = std::max(FRACTION_HAPLO / (2 * t.n_samp), MIN_RARE_FREQ);
	HapList out_haplo, next, minh;
	{ Tick tk(6); init_haplotype(t, out_haplo); }
	o.snpidx.clear();
	const int num_oob = (int)t.outbag.size();
	int global_max_acc = 0;
	double global_min_loss = 1e+30;

	// (the candidate pair lists depend on the current haplotype list and the committed SNPs alone: a growth step that accepted
	// no candidate leaves both as they were, and the step after it starts from the same lists -- half of all steps at config 5's
	// shape; the reference recomputes them, src/LibHLA.cpp:2001, with the same result)
	bool lists_current = false;
	// (the step's containers live across steps: their capacity is reused, a growth step does no allocation in steady state)
	std::vector<HapList> cand;
	std::vector<char> valid, on_host;
	std::vector<int> accv;
	std::vector<double> lossv;
	struct EmScratch {
		std::vector<int> ph1, ph2, hoff, hent, at, which, status, iters;
		std::vector<double> curf, af, freq;
		std::vector<int8_t> gcol;                                       // [candidates][n_ib]
		std::vector<const int8_t *> gp;
		HapList nx;
	} E;
	while (!vs.a.empty() && (int)o.snpidx.size() < MAX_SNP) {
		if (!lists_current) { Tick tk(0); prepare_haplotypes(t, out_haplo, next); lists_current = true; }
		int max_acc = global_max_acc, min_i = -1;
		double min_loss = global_min_loss;
		{ Tick tk(4); select(t, vs, mtry); }

		// The candidates of a step are independent until they are compared (each starts from
		// OutHaplo, :2020-2025): fit them concurrently on the host, score them in one device
		// pass, then apply the reference's sequential comparison to the results.
		const int m = vs.m_try;
		g_prof.t[7] += 1;                                               // growth steps
		cand.resize(m);
		valid.assign(m, 0);
		accv.assign(m, 0);
		lossv.assign(m, 0.0);
		// (Fitting in two halves, the first being scored while the second is fitted -- hibag_build_eval_launch / _collect have
		// two slots for that -- was measured and dropped: the fits of a step differ so much in length that each half lasts
		// about as long as the whole, 0.58 ms against 0.37 per step.  Sending the first 16 candidates off as soon as they are
		// fitted while helper threads fit the last two was measured too: no gain, the second launch and read-back cost what the
		// overlap saves; profiles/r03_cfg5_notes.txt.)
		const int half = m;
		struct Part { std::vector<HibagBuildCandidate> bc; std::vector<int> which; } part[2];
		// The fits: on the device (hibag_em.hip, workgroup = candidate), except the candidates whose stopping test the device's
		// log() cannot decide the way the host's would -- a handful per model -- and where the rank has host threads to spare all of them:
		// those go to the host threads below, as every fit did until round 4.
		// (the device fit does not depend on the host's cores but is the slower one from about three host threads on:
		// hibag_em.hip; HIBAG_TRAIN_EM=host|device and hibag_hip_trainer_set_em_mode override the choice by thread count)
		static const int em_env = !getenv("HIBAG_TRAIN_EM") ? 0 : !strcmp(getenv("HIBAG_TRAIN_EM"), "host") ? 1 : !strcmp(getenv("HIBAG_TRAIN_EM"), "device") ? 2 : 0;
		const int em_mode = t.em_mode ? t.em_mode : em_env;
		const bool em_host = em_mode == 1 || (em_mode == 0 && t.n_threads > 2);
		on_host.assign(m, 1);
		if (!em_host && hibag_em_fits((int)t.pl.size(), (int)t.pl.p.size(), (int)next.list.size())) {
			Tick tk(1);
			const PairSet &pls = t.pl;
			const int n_ib = (int)pls.size(), n_pair = (int)pls.p.size(), n_hap = (int)next.list.size();
			std::vector<int> &ph1 = E.ph1, &ph2 = E.ph2, &hoff = E.hoff, &hent = E.hent;
			ph1.resize(n_pair); ph2.resize(n_pair); hoff.assign(n_hap + 1, 0); hent.resize(2 * (size_t)n_pair);
			for (int j = 0; j < n_pair; j++) { ph1[j] = pls.p[j].h1; ph2[j] = pls.p[j].h2; hoff[ph1[j] + 1]++; hoff[ph2[j] + 1]++; }
			for (int h = 0; h < n_hap; h++) hoff[h + 1] += hoff[h];
			{
				E.at.assign(hoff.begin(), hoff.end() - 1);
				for (int j = 0; j < n_pair; j++) { hent[E.at[ph1[j]]++] = j; hent[E.at[ph2[j]]++] = j; }      // (pair order; (h, h) twice)
			}
			g_em_stats[3] += n_pair; g_em_stats[4] = std::max<long long>(g_em_stats[4], n_pair);
			for (int h = 0; h < n_hap; h++) g_em_stats[5] = std::max<long long>(g_em_stats[5], hoff[h + 1] - hoff[h]);
			for (int k = 0; k < n_ib; k++) g_em_stats[7] = std::max<long long>(g_em_stats[7], pls.off[k + 1] - pls.off[k]);
			std::vector<double> &curf = E.curf;
			curf.resize(out_haplo.list.size());
			for (size_t i = 0; i < curf.size(); i++) curf[i] = out_haplo.list[i].freq;
			// the reference skips a SNP that is monomorphic in the bag (:1140-1143)
			std::vector<int> &which = E.which;
			std::vector<double> &af = E.af;
			which.clear(); af.clear();
			E.gcol.resize((size_t)m * n_ib);
			for (int i = 0; i < m; i++) {
				const int snp = vs.at(i);
				int allele_cnt = 0, valid_cnt = 0;
				int8_t *col = E.gcol.data() + which.size() * (size_t)n_ib;
				const int32_t *const gc = &t.geno_t[(size_t)snp * t.n_samp];
				for (int k = 0; k < n_ib; k++) {
					const int g = gc[pls.samp[k]];
					const bool typed = 0 <= g && g <= 2;
					col[k] = typed ? (int8_t)g : (int8_t)3;
					if (typed) { allele_cnt += g * pls.boot[k]; valid_cnt += 2 * pls.boot[k]; }
				}
				if (allele_cnt == 0 || allele_cnt == valid_cnt) { on_host[i] = 0; continue; }       // (not fitted at all: valid stays 0)
				which.push_back(i); af.push_back((double)allele_cnt / valid_cnt);
			}
			if (!which.empty()) {
				HibagEmPairs P{n_ib, n_pair, n_hap, t.n_samp, ph1.data(), ph2.data(), pls.off.data(), pls.boot.data(), hoff.data(), hent.data(), curf.data()};
				E.gp.clear();
				for (size_t w = 0; w < which.size(); w++) E.gp.push_back(E.gcol.data() + w * (size_t)n_ib);
				E.freq.resize((size_t)which.size() * n_hap);
				E.status.resize(which.size()); E.iters.resize(which.size());
				std::vector<double> &freq = E.freq;
				std::vector<int> &status = E.status, &iters = E.iters;
				hibag_em_fit_batch(P, E.gp.data(), af.data(), (int)which.size(), freq.data(), status.data(), iters.data());
				g_em_stats[6] += *std::max_element(iters.begin(), iters.end());
				for (size_t w = 0; w < which.size(); w++) {
					const int i = which[w];
					g_em_stats[0]++; g_em_stats[2] += iters[w];
					if (status[w] != 1) { g_em_stats[1]++; continue; }                                  // (stays on_host)
					HapList &nx = E.nx;
					nx = next;
					for (int h = 0; h < n_hap; h++) nx.list[h].freq = freq[w * n_hap + h];
					erase_double_haplos(nx, rare_prob, cand[i]);
					set_aux(cand[i]);                                   // _Init_EvalAcc, :1913-1929
					valid[i] = 1; on_host[i] = 0;
				}
			}
		}
		auto fit = [&](int lo, int hi) {
			Tick tk(1), tk_host(6);
			std::atomic<int> next_i(lo);
			auto work = [&]() {
				HapList nx;
				FitScratch scratch;
				for (int i; (i = next_i.fetch_add(1)) < hi;) {
					if (!on_host[i]) continue;
					nx = next;
					if (!fit_new_snp(t, vs.at(i), out_haplo, nx, scratch)) continue;
					erase_double_haplos(nx, rare_prob, cand[i]);
					set_aux(cand[i]);                                   // _Init_EvalAcc, :1913-1929
					valid[i] = 1;
				}
			};
			if (t.pool) t.pool->run(work); else work();
		};
		auto launch = [&](int slot, int lo, int hi) {
			Tick tk(2);
			Part &P = part[slot];
			for (int i = lo; i < hi; i++) if (valid[i]) P.which.push_back(i);
			// (a candidate's genotype column is a row of the SNP-major copy: nothing to gather)
			for (size_t j = 0; j < P.which.size(); j++)
				P.bc.push_back(HibagBuildCandidate{cand[P.which[j]].list.data(), (int)cand[P.which[j]].list.size(),
					&t.geno_t[(size_t)vs.at(P.which[j]) * t.n_samp], vs.at(P.which[j])});
			hibag_build_eval_launch(slot, t.g.data(), t.g_nsnp + 1, P.bc.data(), (int)P.bc.size());
		};
		int acc_floor = global_max_acc;
		auto collect = [&](int slot) {
			Tick tk(2);
			Part &P = part[slot];
			std::vector<int> a(P.bc.size());
			std::vector<double> l(P.bc.size());
			hibag_build_eval_collect(slot, &acc_floor, a.data(), l.data());
			for (size_t j = 0; j < P.which.size(); j++) { accv[P.which[j]] = a[j]; lossv[P.which[j]] = l[j]; }
		};
		fit(0, half);
		launch(0, 0, half);
		if (half < m) { fit(half, m); launch(1, half, m); }
		collect(0);
		if (half < m) collect(1);
		Tick tk3(3);
		for (int i = 0; i < m; i++) {                                   // :2018-2069
			if (!valid[i]) continue;
			const int acc = accv[i];
			const double loss = acc >= max_acc ? lossv[i] : 0;          // the in-bag loss is only looked at then (:2033-2034)
			if (acc > max_acc) { min_i = i; min_loss = loss; max_acc = acc; minh = cand[i]; }
			else if (acc == max_acc && loss < min_loss) { min_i = i; min_loss = loss; minh = cand[i]; }
			if (prune) {
				if (acc < global_max_acc) vs.at(i) = -1;
				else if (acc == global_max_acc && loss > global_min_loss * (1 + PRUNE_RELTOL_LOGLIK) && min_i != i) vs.at(i) = -1;
			}
		}
		bool sign = false;
		if (max_acc > global_max_acc) sign = true;
		else if (max_acc == global_max_acc && min_i >= 0)
			sign = min_loss >= STOP_RELTOL_LOGLIK_ADDSNP && min_loss < global_min_loss * (1 - STOP_RELTOL_LOGLIK_ADDSNP);
		if (sign) {
			global_max_acc = max_acc;
			global_min_loss = min_loss;
			out_haplo = minh;
			lists_current = false;
			o.snpidx.push_back(vs.at(min_i));
			add_snp(t, vs.at(min_i));
			if (prune) { vs.at(min_i) = -1; vs.remove_flag(); } else vs.remove(min_i);
			if (verbose_detail)
				printf("    %2d, SNP: %d, loss: %g, oob acc: %0.2f%%, # of haplo: %d\n", (int)o.snpidx.size(),
					o.snpidx.back() + 1, global_min_loss, double(global_max_acc) / num_oob * 50, (int)out_haplo.list.size());
		} else {
			vs.remove_selection();
			for (int i = 0; i < t.n_samp; i++) geno_set(t.g[i], t.g_nsnp, -1);    // SetMissing, :893-903
		}
	}
	o.acc = 0.5 * global_max_acc / num_oob;
	const size_t H = out_haplo.list.size();
	o.freq.resize(H); o.hla.resize(H); o.bits.resize(2 * H);
	size_t i = 0;
	for (size_t a = 0; a < out_haplo.len.size(); a++)
		for (size_t m = out_haplo.len[a]; m > 0; m--, i++) {
			o.freq[i] = out_haplo.list[i].freq;
			o.hla[i] = (int32_t)a;
			for (int w = 0; w < 2; w++) {                               // report defined bits only
				const int lo = 64 * w, k = (int)o.snpidx.size();
				const uint64_t mask = k >= lo + 64 ? ~(uint64_t)0 : (k <= lo ? 0 : (((uint64_t)1 << (k - lo)) - 1));
				o.bits[2 * i + w] = (uint64_t)out_haplo.list[i].packed[w] & mask;
			}
		}
}

// BuildClassifiers, :2268-2305
void build_classifiers(T &t, int nclassifier, int mtry, bool prune, bool verbose, bool verbose_detail)
{
	struct Scope {                                                      // try_final_train_gpu, :2256-2266
		Scope(int nh, int ns) { hibag_build_init(nh, ns); }
		~Scope() { hibag_build_done(); hibag_em_release(); }      // (both states are the calling thread's own: nothing of them outlives the call)
	} scope(t.n_hla, t.n_samp);
	hibag_build_set_genotypes(t.geno_t.data(), t.n_snp);             // (the candidates of a growth step then travel as SNP indices)
	Sampling vs;
	const int n = t.n_samp;
	for (int k = 0; k < nclassifier; k++) {
		vs.init(t.n_snp);
		std::vector<int> S(n);                                          // NewClassifierBootstrap, :2220-2245
		int n_unique;
		do {
			std::fill(S.begin(), S.end(), 0);
			n_unique = 0;
			for (int i = 0; i < n; i++) {
				const int j = t.random_num(n);
				if (S[j] == 0) n_unique++;
				S[j]++;
			}
		} while (n_unique >= n);
		if (verbose) {
			int n_oob = 0;
			for (int v : S) if (v == 0) n_oob++;
			printf("=== building individual classifier %d, out-of-bag (%d/%.1f%%) ===\n", (int)t.out.size() + 1, n_oob, 100.0 * n_oob / n);
		}
		hibag_build_set_bootstrap(S.data());
		init_selection(t, S);
		OutClassifier o;
		o.samp_num.assign(S.begin(), S.end());
		{ Tick tk(5); search(t, vs, o, mtry, prune, verbose_detail); }
		t.out.push_back(std::move(o));
		if (verbose) {
			const OutClassifier &c = t.out.back();
			printf("[%d] %s, oob acc: %0.2f%%, # of SNPs: %d, # of haplo: %d\n", (int)t.out.size(), date_text(),
				c.acc * 100, (int)c.snpidx.size(), (int)c.freq.size());
			fflush(stdout);
		}
	}
}

} // namespace

// ===========================================================================
// C ABI

extern "C" {

hibag_hip_trainer *hibag_hip_trainer_new(int n_snp, int n_samp, const int32_t *snp_geno, int n_hla,
	const int32_t *H1, const int32_t *H2)
{
	// messages of HIBAG_Training (src/HIBAG.cpp:516-535) and InitTraining (src/LibHLA.cpp:2196-2218)
	if (n_samp <= 0) { hibag_fail(HIBAG_HIP_EINVAL, "Invalid number of samples: %d.", n_samp); return nullptr; }
	if (n_snp <= 0) { hibag_fail(HIBAG_HIP_EINVAL, "Invalid number of SNPs: %d.", n_snp); return nullptr; }
	if (n_hla <= 0) { hibag_fail(HIBAG_HIP_EINVAL, "Invalid number of unique HLA alleles: %d.", n_hla); return nullptr; }
	if (!snp_geno || !H1 || !H2) { hibag_fail(HIBAG_HIP_EINVAL, "NULL argument"); return nullptr; }
	for (int i = 0; i < n_samp; i++) {
		if (H1[i] < 0 || H1[i] >= n_hla) { hibag_fail(HIBAG_HIP_EINVAL, "CAttrBag_Model::InitTraining, H1 error."); return nullptr; }
		if (H2[i] < 0 || H2[i] >= n_hla) { hibag_fail(HIBAG_HIP_EINVAL, "CAttrBag_Model::InitTraining, H2 error."); return nullptr; }
	}
	hibag_hip_trainer *t = new (std::nothrow) hibag_hip_trainer;
	if (!t) { hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	t->device = hibag_selected_device();
	t->n_snp = n_snp; t->n_samp = n_samp; t->n_hla = n_hla;
	t->geno.assign(snp_geno, snp_geno + (size_t)n_samp * n_snp);
	t->geno_t.resize((size_t)n_samp * n_snp);
	for (int i = 0; i < n_samp; i++)
		for (int j = 0; j < n_snp; j++) t->geno_t[(size_t)j * n_samp + i] = snp_geno[(size_t)i * n_snp + j];
	t->h1.assign(H1, H1 + n_samp); t->h2.assign(H2, H2 + n_samp);
	t->g.assign(n_samp, PluginGenotype{});
	t->n_threads = usable_threads();
	return t;
}

void hibag_hip_trainer_free(hibag_hip_trainer *t) { delete t; }

int hibag_hip_trainer_set_threads(hibag_hip_trainer *t, int n_threads)
{
	if (!t) return hibag_fail(HIBAG_HIP_EINVAL, "trainer is NULL");
	std::lock_guard<std::mutex> g(t->lock);
	const int n = n_threads > 0 ? std::min(n_threads, 256) : usable_threads();
	if (n != t->n_threads) { t->pool.reset(); t->n_threads = n; }      // (the helpers are started by the next training call)
	return 0;
}

int hibag_hip_trainer_threads(const hibag_hip_trainer *t) { return t ? t->n_threads : 0; }

int hibag_hip_trainer_set_em_mode(hibag_hip_trainer *t, int mode)
{
	if (!t) return hibag_fail(HIBAG_HIP_EINVAL, "trainer is NULL");
	if (mode < 0 || mode > 2) return hibag_fail(HIBAG_HIP_EINVAL, "EM mode must be 0 (automatic), 1 (host threads) or 2 (device)");
	std::lock_guard<std::mutex> g(t->lock);
	t->em_mode = mode;
	return 0;
}

int hibag_hip_trainer_set_shared(hibag_hip_trainer *t, int shared)
{
	if (!t) return hibag_fail(HIBAG_HIP_EINVAL, "trainer is NULL");
	std::lock_guard<std::mutex> g(t->lock);
	t->shared = shared != 0;
	return 0;
}

int hibag_hip_train_set_thread_budget(int n_threads)
{
	hibag_combine_set_budget(n_threads);
	return 0;
}

int hibag_hip_train_combine_stats(long long *launches, long long *ops, int reset)
{
	hibag_combine_stats(launches, ops, reset);
	return 0;
}

int hibag_hip_train_combine_times(double *out12, int reset)
{
	hibag_combine_times(out12, reset);
	return 0;
}

int hibag_hip_trainer_set_seed(hibag_hip_trainer *t, uint32_t seed)
{
	if (!t) return hibag_fail(HIBAG_HIP_EINVAL, "trainer is NULL");
	t->rng.set_seed(seed);
	t->unif_fn = nullptr;
	return 0;
}

int hibag_hip_trainer_set_rng(hibag_hip_trainer *t, double (*unif_rand)(void *), void *ctx)
{
	if (!t) return hibag_fail(HIBAG_HIP_EINVAL, "trainer is NULL");
	t->unif_fn = unif_rand; t->unif_ctx = ctx;
	return 0;
}

int hibag_hip_trainer_new_classifiers(hibag_hip_trainer *t, int nclassifier, int mtry, int prune, int verbose,
	int verbose_detail)
{
	if (!t) return hibag_fail(HIBAG_HIP_EINVAL, "trainer is NULL");
	if (nclassifier < 0 || mtry < 1) return hibag_fail(HIBAG_HIP_EINVAL, "invalid nclassifier / mtry");
	std::lock_guard<std::mutex> g(t->lock);
	// (the build entries keep their device state per host thread, hibag_build.hip: trainers driven by different threads run
	// side by side on the device -- hlaConcurrentAttrBagging)
	const size_t before = t->out.size();
	if (hibag_hip_set_device(t->device)) return HIBAG_HIP_ENODEV;     // the build entries allocate on the selected device
	if (hipSetDevice(t->device) != hipSuccess) return hibag_fail(HIBAG_HIP_ENODEV, "hipSetDevice(%d) failed", t->device);
	// a trainer that runs beside others hands its device work to the device's combiners (one fused launch per kind of
	// operation for all of them) and counts against the shared host-thread budget while it is runnable
	struct Shared {
		explicit Shared(bool on) { hibag_combine_set_shared(on); hibag_combine_enter(); }
		~Shared() { hibag_combine_leave(); hibag_combine_set_shared(false); }
	} shared_scope(t->shared);
	try {
		if (!t->pool && t->n_threads > 1) t->pool.reset(new Pool(t->n_threads - 1));
		g_prof = Profile();
		for (long long &v : g_em_stats) v = 0;
		for (double &v : g_em_prof) v = 0;
		for (double &v : g_batch_prof) v = 0;
		const double t0 = Profile::now();
		build_classifiers(*t, nclassifier, mtry, prune != 0, verbose != 0 || verbose_detail != 0, verbose_detail != 0);
		if (getenv("HIBAG_TRAIN_PROFILE"))
			fprintf(stderr, "[hibag train] EM fits on the device: %lld candidates, %lld handed back to the host's log(), %.1f iterations each; "
				"pairs per growth step: mean %.0f, max %lld; longest list of a haplotype %lld; in the batch call %.3f s (staging %.3f, copy + kernel + copy %.3f); the slowest candidate of a step: %.1f iterations; most pairs of one sample %lld\n",
				g_em_stats[0], g_em_stats[1], g_em_stats[0] ? (double)g_em_stats[2] / g_em_stats[0] : 0.0,
				g_prof.t[7] > 0 ? (double)g_em_stats[3] / g_prof.t[7] : 0.0, g_em_stats[4], g_em_stats[5], g_em_prof[2], g_em_prof[0], g_em_prof[1], g_prof.t[7] > 0 ? (double)g_em_stats[6] / g_prof.t[7] : 0.0, g_em_stats[7]);
		if (getenv("HIBAG_TRAIN_PROFILE"))
			fprintf(stderr, "[hibag train] total %.3f s: pair lists (device) %.3f, EM (host) %.3f, scoring (device) %.3f "
				"[pack %.3f (staging %.3f, allocation %.3f), copy+kernels %.3f, read-back %.3f, reductions %.3f], compare + accept %.3f, select %.3f; search() %.3f, %d growth steps\n",
				Profile::now() - t0, g_prof.t[0], g_prof.t[1], g_prof.t[2], g_batch_prof[0], g_batch_prof[4], g_batch_prof[5], g_batch_prof[1], g_batch_prof[2], g_batch_prof[3],
				g_prof.t[3], g_prof.t[4], g_prof.t[5], (int)g_prof.t[7]);
		if (getenv("HIBAG_TRAIN_PROFILE"))
			fprintf(stderr, "[hibag train] this thread's CPU time: pair lists %.3f, EM %.3f, scoring %.3f, compare %.3f, select %.3f; search() %.3f s = %.3f ms per growth step; of EM: fits on this thread %.3f\n",
				g_prof.c[0], g_prof.c[1], g_prof.c[2], g_prof.c[3], g_prof.c[4], g_prof.c[5], g_prof.t[7] > 0 ? 1e3 * g_prof.c[5] / g_prof.t[7] : 0.0, g_prof.c[6]);
	} catch (const char *msg) {
		t->out.resize(before);                         // a failed call adds nothing
		return hibag_fail(hibag_hip_device_count() <= 0 ? HIBAG_HIP_ENODEV : HIBAG_HIP_EINVAL, "%s", msg);
	} catch (const std::bad_alloc &) {
		t->out.resize(before);
		return hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory");
	}
	return 0;
}

int hibag_hip_trainer_n_classifier(const hibag_hip_trainer *t) { return t ? (int)t->out.size() : 0; }

int hibag_hip_trainer_classifier_dims(const hibag_hip_trainer *t, int idx, int *n_snp_c, int *n_haplo)
{
	if (!t || idx < 0 || idx >= (int)t->out.size()) return hibag_fail(HIBAG_HIP_EINVAL, "invalid classifier index");
	if (n_snp_c) *n_snp_c = (int)t->out[idx].snpidx.size();
	if (n_haplo) *n_haplo = (int)t->out[idx].freq.size();
	return 0;
}

int hibag_hip_trainer_classifier_get(const hibag_hip_trainer *t, int idx, int32_t *snpidx, int32_t *samp_num,
	double *freq, int32_t *hla, uint64_t *bits, double *outofbag_acc)
{
	if (!t || idx < 0 || idx >= (int)t->out.size()) return hibag_fail(HIBAG_HIP_EINVAL, "invalid classifier index");
	const OutClassifier &o = t->out[idx];
	if (snpidx) std::copy(o.snpidx.begin(), o.snpidx.end(), snpidx);
	if (samp_num) std::copy(o.samp_num.begin(), o.samp_num.end(), samp_num);
	if (freq) std::copy(o.freq.begin(), o.freq.end(), freq);
	if (hla) std::copy(o.hla.begin(), o.hla.end(), hla);
	if (bits) std::copy(o.bits.begin(), o.bits.end(), bits);
	if (outofbag_acc) *outofbag_acc = o.acc;
	return 0;
}

} // extern "C"
