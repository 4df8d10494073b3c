// hibag_build.hip -- the training-side kernels of the HIBAG plugin table
// (build_* entries of TypeGPUExtProc, inst/include/LibHLA_ext.h:358-388): the
// same haplotype-pair loop as prediction, with the reductions the greedy SNP
// search needs while it grows a classifier (src/LibHLA.cpp:1981-2122):
//
//   build_acc_oob  sum over out-of-bag samples of #alleles called correctly by
//                  _BestGuess (src/LibHLA.cpp:1639-1704, :1934-1955)
//   build_acc_ib   -2 * sum over in-bag samples of count * log(_PostProb(true
//                  pair)) (src/LibHLA.cpp:1706-1767, :1957-1979)
//   build_haplomatch  per in-bag sample the haplotype pairs of its true alleles
//                  at distance 0, else at the minimum distance
//                  (_PrepHaploMatch, src/LibHLA.cpp:1569-1637, consumed at :1037-1063)
//
// The haplotype list changes with every candidate SNP, so unlike prediction
// nothing is flattened on the host: k_build_eval walks the allele-grouped SoA
// list directly (lane = sample, haplotype words and frequencies through the
// scalar cache), adding terms in the reference's order.  Sums that the
// reference forms on the host (log, the in-order log-likelihood) stay on the
// host so that they round identically.

#include <hip/hip_runtime.h>
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hibag_device.h"

int hibag_selected_device();      // hibag_api.hip
#include "hibag_plugin.h"
#include "hibag_combine.h"

namespace {

constexpr int NW = 4;                       // 32-bit words of a 128-SNP string

struct BuildState {
	bool active = false;
	int n_hla = 0, n_sample = 0, n_pad = 0;
	std::vector<int> boot;                  // bootstrap count per sample (0 = out-of-bag)
	std::vector<int> inbag, oob;            // sample indices, ascending (src/LibHLA.cpp:1858-1874)
	// current candidate (build_set_haplo_geno)
	int n_haplo = 0, n_snp = 0;
	std::vector<int> true1, true2;          // true allele pair per sample, a1 <= a2
	bool evaluated = false;
	std::vector<int> best1, best2;
	std::vector<double> postprob;
	// device
	void *d_hb = nullptr, *d_hf = nullptr, *d_start = nullptr, *d_planes = nullptr, *d_true = nullptr,
		*d_best = nullptr, *d_post = nullptr, *d_tab = nullptr, *d_match = nullptr, *d_batch = nullptr;
	void *h_stage = nullptr;                // pinned host staging of the batched evaluation
	size_t cap_h = 0, cap_s = 0, cap_match = 0, cap_batch = 0, cap_stage = 0;
	// small uploads of the pair-list step: staged in pinned memory and sent asynchronously (a pageable hipMemcpy is a
	// synchronisation each; a growth step made a dozen of them)
	// the batched evaluation runs in up to two slots at a time (hibag_build_eval_launch / _collect): while the device scores
	// one half of a growth step's candidates the host threads still fit the other half
	struct Slot {
		void *d = nullptr, *h = nullptr; size_t cap_d = 0, cap_h = 0;
		int n_cand = 0, np = 0; size_t o_best = 0, o_post = 0;
		double t_launch = 0;
	} slot[2];
	void *h_up = nullptr; size_t cap_up = 0, up_at = 0;
	std::vector<void *> h_retired;          // staging areas outgrown while an operation was being put together (freed at the next rewind)
	void *h_dn = nullptr; size_t cap_dn = 0; // pinned landing area of the pair-list step's read-backs
	void *d_geno8 = nullptr; int geno8_snps = 0;   // the cohort's genotype codes [n_snp][n_pad] (hibag_build_set_genotypes), or null
	void *d_pairs = nullptr; size_t cap_pairs = 0;
};
// One state per HOST THREAD (thread_local), and every copy / launch of this file on the calling thread's own default
// stream (the file is compiled with -fgpu-default-stream=per-thread): several trainers of one process -- each driven by its
// own host thread, hibag_amd.train.hlaConcurrentAttrBagging -- then share the device instead of queueing behind one another.
// An unmodified HIBAG drives the build entries from one thread (nthread = 1 with a GPU plugin, src/LibHLA.h:680).
thread_local BuildState g;
thread_local char g_msg[400];
// The operation the calling thread is putting together (hibag_combine.h): upload() then RECORDS its copies instead of
// issuing them -- the driver's entries (pair lists, batched scoring) run as operations, alone or fused with other trainers';
// the plugin-table entries an unmodified HIBAG drives (set_haplo_geno + acc_oob / acc_ib) copy and launch directly.
thread_local HibagOp *g_op = nullptr;

[[noreturn]] void build_throw(const char *what, hipError_t e = hipSuccess)
{
	if (e != hipSuccess) snprintf(g_msg, sizeof(g_msg), "HIBAG HIP plugin: %s: %s", what, hipGetErrorString(e));
	else snprintf(g_msg, sizeof(g_msg), "HIBAG HIP plugin: %s", what);
	throw (const char *)g_msg;
}
#define HIP_OK(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) build_throw(what, e_); } while (0)

void dev_free(void *&p) { if (p) (void)hipFree(p); p = nullptr; }

// Host -> device through the pinned staging area, asynchronously on the null stream (in order with the kernels that
// follow).  The area is rewound by upload_rewind() at a point where everything sent before has been consumed.
void upload_rewind()
{
	g.up_at = 0;
	for (void *p : g.h_retired) (void)hipHostFree(p);
	g.h_retired.clear();
}
void upload(void *dst, const void *src, size_t bytes, const char *what)
{
	if (bytes == 0) return;
	if (g.up_at + bytes > g.cap_up) {
		// (rare: grow; what is in flight from the old area must land first -- or, while an operation is being put together,
		// the old area stays alive until that operation has run: its recorded copies point into it)
		if (!g_op) HIP_OK(hipStreamSynchronize(0), what);
		if (g.h_up) { if (g_op) g.h_retired.push_back(g.h_up); else (void)hipHostFree(g.h_up); }
		g.h_up = nullptr;
		g.cap_up = std::max<size_t>((g.up_at + bytes) * 2, 1 << 20);
		HIP_OK(hipHostMalloc(&g.h_up, g.cap_up, hipHostMallocDefault), "hipHostMalloc(upload staging)");
		g.up_at = 0;
	}
	char *at = (char *)g.h_up + g.up_at;
	memcpy(at, src, bytes);
	g.up_at += (bytes + 63) & ~(size_t)63;
	if (g_op) g_op->up.push_back(HibagCopy{dst, at, bytes});
	else HIP_OK(hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, 0), what);
}

void *landing(size_t bytes)
{
	if (bytes > g.cap_dn) {
		if (g.h_dn) (void)hipHostFree(g.h_dn);
		g.h_dn = nullptr; g.cap_dn = 0;
		HIP_OK(hipHostMalloc(&g.h_dn, bytes * 2 + 4096, hipHostMallocDefault), "hipHostMalloc(read-back staging)");
		g.cap_dn = bytes * 2 + 4096;
	}
	return g.h_dn;
}

void reserve(void *&p, size_t &cap, size_t bytes, const char *what)
{
	if (bytes <= cap && p) return;
	dev_free(p);
	bytes = std::max(bytes, cap + cap / 2);      // geometric growth: a free + allocation of ~100 MB costs about a millisecond
	HIP_OK(hipMalloc(&p, bytes), what);
	cap = bytes;
}

// ---------------------------------------------------------------------------
// device side

struct BuildView {
	int n_hla, n_sample, n_pad, n_haplo, nw;
	const uint32_t *hb;      // [nw][n_haplo]
	const double *hf;        // [n_haplo]
	const int *start;        // [n_hla+1]
	const uint32_t *planes;  // [2*NW][n_pad]: S1 words then S2 words
	const int *true_cell;    // [n_pad] posterior index of the true pair
	const double *tab;
	int *best;               // [2][n_pad]
	double *post;            // [n_pad]
};

template <int W>
struct LaneG { uint32_t zt[W], t[W], e[W]; int n_het; };

// One allele-pair cell in the reference's order (src/LibHLA.cpp:1653-1668 diagonal,
// :1680-1691 off-diagonal): d = popc((A^T)&ZT) + popc((B^T)&ZT) + popc(~(A^B)&E),
// the per-SNP form of hamm_d (:747-819): g=0 -> h1+h2, g=2 -> 2-h1-h2, g=1 -> [h1==h2].
template <int W>
__device__ __forceinline__ double build_cell(const BuildView &V, int a0, int a1, int b0, int b1, bool diagonal,
	const LaneG<W> &G, const double *tab_s)
{
	double cell = 0;
	for (int a = a0; a < a1; a++) {
		uint32_t A[W];
		int ca = 0;
#pragma unroll
		for (int w = 0; w < W; w++) { A[w] = V.hb[w * V.n_haplo + a]; ca += __popc((A[w] ^ G.t[w]) & G.zt[w]); }
		const double fa = V.hf[a];
		int b = b0;
		if (diagonal) { cell += (fa * fa) * tab_s[2 * ca + G.n_het]; b = a + 1; }
		const double ff = 2 * fa;
		for (; b < b1; b++) {
			int d = ca;
#pragma unroll
			for (int w = 0; w < W; w++) {
				const uint32_t Bw = V.hb[w * V.n_haplo + b];
				d += __popc((Bw ^ G.t[w]) & G.zt[w]) + __popc(~(A[w] ^ Bw) & G.e[w]);
			}
			cell += (ff * V.hf[b]) * tab_s[d];
		}
	}
	return cell;
}

// lane = sample: strict first maximum over cells (_BestGuess) and
// cell(true pair) / in-order total (_PostProb) in one walk.
template <int W>
__device__ __forceinline__ void build_eval(const BuildView &V, int s, const double *tab_s)
{
	LaneG<W> G;
	G.n_het = 0;
#pragma unroll
	for (int w = 0; w < W; w++) {
		const uint32_t s1 = V.planes[(size_t)w * V.n_pad + s], s2 = V.planes[(size_t)(NW + w) * V.n_pad + s];
		G.zt[w] = ~(s1 ^ s2); G.t[w] = s1 & s2; G.e[w] = s1 & ~s2;
		G.n_het += __popc(G.e[w]);
	}
	const int want = V.true_cell[s];
	double best = 0, total = 0, hit = 0;
	int b1 = -2147483647 - 1, b2 = -2147483647 - 1, p = 0;
	for (int h1 = 0; h1 < V.n_hla; h1++) {
		const int a0 = V.start[h1], a1 = V.start[h1 + 1];
		if (a0 == a1) { p += V.n_hla - h1; continue; }      // empty row: cells are +0.0 (never > max, add nothing)
		for (int h2 = h1; h2 < V.n_hla; h2++, p++) {
			const int c0 = V.start[h2], c1 = V.start[h2 + 1];
			if (c0 == c1) continue;
			const double cell = build_cell<W>(V, a0, a1, c0, c1, h1 == h2, G, tab_s);
			if (best < cell) { best = cell; b1 = h1; b2 = h2; }
			if (p == want) hit = cell;
			total += cell;
		}
	}
	V.best[s] = b1;
	V.best[V.n_pad + s] = b2;
	V.post[s] = hit / total;
}

__global__ __launch_bounds__(HIBAG_WAVE) void k_build_eval(BuildView V)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	for (int i = threadIdx.x; i < HIBAG_TAB_N; i += blockDim.x) tab_s[i] = V.tab[i];
	__syncthreads();
	const int s = blockIdx.x * HIBAG_WAVE + threadIdx.x;
	switch (V.nw) {
	case 1:  build_eval<1>(V, s, tab_s); break;
	case 2:  build_eval<2>(V, s, tab_s); break;
	case 3:  build_eval<3>(V, s, tab_s); break;
	default: build_eval<4>(V, s, tab_s); break;
	}
}

// ---------------------------------------------------------------------------
// Batched evaluation for the library's own training driver (hibag_train.hip): all candidate
// SNPs of one growth step in one pass, parallel over allele-pair cells as well as samples
// (a training cohort is ~10^3 samples -- 16 wavefronts if only samples were spread out).
//   k_batch_cells  (sample group, cell segment, candidate): lane = sample, walks the
//                  segment's non-empty cells, stores each cell sum
//   k_batch_scan   (sample group, candidate): lane = sample, scans the candidate's cell
//                  sums in posterior order: first strict maximum (_BestGuess), in-order
//                  total and the true pair's cell (_PostProb)
// Cell sums are formed exactly as in k_build_eval, and added in the same order.
struct BatchView {
	int n_hla, n_pad, nw, n_cand, n_seg, word;      // word = 32-bit word that holds the candidate SNP's bit
	const uint32_t *hb;        // [nw][n_haplo_total], candidates back to back
	const double *hf;          // [n_haplo_total]
	int n_haplo_total;
	const int *start;          // [n_cand][n_hla+1], absolute haplotype indices
	const uint32_t *planes;    // [2*NW][n_pad] base genotypes (candidate position missing)
	const uint32_t *cand_w;    // [n_cand][2][n_pad]: words `word` of S1 / S2 with the candidate SNP set
	const int *cells;          // [n_cand][max_cells] packed (h1 << 16 | h2) of the non-empty cells, posterior order
	const int4 *cellb;         // [n_cand][max_cells] the cell's haplotype ranges {a0, a1, b0, b1} (absolute indices): one 16-byte
	                           // scalar load per cell, requested a cell ahead, instead of a chain of five dependent ones
	const int *seg;            // [n_cand][n_seg+1] segment bounds in that list
	int max_cells;
	const int *true_cell;      // [n_pad]
	const int *wpos;           // [n_cand][n_pad] position of the sample's true pair in the candidate's cell list (-1: an empty cell)
	const double *tab;
	double *cellv;             // [n_cand][n_pad / 64][max_cells][64]: a sample group's cell sums of a candidate are one contiguous stream
	int *best;                 // [n_cand][2][n_pad]
	double *post;              // [n_cand][n_pad]
	// Round 6 (the driver's trainers): what the host used to expand per growth step stays on the device or is derived there --
	// the cohort's genotype matrix is uploaded once per training call (hibag_build_set_genotypes), so a candidate travels as
	// its SNP index instead of two bit planes per sample (`cand_w` is then null); a cell's haplotype ranges come from `cells`
	// and `start` (`cellb` null); the true pair's place in a candidate's cell list is found by the scan itself, comparing the
	// cells' packed allele pairs with the sample's (`wpos` null).  A growth step's upload shrinks from 520 KB to 125 KB and
	// its packing on the host by half (profiles/r06_notes.txt item 4).
	const int8_t *gdev;        // [n_snp_total][n_pad] genotype codes 0 / 1 / 2 / 3 = missing (padding lanes: 3), or null
	const int *cand_snp;       // [n_cand] row of `gdev` of each candidate
	int bit;                   // the candidate SNP's bit inside word `word`
	const int *true_pair;      // [n_pad] (a1 << 16) | a2 of the sample's true alleles, -1 on padding lanes
};

#define SCAN_NB 32          // cells k_batch_scan has in flight; BatchView::max_cells is a multiple of it

// One workgroup = (four sample groups, cell segment, candidate): the candidate's haplotype words and frequencies are staged
// in LDS once for the four wavefronts (round 2 fetched them pair by pair through the scalar cache -- two dependent loads
// per pair with one or two wavefronts per SIMD to hide them: 93 us per launch, profiles/r03_cfg5_*), lane = sample.
// A candidate with more haplotypes than the staging area holds (BATCH_LDS_HAPLO) keeps the direct path.
#define BATCH_WAVES 4
#define BATCH_LDS_HAPLO 1024

template <int W>
__device__ __forceinline__ double batch_cell(const uint32_t *hb, const double *hf, int n_haplo, int a0, int a1, int b0, int b1,
	bool diagonal, const LaneG<W> &G, const double *tab_s)
{
	// (the arithmetic and its order are build_cell's: src/LibHLA.cpp:1653-1668 diagonal, :1680-1691 off-diagonal)
	double cell = 0;
	for (int a = a0; a < a1; a++) {
		uint32_t A[W];
		int ca = 0;
#pragma unroll
		for (int w = 0; w < W; w++) { A[w] = hb[w * n_haplo + a]; ca += __popc((A[w] ^ G.t[w]) & G.zt[w]); }
		const double fa = hf[a];
		int b = b0;
		if (diagonal) { cell += (fa * fa) * tab_s[2 * ca + G.n_het]; b = a + 1; }
		const double ff = 2 * fa;
		for (; b + 4 <= b1; b += 4) {                 // four pairs' look-ups in flight
			int d[4];
			double fb[4];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				d[q] = ca;
#pragma unroll
				for (int w = 0; w < W; w++) {
					const uint32_t Bw = hb[w * n_haplo + b + q];
					d[q] += __popc((Bw ^ G.t[w]) & G.zt[w]) + __popc(~(A[w] ^ Bw) & G.e[w]);
				}
				fb[q] = hf[b + q];
			}
			double t[4];
#pragma unroll
			for (int q = 0; q < 4; q++) t[q] = tab_s[d[q]];
#pragma unroll
			for (int q = 0; q < 4; q++) cell += (ff * fb[q]) * t[q];
		}
		for (; b < b1; b++) {
			int d = ca;
#pragma unroll
			for (int w = 0; w < W; w++) {
				const uint32_t Bw = hb[w * n_haplo + b];
				d += __popc((Bw ^ G.t[w]) & G.zt[w]) + __popc(~(A[w] ^ Bw) & G.e[w]);
			}
			cell += (ff * hf[b]) * tab_s[d];
		}
	}
	return cell;
}

// the lane's genotype with the candidate SNP of candidate c set (TGenotype::_SetSNP on the base genotype)
template <int W>
__device__ __forceinline__ void batch_lane_genotype(const BatchView &B, int c, int s, LaneG<W> &G)
{
	G.n_het = 0;
#pragma unroll
	for (int w = 0; w < W; w++) {
		uint32_t s1 = B.planes[(size_t)w * B.n_pad + s], s2 = B.planes[(size_t)(NW + w) * B.n_pad + s];
		if (w == B.word) {
			if (B.gdev) {
				// TGenotype::_SetSNP (src/LibHLA.cpp:609-622) on the base genotype's word: g = 0 -> (0, 0), 1 -> (1, 0), 2 -> (1, 1), missing -> (0, 1)
				const uint32_t v = (uint32_t)(uint8_t)B.gdev[(size_t)B.cand_snp[c] * B.n_pad + s];
				const uint32_t b1 = (uint32_t)(v == 1u) | (uint32_t)(v == 2u), b2 = (uint32_t)(v > 1u), keep = ~(1u << B.bit);
				s1 = (s1 & keep) | (b1 << B.bit);
				s2 = (s2 & keep) | (b2 << B.bit);
			} else { s1 = B.cand_w[((size_t)c * 2) * B.n_pad + s]; s2 = B.cand_w[((size_t)c * 2 + 1) * B.n_pad + s]; }
		}
		G.zt[w] = ~(s1 ^ s2); G.t[w] = s1 & s2; G.e[w] = s1 & ~s2;
		G.n_het += __popc(G.e[w]);
	}
}

// a cell's haplotype ranges {a0, a1, b0, b1}: one 16-byte load from `cellb`, or -- where the host no longer writes that
// table -- from the cell's packed allele pair and the candidate's allele starts (all wave-uniform: scalar loads)
__device__ __forceinline__ int4 batch_cell_ranges(const BatchView &B, int c, int i)
{
	if (B.cellb) return B.cellb[(size_t)c * B.max_cells + i];
	const int *__restrict__ st = B.start + (size_t)c * (B.n_hla + 1);
	const int hh = B.cells[(size_t)c * B.max_cells + i], h1 = hh >> 16, h2 = hh & 0xFFFF;
	return int4{st[h1], st[h1 + 1], st[h2], st[h2 + 1]};
}

template <int W>
__device__ __forceinline__ void batch_cells(const BatchView &B, int c, int sg, int s, const double *tab_s,
	const uint32_t *hb_s, const double *hf_s, int h_lo, int n_h)
{
	LaneG<W> G;
	batch_lane_genotype<W>(B, c, s, G);
	const int i0 = B.seg[c * (B.n_seg + 1) + sg], i1 = B.seg[c * (B.n_seg + 1) + sg + 1];
	double *__restrict__ out = B.cellv + ((size_t)c * (B.n_pad / HIBAG_WAVE) + (s >> 6)) * B.max_cells * HIBAG_WAVE + (s & 63);
	auto ranges = [&](int i) { return batch_cell_ranges(B, c, i); };
	int4 nb = ranges(i0 < i1 ? i0 : 0);
	for (int i = i0; i < i1; i++) {
		const int4 r = nb;
		nb = ranges(i + 1 < i1 ? i + 1 : i);       // the next cell's ranges travel while this cell is summed
		double cell;
		if (hb_s)      // the candidate's list sits in LDS, indices relative to its first haplotype
			cell = batch_cell<W>(hb_s, hf_s, n_h, r.x - h_lo, r.y - h_lo, r.z - h_lo, r.w - h_lo, r.x == r.z, G, tab_s);
		else
			cell = batch_cell<W>(B.hb, B.hf, B.n_haplo_total, r.x, r.y, r.z, r.w, r.x == r.z, G, tab_s);
		out[(size_t)i * HIBAG_WAVE] = cell;
	}
	if (sg == B.n_seg - 1)                      // k_batch_scan reads whole groups of SCAN_NB cells: the list's end is padded with +0.0
		for (int i = i1; i < (i1 + SCAN_NB - 1) / SCAN_NB * SCAN_NB; i++) out[(size_t)i * HIBAG_WAVE] = 0.0;
}

// (a launch scores the growth steps of several trainers at once, hibag_combine.h: workgroup -> its step's view and its place in
// that step's own grid (sample group quads, cell segments, candidates))
__global__ __launch_bounds__(BATCH_WAVES * HIBAG_WAVE) void k_batch_cells(HibagMulti<BatchView> M)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	__shared__ double hf_s[BATCH_LDS_HAPLO];
	__shared__ uint32_t hb_s[NW * BATCH_LDS_HAPLO];
	const int owner = hibag_multi_owner(M, (int)blockIdx.x);
	const BatchView B = M.v[owner];
	const int local = (int)blockIdx.x - M.first[owner];
	const int gx = (B.n_pad / HIBAG_WAVE + BATCH_WAVES - 1) / BATCH_WAVES;
	const int bx = local % gx, by = (local / gx) % B.n_seg, c = local / (gx * B.n_seg);
	const int *st = B.start + (size_t)c * (B.n_hla + 1);
	const int h_lo = st[0], n_h = st[B.n_hla] - st[0];
	const bool staged = n_h <= BATCH_LDS_HAPLO;
	for (int i = threadIdx.x; i < HIBAG_TAB_N; i += blockDim.x) tab_s[i] = B.tab[i];
	if (staged) {
		for (int i = threadIdx.x; i < n_h; i += blockDim.x) hf_s[i] = B.hf[h_lo + i];
		for (int w = 0; w < B.nw; w++)
			for (int i = threadIdx.x; i < n_h; i += blockDim.x) hb_s[w * n_h + i] = B.hb[(size_t)w * B.n_haplo_total + h_lo + i];
	}
	__syncthreads();
	const int s = bx * (int)blockDim.x + (int)threadIdx.x;
	if (s >= B.n_pad) return;
	const uint32_t *hbp = staged ? hb_s : nullptr;
	switch (B.nw) {
	case 1:  batch_cells<1>(B, c, by, s, tab_s, hbp, hf_s, h_lo, n_h); break;
	case 2:  batch_cells<2>(B, c, by, s, tab_s, hbp, hf_s, h_lo, n_h); break;
	case 3:  batch_cells<3>(B, c, by, s, tab_s, hbp, hf_s, h_lo, n_h); break;
	default: batch_cells<4>(B, c, by, s, tab_s, hbp, hf_s, h_lo, n_h); break;
	}
}

__global__ __launch_bounds__(HIBAG_WAVE) void k_batch_scan(HibagMulti<BatchView> M)
{
	const int owner = hibag_multi_owner(M, (int)blockIdx.x);
	const BatchView B = M.v[owner];
	const int local = (int)blockIdx.x - M.first[owner];
	const int ng = B.n_pad / HIBAG_WAVE;
	const int c = local / ng;
	const int s = (local % ng) * HIBAG_WAVE + (int)threadIdx.x;
	const int *cl = B.cells + (size_t)c * B.max_cells;
	const int n = B.seg[c * (B.n_seg + 1) + B.n_seg];
	const int wpos = B.wpos ? B.wpos[(size_t)c * B.n_pad + s] : -2;
	const int tp = B.true_pair ? B.true_pair[s] : -1;                 // (the cells' packed pairs are >= 0; the list's padding holds -1 ... and so do padding lanes)
	double best = 0, total = 0, hit = 0;
	int bi = -1;
	const double *__restrict__ col = B.cellv + ((size_t)c * (B.n_pad / HIBAG_WAVE) + (s >> 6)) * B.max_cells * HIBAG_WAVE + (s & 63);
	// SCAN_NB loads in flight, then the strictly ordered scan over them: one load per dependent iteration would make
	// this kernel pure memory latency (hundreds of cells per candidate).  No condition anywhere near the loads -- with
	// one branch per load (round 2's form) the compiler serialised them: 146 us per launch (profiles/r03_cfg5_*) -- the list
	// is padded with +0.0 instead, which can neither become the maximum nor change the total.
	// Round 6: two such groups take turns -- the next group's loads are on their way while this one is scanned (the kernel was
	// still twenty load latencies long, 66 us: as long as the evaluation of the cells itself).  Same operations, same order.
	auto request = [&](double (&v)[SCAN_NB], const double *from) {
#pragma unroll
		for (int j = 0; j < SCAN_NB; j++) v[j] = __builtin_nontemporal_load(from + (size_t)j * HIBAG_WAVE);
	};
	auto scan = [&](const double (&v)[SCAN_NB], int i0) {
#pragma unroll
		for (int j = 0; j < SCAN_NB; j++) {
			const double cell = v[j];
			if (best < cell) { best = cell; bi = i0 + j; }          // first strict maximum (_BestGuess, src/LibHLA.cpp:1639-1704)
			// the true pair's cell (_PostProb, :1706-1767): by its place in the list, or by comparing the cell's alleles with the sample's
			if (B.wpos ? i0 + j == wpos : (cl[i0 + j] == tp && tp >= 0)) hit = cell;
			total += cell;
		}
	};
	double va[SCAN_NB], vb[SCAN_NB];
	const size_t step = (size_t)SCAN_NB * HIBAG_WAVE;
	if (n > 0) request(va, col);
	for (int i0 = 0; i0 < n; i0 += 2 * SCAN_NB, col += 2 * step) {
		if (i0 + SCAN_NB < n) request(vb, col + step);
		scan(va, i0);
		if (i0 + SCAN_NB >= n) break;
		if (i0 + 2 * SCAN_NB < n) request(va, col + 2 * step);
		scan(vb, i0 + SCAN_NB);
	}
	const int hh = bi >= 0 ? cl[bi] : 0;
	B.best[((size_t)c * 2) * B.n_pad + s] = bi >= 0 ? hh >> 16 : -2147483647 - 1;
	B.best[((size_t)c * 2 + 1) * B.n_pad + s] = bi >= 0 ? hh & 0xFFFF : -2147483647 - 1;
	B.post[(size_t)c * B.n_pad + s] = hit / total;
}

// Round 6, a VARIANT (HIBAG_BATCH_FUSED=1; the default stays the two kernels above -- the measurement is at the end of this
// comment): both steps in ONE kernel, the cell sums never leave the CU.  The two kernels above exchange every cell sum of
// every sample and candidate through HBM -- 190 MB written and read again per growth step of config 5's shape -- and the scan,
// 288 wavefronts each waiting for its own loads forty times over, took as long as the evaluation itself (116 of 228 us per
// growth step in a trace of sixteen trainers, profiles/r06_notes.txt item 4h).  Here one workgroup = (candidate, sample
// group): eleven PRODUCER wavefronts take the candidate's cells one at a time from a counter, in posterior order, evaluate
// them for the group's 64 samples (lane = sample, batch_cell as above) and put the sums into a ring in LDS; ONE wavefront scans
// the ring in cell order -- first strict maximum, in-order total, the true pair's cell: k_batch_scan's operations in its
// order.  Same bits; no cell segments, no cellv.  Measured (16 trainers, config 5's shape): 352 us per launch against
// 180 + 185 for the two kernels, 170 against 173 classifiers/s -- the cells of one (candidate, sample group) are now evaluated
// on ONE CU (eleven wavefronts) instead of spread over the device in 28 segments, and a workgroup's 70 KB of LDS keeps it off
// the CUs that hold an EM fit; a trainer alone: 31.1 against 30.2.  Not a gain: not the default.
#define SCORE_WAVES 12
#define SCORE_RING 64       // slots of 64 doubles; a producer is at most this far ahead of the scan
#define SCORE_STEP 8        // cells the scan takes at a time (SCORE_RING is a multiple)
#define SCORE_CELLS_LDS 2048 // the candidate's cell list sits in LDS for the scan if it has at most this many cells (50 alleles: 1,275)
struct ScoreShared {
	double ring[SCORE_RING][HIBAG_WAVE];
	int ready[SCORE_RING];      // cell index + 1 of what the slot holds
	int next_cell, scan_pos;    // the next cell to hand out; cells the scan has consumed
	int sink[SCORE_WAVES][HIBAG_WAVE];   // where the lanes that have nothing to publish store (see score_publish)
};

// One lane of the wavefront publishes `value` at `word`.  Written WITHOUT a branch on the lane -- every lane stores, all but
// lane 0 into a word of their own that nobody reads: with `if (lane == 0)` around the store and around the counter's fetch-add
// the compiler threaded the two branches through the loop's back edge into separate loops for lane 0 and for the other lanes,
// whose readfirstlane then read a lane that had never fetched anything (ISA of the first build; the kernel never ended).
__device__ __forceinline__ void score_publish(ScoreShared &S, int *word, int value)
{
	const int lane = (int)threadIdx.x & (HIBAG_WAVE - 1), wave = (int)threadIdx.x >> 6;
	int *to = lane == 0 ? word : &S.sink[wave][lane];
	__hip_atomic_store(to, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int score_peek(const int *word)          // the same value in every lane, and the compiler knows it
{
	return __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}

template <int W>
__device__ __forceinline__ void batch_produce(const BatchView &B, int c, int s, int n, ScoreShared &S, const double *tab_s,
	const uint32_t *hb_s, const double *hf_s, int h_lo, int n_h)
{
	LaneG<W> G;
	batch_lane_genotype<W>(B, c, s, G);
	const int lane = (int)threadIdx.x & (HIBAG_WAVE - 1);
	for (;;) {
		// (every lane adds one -- the compiler makes that ONE add of 64 by one lane -- so the counter runs in units of 64: no
		// branch on the lane in the source)
		const int k = __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add(&S.next_cell, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 6;
		if (k >= n) break;
		const int4 r = batch_cell_ranges(B, c, k);
		double cell;
		if (hb_s) cell = batch_cell<W>(hb_s, hf_s, n_h, r.x - h_lo, r.y - h_lo, r.z - h_lo, r.w - h_lo, r.x == r.z, G, tab_s);
		else      cell = batch_cell<W>(B.hb, B.hf, B.n_haplo_total, r.x, r.y, r.z, r.w, r.x == r.z, G, tab_s);
		const int slot = k & (SCORE_RING - 1);
		// (the slot's previous cell, k - SCORE_RING, must have been scanned)
		while (k - score_peek(&S.scan_pos) >= SCORE_RING) __builtin_amdgcn_s_sleep(2);
		S.ring[slot][lane] = cell;
		score_publish(S, &S.ready[slot], k + 1);                       // (release: behind the wavefront's ring store)
	}
}

__global__ __launch_bounds__(SCORE_WAVES * HIBAG_WAVE) void k_batch_score(HibagMulti<BatchView> M)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	__shared__ double hf_s[BATCH_LDS_HAPLO];
	__shared__ uint32_t hb_s[NW * BATCH_LDS_HAPLO];
	__shared__ ScoreShared S;
	__shared__ int cells_s[SCORE_CELLS_LDS];
	const int owner = hibag_multi_owner(M, (int)blockIdx.x);
	const BatchView B = M.v[owner];
	const int local = (int)blockIdx.x - M.first[owner];
	const int ng = B.n_pad / HIBAG_WAVE;
	const int c = local / ng, grp = local % ng;
	const int n = B.seg[c * (B.n_seg + 1) + B.n_seg];                 // the candidate's cells
	const int *cl = B.cells + (size_t)c * B.max_cells;
	if (n <= SCORE_CELLS_LDS)                                         // (the scan compares every cell's allele pair with the sample's: not from global memory, cell by cell)
		for (int i = threadIdx.x; i < n; i += blockDim.x) cells_s[i] = cl[i];
	const int *st = B.start + (size_t)c * (B.n_hla + 1);
	const int h_lo = st[0], n_h = st[B.n_hla] - st[0];
	const bool staged = n_h <= BATCH_LDS_HAPLO;
	for (int i = threadIdx.x; i < HIBAG_TAB_N; i += blockDim.x) tab_s[i] = B.tab[i];
	if (staged) {
		for (int i = threadIdx.x; i < n_h; i += blockDim.x) hf_s[i] = B.hf[h_lo + i];
		for (int w = 0; w < B.nw; w++)
			for (int i = threadIdx.x; i < n_h; i += blockDim.x) hb_s[w * n_h + i] = B.hb[(size_t)w * B.n_haplo_total + h_lo + i];
	}
	if (threadIdx.x < SCORE_RING) S.ready[threadIdx.x] = 0;
	if (threadIdx.x == 0) { S.next_cell = 0; S.scan_pos = 0; }
	__syncthreads();
	const int lane = (int)threadIdx.x & (HIBAG_WAVE - 1), wave = (int)threadIdx.x >> 6;
	const int s = grp * HIBAG_WAVE + lane;
	if (wave > 0) {
		const uint32_t *hbp = staged ? hb_s : nullptr;
		switch (B.nw) {
		case 1:  batch_produce<1>(B, c, s, n, S, tab_s, hbp, hf_s, h_lo, n_h); break;
		case 2:  batch_produce<2>(B, c, s, n, S, tab_s, hbp, hf_s, h_lo, n_h); break;
		case 3:  batch_produce<3>(B, c, s, n, S, tab_s, hbp, hf_s, h_lo, n_h); break;
		default: batch_produce<4>(B, c, s, n, S, tab_s, hbp, hf_s, h_lo, n_h); break;
		}
		return;
	}
	// ---- the scan (k_batch_scan's, fed from the ring)
	const int *clp = n <= SCORE_CELLS_LDS ? cells_s : cl;
	const int wpos = B.wpos ? B.wpos[(size_t)c * B.n_pad + s] : -2;
	const int tp = B.true_pair ? B.true_pair[s] : -1;
	double best = 0, total = 0, hit = 0;
	int bi = -1;
	// SCORE_STEP cells at a time: their flags in one look (lane j of every eight reads flag j), their sums requested
	// together -- cell by cell the scan was two dependent LDS round trips per cell, a chain as long as the producers' work
	for (int k0 = 0; k0 < n; k0 += SCORE_STEP) {
		const int m = min(SCORE_STEP, n - k0), s0 = k0 & (SCORE_RING - 1), j8 = lane & (SCORE_STEP - 1);
		for (;;) {
			const int seen = __hip_atomic_load(&S.ready[s0 + j8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (__ballot(j8 >= m || seen == k0 + j8 + 1) == ~0ull) break;
			__builtin_amdgcn_s_sleep(1);
		}
		double v[SCORE_STEP];
#pragma unroll
		for (int j = 0; j < SCORE_STEP; j++) v[j] = S.ring[s0 + j][lane];
#pragma unroll
		for (int j = 0; j < SCORE_STEP; j++) {
			if (j >= m) break;
			const double cell = v[j];
			const int k = k0 + j;
			if (best < cell) { best = cell; bi = k; }                    // first strict maximum (_BestGuess, src/LibHLA.cpp:1639-1704)
			if (B.wpos ? k == wpos : (clp[k] == tp && tp >= 0)) hit = cell;   // the true pair's cell (_PostProb, :1706-1767)
			total += cell;
		}
		// (the slots are free once their values are in registers: LDS serves a wavefront's accesses in order, the release keeps the compiler to it)
		score_publish(S, &S.scan_pos, k0 + m);
	}
	const int hh = bi >= 0 ? cl[bi] : 0;
	B.best[((size_t)c * 2) * B.n_pad + s] = bi >= 0 ? hh >> 16 : -2147483647 - 1;
	B.best[((size_t)c * 2 + 1) * B.n_pad + s] = bi >= 0 ? hh & 0xFFFF : -2147483647 - 1;
	B.post[(size_t)c * B.n_pad + s] = hit / total;
}

// One wavefront per in-bag sample, lanes over the haplotype pairs of its true
// alleles.  Pass 0 finds the minimum distance and counts the pairs at it; pass 1
// (after the host turned counts into offsets) writes them in the reference's
// order (i1 outer, i2 inner), compacted with ballots.
struct MatchView {
	int n_haplo, nw, n_pad;
	const uint32_t *hb;
	const int *start;
	const uint32_t *planes;
	const int *samp;         // [n_inbag] sample index
	const int *a1, *a2;      // [n_inbag] true alleles, a1 <= a2
	int *min_d;              // [n_inbag]
	int *count;              // [n_inbag]
	const int *offset;       // [n_inbag] (pass 1); nullptr: pass 1 sums the counts of the samples before its own
	uint32_t *out;           // pairs: {k, (i2<<16)|i1}
	int n_inbag;             // = the step's workgroups in a fused launch
};

__device__ __forceinline__ int match_dist(const MatchView &V, int s, int i, int j)
{
	int d = 0;
	for (int w = 0; w < V.nw; w++) {
		const uint32_t s1 = V.planes[(size_t)w * V.n_pad + s], s2 = V.planes[(size_t)(NW + w) * V.n_pad + s];
		const uint32_t A = V.hb[w * V.n_haplo + i], B = V.hb[w * V.n_haplo + j];
		d += __popc((A ^ (s1 & s2)) & ~(s1 ^ s2)) + __popc((B ^ (s1 & s2)) & ~(s1 ^ s2)) + __popc(~(A ^ B) & s1 & ~s2);
	}
	return d;
}

template <int PASS>
__global__ __launch_bounds__(HIBAG_WAVE) void k_build_match(HibagMulti<MatchView> M)
{
	const int owner = hibag_multi_owner(M, (int)blockIdx.x);
	const MatchView V = M.v[owner];
	const int k = (int)blockIdx.x - M.first[owner], lane = threadIdx.x;
	const int s = V.samp[k];
	const int a1 = V.a1[k], a2 = V.a2[k];
	const int st1 = V.start[a1], n1 = V.start[a1 + 1] - st1;
	const int st2 = V.start[a2], n2 = V.start[a2 + 1] - st2;
	const bool same = a1 == a2;
	const long long npair = (long long)n1 * n2;
	int best = 0x7FFFFFFF, cnt = 0;
	const int target = PASS ? V.min_d[k] : 0;
	int written = 0;
	if (PASS) {
		if (V.offset) written = V.offset[k];
		else {
			// both passes in one operation: the pairs of the samples before this one, from the first pass's counts
			int acc = 0;
			for (int i = lane; i < k; i += HIBAG_WAVE) acc += V.count[i];
			for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
			written = acc;
		}
	}
	for (long long q0 = 0; q0 < npair; q0 += HIBAG_WAVE) {
		const long long q = q0 + lane;
		bool valid = q < npair;
		int i1 = 0, i2 = 0, d = 0x7FFFFFFF;
		if (valid) {
			i1 = (int)(q / n2); i2 = (int)(q - (long long)i1 * n2);
			if (same && i2 < i1) valid = false;          // triangular: i1 <= i2 (src/LibHLA.cpp:1611-1612)
		}
		if (valid) d = match_dist(V, s, st1 + i1, st2 + i2);
		if (PASS == 0) {
			if (d < best) { best = d; cnt = 0; }
			if (valid && d == best) cnt++;
		} else {
			const bool hit = valid && d == target;
			const unsigned long long m = __ballot(hit);
			if (hit) {
				const int pos = written + __popcll(m & ((1ull << lane) - 1));
				V.out[2 * (size_t)pos] = (uint32_t)k;
				V.out[2 * (size_t)pos + 1] = ((uint32_t)i2 << 16) | (uint32_t)i1;
			}
			written += __popcll(m);
		}
	}
	if (PASS == 0) {
		// wave reduction: global minimum, then the number of pairs at it
		int gmin = best;
		for (int off = 32; off > 0; off >>= 1) gmin = min(gmin, __shfl_xor(gmin, off));
		int c = (best == gmin) ? cnt : 0;
		for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
		if (lane == 0) { V.min_d[k] = gmin; V.count[k] = c; }
	}
}

// ---------------------------------------------------------------------------
// host side

void upload_haplo(const PluginHaplotype haplo[], int n_haplo, int n_snp, bool alleles_from_aux,
	const size_t *len_per_hla_or_null)
{
	const int nh = g.n_hla;
	g.n_haplo = n_haplo; g.n_snp = n_snp;
	const int nw = std::max(1, (n_snp + 31) / 32);
	std::vector<uint32_t> hb((size_t)NW * std::max(n_haplo, 1), 0);
	std::vector<double> hf(std::max(n_haplo, 1), 0.0);
	std::vector<int> start(nh + 1, 0);
	for (int i = 0; i < n_haplo; i++) {
		for (int w = 0; w < nw; w++) {
			uint32_t v = (uint32_t)((uint64_t)haplo[i].packed[w >> 1] >> (32 * (w & 1)));
			const int lo = 32 * w;                       // clear bits >= n_snp (uninitialised in the reference, src/LibHLA.cpp:287-292)
			if (n_snp < lo + 32) v &= (n_snp <= lo) ? 0u : ((1u << (n_snp - lo)) - 1);
			hb[(size_t)w * n_haplo + i] = v;
		}
		hf[i] = haplo[i].freq;
		if (alleles_from_aux) {
			const int a = haplo[i].aux.hla_allele;
			if (a < 0 || a >= nh) build_throw("haplotype with an invalid HLA allele index");
			start[a + 1]++;
		}
	}
	if (len_per_hla_or_null) for (int h = 0; h < nh; h++) start[h + 1] = (int)len_per_hla_or_null[h];
	for (int h = 0; h < nh; h++) start[h + 1] += start[h];
	if (start[nh] != n_haplo) build_throw("haplotype counts per allele do not add up");
	reserve(g.d_hb, g.cap_h, (size_t)NW * std::max(n_haplo, 1) * 4 + std::max(n_haplo, 1) * 8 + 64, "hipMalloc(haplotypes)");
	g.d_hf = (char *)g.d_hb + (((size_t)NW * std::max(n_haplo, 1) * 4 + 7) & ~(size_t)7);
	upload(g.d_hb, hb.data(), (size_t)nw * std::max(n_haplo, 1) * 4, "copy haplotypes");
	upload(g.d_hf, hf.data(), hf.size() * 8, "copy frequencies");
	upload(g.d_start, start.data(), (nh + 1) * sizeof(int), "copy allele starts");
}

void upload_geno(const PluginGenotype geno[])
{
	const int n = g.n_sample, np = g.n_pad, nh = g.n_hla;
	std::vector<uint32_t> planes((size_t)2 * NW * np);
	std::vector<int> true_cell(np, -1);
	g.true1.assign(n, 0); g.true2.assign(n, 0);
	for (int w = 0; w < NW; w++)
		for (int s = 0; s < np; s++) {                     // padding lanes: all missing
			planes[(size_t)w * np + s] = s < n ? (uint32_t)((uint64_t)geno[s].snp1[w >> 1] >> (32 * (w & 1))) : 0u;
			planes[(size_t)(NW + w) * np + s] = s < n ? (uint32_t)((uint64_t)geno[s].snp2[w >> 1] >> (32 * (w & 1))) : 0xFFFFFFFFu;
		}
	for (int s = 0; s < n; s++) {
		int a1 = geno[s].hla1, a2 = geno[s].hla2;
		if (a1 > a2) std::swap(a1, a2);                    // src/LibHLA.cpp:1710, :1863-1868
		if (a1 < 0 || a2 >= nh) build_throw("genotype with an invalid true HLA pair");
		g.true1[s] = a1; g.true2[s] = a2;
		true_cell[s] = a2 + a1 * (2 * nh - a1 - 1) / 2;    // src/LibHLA.cpp:1712
	}
	upload(g.d_planes, planes.data(), planes.size() * 4, "copy genotypes");
	upload(g.d_true, true_cell.data(), np * sizeof(int), "copy true pairs");
}

void evaluate()
{
	if (g.evaluated) return;
	BuildView V{g.n_hla, g.n_sample, g.n_pad, g.n_haplo, std::max(1, (g.n_snp + 31) / 32),
		(const uint32_t *)g.d_hb, (const double *)g.d_hf, (const int *)g.d_start, (const uint32_t *)g.d_planes,
		(const int *)g.d_true, (const double *)g.d_tab, (int *)g.d_best, (double *)g.d_post};
	hipLaunchKernelGGL(k_build_eval, dim3(g.n_pad / HIBAG_WAVE), dim3(HIBAG_WAVE), 0, 0, V);
	HIP_OK(hipGetLastError(), "k_build_eval");
	std::vector<int> best((size_t)2 * g.n_pad);
	std::vector<double> post(g.n_pad);
	HIP_OK(hipMemcpy(best.data(), g.d_best, best.size() * sizeof(int), hipMemcpyDeviceToHost), "read best guesses");
	HIP_OK(hipMemcpy(post.data(), g.d_post, post.size() * sizeof(double), hipMemcpyDeviceToHost), "read posteriors");
	g.best1.assign(best.begin(), best.begin() + g.n_sample);
	g.best2.assign(best.begin() + g.n_pad, best.begin() + g.n_pad + g.n_sample);
	g.postprob.assign(post.begin(), post.begin() + g.n_sample);
	g.evaluated = true;
}

// CHLATypeList::Compare, src/LibHLA.cpp:912-924
int compare_hla(int p1, int p2, int t1, int t2)
{
	int cnt = 0;
	if (p1 == t1 || p1 == t2) { cnt = 1; if (p1 == t1) t1 = -1; else t2 = -1; }
	if (p2 == t1 || p2 == t2) cnt++;
	return cnt;
}

// ---- fused launches (hibag_combine.h): the operations' views as kernel arguments, their workgroups back to back ----
template <class V, class Blocks>
HibagMulti<V> multi_of(const HibagOp *const ops[], int n, Blocks &&blocks, int &total)
{
	HibagMulti<V> M;
	M.n = n;
	int at = 0;
	for (int j = 0; j < n; j++) {
		M.v[j] = *(const V *)ops[j]->view;
		M.first[j] = at;
		at += blocks(M.v[j]);
	}
	for (int j = n; j <= HIBAG_COMBINE_MAX; j++) M.first[j] = at;
	total = at;
	return M;
}

void match0_launch(const HibagOp *const ops[], int n, hipStream_t st)
{
	int total = 0;
	const HibagMulti<MatchView> M = multi_of<MatchView>(ops, n, [](const MatchView &v) { return v.n_inbag; }, total);
	if (total > 0) hipLaunchKernelGGL(k_build_match<0>, dim3(total), dim3(HIBAG_WAVE), 0, st, M);
}

void match1_launch(const HibagOp *const ops[], int n, hipStream_t st)
{
	int total = 0;
	const HibagMulti<MatchView> M = multi_of<MatchView>(ops, n, [](const MatchView &v) { return v.n_inbag; }, total);
	if (total > 0) hipLaunchKernelGGL(k_build_match<1>, dim3(total), dim3(HIBAG_WAVE), 0, st, M);
}

void match_launch(const HibagOp *const ops[], int n, hipStream_t st)
{
	int total = 0;
	const HibagMulti<MatchView> M = multi_of<MatchView>(ops, n, [](const MatchView &v) { return v.n_inbag; }, total);
	if (total > 0) {
		hipLaunchKernelGGL(k_build_match<0>, dim3(total), dim3(HIBAG_WAVE), 0, st, M);
		hipLaunchKernelGGL(k_build_match<1>, dim3(total), dim3(HIBAG_WAVE), 0, st, M);
	}
}

void eval_launch(const HibagOp *const ops[], int n, hipStream_t st)
{
	int total = 0;
	// HIBAG_BATCH_FUSED=1: the one-kernel form (k_batch_score) -- bit-identical, measured equal to the two kernels beside other
	// trainers and 3 % faster for a trainer alone (profiles/r06_notes.txt item 4h): kept as a variant, not the default
	const char *fused = getenv("HIBAG_BATCH_FUSED");
	if (fused && *fused == '1') {
		const HibagMulti<BatchView> M1 = multi_of<BatchView>(ops, n, [](const BatchView &b) { return (b.n_pad / HIBAG_WAVE) * b.n_cand; }, total);
		if (total > 0) hipLaunchKernelGGL(k_batch_score, dim3(total), dim3(SCORE_WAVES * HIBAG_WAVE), 0, st, M1);
		return;
	}
	const HibagMulti<BatchView> Mc = multi_of<BatchView>(ops, n, [](const BatchView &b) {
		return ((b.n_pad / HIBAG_WAVE + BATCH_WAVES - 1) / BATCH_WAVES) * b.n_seg * b.n_cand; }, total);
	if (total > 0) hipLaunchKernelGGL(k_batch_cells, dim3(total), dim3(BATCH_WAVES * HIBAG_WAVE), 0, st, Mc);
	const HibagMulti<BatchView> Ms = multi_of<BatchView>(ops, n, [](const BatchView &b) { return (b.n_pad / HIBAG_WAVE) * b.n_cand; }, total);
	if (total > 0) hipLaunchKernelGGL(k_batch_scan, dim3(total), dim3(HIBAG_WAVE), 0, st, Ms);
}

const bool g_build_registered = (hibag_combine_register(HIBAG_OP_MATCH0, match0_launch), hibag_combine_register(HIBAG_OP_MATCH1, match1_launch),
	hibag_combine_register(HIBAG_OP_EVAL, eval_launch), hibag_combine_register(HIBAG_OP_MATCH, match_launch), true);

} // namespace

// build_init(nHLA, nSample): src/LibHLA.cpp:2256-2260
void hibag_build_init(int n_hla, int n_sample)
{
	hibag_build_done();
	if (n_hla <= 0 || n_sample < 0) build_throw("build_init: invalid dimensions");
	// every allocation and launch of the build entries goes to the device the calling thread selected
	// with hibag_hip_set_device (one process per GPU: LOCAL_RANK), not to whatever HIP's current device is
	HIP_OK(hipSetDevice(hibag_selected_device()), "hipSetDevice");
	g.n_hla = n_hla; g.n_sample = n_sample;
	g.n_pad = (std::max(n_sample, 1) + HIBAG_WAVE - 1) / HIBAG_WAVE * HIBAG_WAVE;
	g.boot.assign(n_sample, 1);
	g.inbag.clear(); g.oob.clear();
	for (int i = 0; i < n_sample; i++) g.inbag.push_back(i);
	double tab[HIBAG_TAB_N];
	for (int i = 0; i < HIBAG_TAB_N; i++) tab[i] = std::exp(i * std::log(1e-5));     // src/LibHLA.cpp:166-183
	tab[0] = 1;
	for (int i = 0; i < HIBAG_TAB_N; i++) if (!std::isfinite(tab[i])) tab[i] = 0;
	HIP_OK(hipMalloc(&g.d_tab, sizeof(tab)), "hipMalloc(table)");
	HIP_OK(hipMemcpy(g.d_tab, tab, sizeof(tab), hipMemcpyHostToDevice), "copy table");
	HIP_OK(hipMalloc(&g.d_start, (n_hla + 1) * sizeof(int)), "hipMalloc(starts)");
	HIP_OK(hipMalloc(&g.d_planes, (size_t)2 * NW * g.n_pad * 4), "hipMalloc(genotypes)");
	HIP_OK(hipMalloc(&g.d_true, (size_t)g.n_pad * sizeof(int)), "hipMalloc(true pairs)");
	HIP_OK(hipMalloc(&g.d_best, (size_t)2 * g.n_pad * sizeof(int)), "hipMalloc(best)");
	HIP_OK(hipMalloc(&g.d_post, (size_t)g.n_pad * sizeof(double)), "hipMalloc(post)");
	g.active = true;
}

// build_done(): called from a destructor (src/LibHLA.cpp:2262-2266) -- must not throw
void hibag_build_done()
{
	for (void **p : {&g.d_hb, &g.d_start, &g.d_planes, &g.d_true, &g.d_best, &g.d_post, &g.d_tab, &g.d_match, &g.d_batch}) dev_free(*p);
	for (BuildState::Slot &sl : g.slot) {
		dev_free(sl.d);
		if (sl.h) (void)hipHostFree(sl.h);
		sl = BuildState::Slot();
	}
	g.d_hf = nullptr;
	if (g.h_stage) (void)hipHostFree(g.h_stage);
	g.h_stage = nullptr;
	if (g.h_up) (void)hipHostFree(g.h_up);
	g.h_up = nullptr;
	for (void *p : g.h_retired) (void)hipHostFree(p);
	g.h_retired.clear();
	if (g.h_dn) (void)hipHostFree(g.h_dn);
	g.h_dn = nullptr; g.cap_dn = 0;
	dev_free(g.d_pairs);
	dev_free(g.d_geno8); g.geno8_snps = 0;
	g.cap_h = g.cap_s = g.cap_match = g.cap_batch = g.cap_stage = g.cap_up = g.up_at = g.cap_pairs = 0;
	g.active = false; g.evaluated = false;
}

// build_set_bootstrap(counts[nSample]): src/LibHLA.cpp:2290-2293; 0 = out-of-bag
void hibag_build_set_bootstrap(const int oob_cnt[])
{
	if (!g.active) build_throw("build_set_bootstrap before build_init");
	g.boot.assign(oob_cnt, oob_cnt + g.n_sample);
	g.inbag.clear(); g.oob.clear();
	for (int i = 0; i < g.n_sample; i++) (g.boot[i] > 0 ? g.inbag : g.oob).push_back(i);
	g.evaluated = false;
}

// The library's own driver: the cohort's genotype matrix, SNP-major int32 [n_snp][n_sample] (values outside 0..2 = missing), kept
// on the device as byte codes for the whole training call -- a growth step's candidates then travel as SNP indices
// (HibagBuildCandidate::snp) instead of two bit planes per candidate and sample.  After build_init.
void hibag_build_set_genotypes(const int32_t *geno_snp_major, int n_snp)
{
	if (!g.active) build_throw("build_set_genotypes before build_init");
	dev_free(g.d_geno8); g.geno8_snps = 0;
	if (!geno_snp_major || n_snp <= 0) return;
	const size_t np = (size_t)g.n_pad, n = (size_t)g.n_sample;
	std::vector<int8_t> codes((size_t)n_snp * np, (int8_t)3);
	for (int j = 0; j < n_snp; j++)
		for (size_t s = 0; s < n; s++) {
			const int32_t v = geno_snp_major[(size_t)j * n + s];
			codes[(size_t)j * np + s] = (0 <= v && v <= 2) ? (int8_t)v : (int8_t)3;
		}
	HIP_OK(hipMalloc(&g.d_geno8, codes.size()), "hipMalloc(genotype codes)");
	HIP_OK(hipMemcpy(g.d_geno8, codes.data(), codes.size(), hipMemcpyHostToDevice), "copy genotype codes");
	g.geno8_snps = n_snp;
}

// build_set_haplo_geno(haplo, n_haplo, geno, n_snp): src/LibHLA.cpp:1916-1920
void hibag_build_set_haplo_geno(const PluginHaplotype haplo[], int n_haplo, const PluginGenotype geno[], int n_snp)
{
	if (!g.active) build_throw("build_set_haplo_geno before build_init");
	if (n_snp < 0 || n_snp > 128 || n_haplo < 0) build_throw("build_set_haplo_geno: invalid sizes");
	HIP_OK(hipStreamSynchronize(0), "build_set_haplo_geno");       // (nothing of an earlier call may still read the staging area)
	upload_rewind();
	upload_haplo(haplo, n_haplo, n_snp, true, nullptr);
	upload_geno(geno);
	g.evaluated = false;
}

// build_acc_oob(): src/LibHLA.cpp:1938-1941 -- number of correct alleles over the out-of-bag samples
int hibag_build_acc_oob()
{
	if (!g.active) build_throw("build_acc_oob before build_init");
	evaluate();
	int correct = 0;
	for (int s : g.oob) correct += compare_hla(g.best1[s], g.best2[s], g.true1[s], g.true2[s]);
	return correct;
}

// build_acc_ib(): src/LibHLA.cpp:1961-1977 -- -2 * sum_i count_i * log(p_i), in-bag order, host libm
double hibag_build_acc_ib()
{
	if (!g.active) build_throw("build_acc_ib before build_init");
	evaluate();
	double loglik = 0;
	for (int s : g.inbag) loglik += g.boot[s] * std::log(g.postprob[s]);
	loglik *= -2;
	return loglik;
}

// build_haplomatch(haplo, nHaplo[nHLA], n_snp, geno, out_n): src/LibHLA.cpp:1037-1063.
// Returns a malloc()ed buffer the host free()s: buf[0] = 2*npairs, then npairs x
// {in-bag sample index, (i2 << 16) | i1} with i1/i2 inside the allele-specific sub-lists.
uint32_t *hibag_build_haplomatch(const PluginHaplotype haplo[], const size_t n_haplo[], int n_snp,
	const PluginGenotype geno[], size_t &out_n)
{
	if (!g.active) build_throw("build_haplomatch before build_init");
	size_t H = 0;
	for (int h = 0; h < g.n_hla; h++) {
		if (n_haplo[h] > 65535) build_throw("There are too many HLA allele-specific haplotypes (# > 65535).");
		H += n_haplo[h];
	}
	upload_rewind();                               // (every earlier upload was consumed: an operation returns when its results are back)
	{
		// ONE operation where an upper bound on the number of pairs is small (it always is for the driver's growth steps: the
		// bound is the number of candidate pairs, a few per in-bag sample): the step's inputs in one copy into one arena, both
		// passes back to back -- the second finds its offsets from the first's counts on the device --, counts and pairs back in
		// one copy sized by the bound.  Two device round trips and a dozen copies less per growth step.
		const int nib1 = (int)g.inbag.size(), nh = g.n_hla, np = g.n_pad, n = g.n_sample;
		const int nw = std::max(1, (n_snp + 31) / 32);
		std::vector<int> start(nh + 1, 0);
		for (int h = 0; h < nh; h++) start[h + 1] = start[h] + (int)n_haplo[h];
		// (g.n_haplo / g.n_snp / g.d_hb describe what build_set_haplo_geno uploaded for evaluate(): this path has an arena of its own and leaves them alone)
		g.true1.assign(n, 0); g.true2.assign(n, 0);
		for (int s = 0; s < n; s++) {
			int a1 = geno[s].hla1, a2 = geno[s].hla2;
			if (a1 > a2) std::swap(a1, a2);
			if (a1 < 0 || a2 >= nh) build_throw("genotype with an invalid true HLA pair");
			g.true1[s] = a1; g.true2[s] = a2;
		}
		size_t bound = 0;
		for (int k = 0; k < nib1; k++) {
			const size_t n1 = n_haplo[g.true1[g.inbag[k]]], n2 = n_haplo[g.true2[g.inbag[k]]];
			bound += g.true1[g.inbag[k]] == g.true2[g.inbag[k]] ? n1 * (n1 + 1) / 2 : n1 * n2;
		}
		if (nib1 > 0 && bound > 0 && bound <= ((size_t)1 << 20)) {
			g.evaluated = false;
			const size_t Hs = std::max<size_t>(H, 1);
			// the arena, in 4-byte units: inputs [hb | start | planes | samp | a1 | a2], scratch [min_d], results [count | pairs]
			const size_t o_hb = 0, o_start = o_hb + (size_t)nw * Hs, o_planes = o_start + (nh + 1), o_samp = o_planes + (size_t)2 * NW * np,
				o_a1 = o_samp + nib1, o_a2 = o_a1 + nib1, in_end = o_a2 + nib1, o_min = in_end, o_cnt = o_min + nib1, o_pairs = o_cnt + nib1,
				end = o_pairs + 2 * bound;
			std::vector<uint32_t> blob(in_end, 0);
			for (size_t i = 0; i < H; i++)
				for (int w = 0; w < nw; w++) {
					uint32_t v = (uint32_t)((uint64_t)haplo[i].packed[w >> 1] >> (32 * (w & 1)));
					const int lo = 32 * w;                   // clear bits >= n_snp (uninitialised in the reference, src/LibHLA.cpp:287-292)
					if (n_snp < lo + 32) v &= (n_snp <= lo) ? 0u : ((1u << (n_snp - lo)) - 1);
					blob[o_hb + (size_t)w * Hs + i] = v;
				}
			for (int h = 0; h <= nh; h++) blob[o_start + h] = (uint32_t)start[h];
			for (int w = 0; w < NW; w++)
				for (int s = 0; s < np; s++) {               // padding lanes: all missing
					blob[o_planes + (size_t)w * np + s] = s < n ? (uint32_t)((uint64_t)geno[s].snp1[w >> 1] >> (32 * (w & 1))) : 0u;
					blob[o_planes + (size_t)(NW + w) * np + s] = s < n ? (uint32_t)((uint64_t)geno[s].snp2[w >> 1] >> (32 * (w & 1))) : 0xFFFFFFFFu;
				}
			for (int k = 0; k < nib1; k++) {
				blob[o_samp + k] = (uint32_t)g.inbag[k];
				blob[o_a1 + k] = (uint32_t)g.true1[g.inbag[k]];
				blob[o_a2 + k] = (uint32_t)g.true2[g.inbag[k]];
			}
			reserve(g.d_match, g.cap_match, end * 4 + 64, "hipMalloc(match)");
			uint32_t *const d = (uint32_t *)g.d_match;
			HibagOp op;
			op.kind = HIBAG_OP_MATCH;
			{
				struct Collect { Collect(HibagOp *o) { g_op = o; } ~Collect() { g_op = nullptr; } } collecting(&op);
				upload(d, blob.data(), in_end * 4, "copy match inputs");
			}
			const MatchView V{(int)Hs, nw, np, d + o_hb, (const int *)(d + o_start), d + o_planes, (const int *)(d + o_samp),
				(const int *)(d + o_a1), (const int *)(d + o_a2), (int *)(d + o_min), (int *)(d + o_cnt), nullptr, d + o_pairs, nib1};
			op.view = &V;
			uint32_t *const land = (uint32_t *)landing((end - o_cnt) * 4);
			op.down.push_back(HibagCopy{land, d + o_cnt, (end - o_cnt) * 4});
			hibag_combine_run(op);
			size_t total = 0;
			for (int k = 0; k < nib1; k++) total += (size_t)(int)land[k];
			if (total > bound) build_throw("build_haplomatch: the device returned more pairs than the candidates allow");
			uint32_t *buf = (uint32_t *)malloc((1 + 2 * total) * sizeof(uint32_t));
			if (!buf) build_throw("out of memory");
			buf[0] = (uint32_t)(2 * total);
			memcpy(buf + 1, land + nib1, 2 * total * sizeof(uint32_t));
			out_n = 1 + 2 * total;
			return buf;
		}
	}
	// Otherwise two operations (hibag_combine.h): pass 0 -- the step's uploads, per in-bag sample the minimum distance and the
	// number of pairs at it --, then, once the host has turned the counts into offsets, pass 1, which writes the pairs.
	HibagOp op0;
	op0.kind = HIBAG_OP_MATCH0;
	struct Collect { Collect(HibagOp *o) { g_op = o; } ~Collect() { g_op = nullptr; } };
	const int nib = (int)g.inbag.size();
	MatchView V{};                                 // (no in-bag sample: no workgroup)
	op0.view = &V;
	{
		Collect collecting(&op0);
		upload_haplo(haplo, (int)H, n_snp, false, n_haplo);
		upload_geno(geno);
		g.evaluated = false;
		if (nib == 0) {
			hibag_combine_run(op0);                        // (the uploads alone: a later build_set_haplo_geno-free evaluation may rely on them)
			uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t));
			if (!buf) build_throw("out of memory");
			buf[0] = 0; out_n = 1;
			return buf;
		}
		std::vector<int> samp(g.inbag), a1(nib), a2(nib);
		for (int k = 0; k < nib; k++) { a1[k] = g.true1[samp[k]]; a2[k] = g.true2[samp[k]]; }
		// layout of the scratch buffer: samp, a1, a2, min_d, count, offset (ints), then the pairs
		const size_t ints = (size_t)6 * nib;
		reserve(g.d_match, g.cap_match, ints * sizeof(int) + 64, "hipMalloc(match)");
		int *d_i = (int *)g.d_match;
		upload(d_i, samp.data(), nib * sizeof(int), "copy match args");
		upload(d_i + nib, a1.data(), nib * sizeof(int), "copy match args");
		upload(d_i + 2 * nib, a2.data(), nib * sizeof(int), "copy match args");
		V = MatchView{(int)H, std::max(1, (n_snp + 31) / 32), g.n_pad, (const uint32_t *)g.d_hb, (const int *)g.d_start,
			(const uint32_t *)g.d_planes, d_i, d_i + nib, d_i + 2 * nib, d_i + 3 * nib, d_i + 4 * nib, d_i + 5 * nib, nullptr, nib};
	}
	int *const d_i = (int *)g.d_match;
	int *const count = (int *)landing((size_t)nib * sizeof(int));
	op0.down.push_back(HibagCopy{count, d_i + 4 * nib, (size_t)nib * sizeof(int)});
	hibag_combine_run(op0);
	std::vector<int> offset(nib);
	size_t total = 0;
	for (int k = 0; k < nib; k++) { offset[k] = (int)total; total += (size_t)count[k]; }
	uint32_t *buf = (uint32_t *)malloc((1 + 2 * total) * sizeof(uint32_t));
	if (!buf) build_throw("out of memory");
	buf[0] = (uint32_t)(2 * total);
	if (total > 0) {
		struct Guard { uint32_t *b; ~Guard() { free(b); } } guard{buf};       // (the calls below throw on failure)
		reserve(g.d_pairs, g.cap_pairs, 2 * total * sizeof(uint32_t), "hipMalloc(pairs)");
		V.out = (uint32_t *)g.d_pairs;
		HibagOp op1;
		op1.kind = HIBAG_OP_MATCH1;
		op1.view = &V;
		{
			Collect collecting(&op1);
			upload(d_i + 5 * nib, offset.data(), nib * sizeof(int), "copy match offsets");
		}
		uint32_t *const pairs = (uint32_t *)landing(2 * total * sizeof(uint32_t));
		op1.down.push_back(HibagCopy{pairs, g.d_pairs, 2 * total * sizeof(uint32_t)});
		hibag_combine_run(op1);
		memcpy(buf + 1, pairs, 2 * total * sizeof(uint32_t));
		guard.b = nullptr;
	}
	out_n = 1 + 2 * total;
	return buf;
}

// ---------------------------------------------------------------------------
// hibag_build_eval_batch: what a sequence of build_set_haplo_geno + build_acc_oob +
// build_acc_ib calls returns for n_cand candidate SNPs that extend the same genotype list
// (src/LibHLA.cpp:2018-2038), evaluated together.  base_geno holds the committed SNPs
// (position n_snp-1 missing); cand[i].column is the raw genotype of candidate i per sample.
thread_local double g_batch_prof[6] = {0, 0, 0, 0, 0, 0};     // host packing, copies + kernels, read-back, host reductions; of the packing: staging copy, (re)allocation (s)
static double batch_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

void hibag_build_eval_batch(const PluginGenotype base_geno[], int n_snp, const HibagBuildCandidate cand[], int n_cand,
	int acc_floor, int acc_oob[], double loss_ib[])
{
	hibag_build_eval_launch(0, base_geno, n_snp, cand, n_cand);
	hibag_build_eval_collect(0, &acc_floor, acc_oob, loss_ib);
}

// The two halves of the above: `launch` packs the candidates, sends them and enqueues the kernels and the read-back of the
// results -- it returns while the device works; `collect` waits for the slot and forms the two scores per candidate.
// *acc_floor carries the comparison's running maximum from one collect to the next (candidates in their order).
void hibag_build_eval_launch(int slot, const PluginGenotype base_geno[], int n_snp, const HibagBuildCandidate cand[], int n_cand)
{
	const double t0 = batch_now();
	if (!g.active) build_throw("build_eval_batch before build_init");
	if (slot < 0 || slot > 1) build_throw("build_eval_launch: slot must be 0 or 1");
	if (n_snp < 1 || n_snp > 128 || n_cand < 0) build_throw("build_eval_batch: invalid sizes");
	BuildState::Slot &SL = g.slot[slot];
	SL.n_cand = n_cand;
	if (n_cand == 0) return;
	const int nh = g.n_hla, n = g.n_sample, np = g.n_pad;
	const int nw = (n_snp + 31) / 32, word = (n_snp - 1) >> 5, bit = (n_snp - 1) & 31;
	size_t H = 0;
	for (int c = 0; c < n_cand; c++) H += (size_t)cand[c].n_haplo;
	const size_t Hs = std::max<size_t>(H, 1);
	static const int wave_target = getenv("HIBAG_BATCH_WAVES") ? atoi(getenv("HIBAG_BATCH_WAVES")) : 8192;
	const int n_seg = std::max(1, std::min(64, wave_target / std::max(1, (np / HIBAG_WAVE) * n_cand)));

	// Pass 1 (small): per candidate the haplotypes per allele and the list of its non-empty cells, in posterior order.  The
	// scratch lives with the thread: a growth step is a millisecond, and allocating (and page-faulting) a few hundred KB of
	// vectors per step was a quarter of what the step cost the host.
	struct Scratch {
		std::vector<int> start, true_cell, at, present;
		std::vector<std::vector<int>> cell_list;
		std::vector<std::vector<uint64_t>> cell_work;
		std::vector<uint32_t> base_w1, base_w2;
	};
	static thread_local Scratch S;
	S.start.assign((size_t)n_cand * (nh + 1), 0);
	if ((int)S.cell_list.size() < n_cand) { S.cell_list.resize(n_cand); S.cell_work.resize(n_cand); }
	int max_cells = 1;
	{
		size_t off = 0;
		for (int c = 0; c < n_cand; c++) {
			const PluginHaplotype *hp = cand[c].haplo;
			int *st = &S.start[(size_t)c * (nh + 1)];
			for (int i = 0; i < cand[c].n_haplo; i++) {
				const int a = hp[i].aux.hla_allele;
				if (a < 0 || a >= nh) build_throw("haplotype with an invalid HLA allele index");
				st[a + 1]++;
			}
			st[0] = (int)off;
			for (int h = 0; h < nh; h++) st[h + 1] += st[h];
			off += (size_t)cand[c].n_haplo;
			std::vector<int> &cl = S.cell_list[c];
			std::vector<uint64_t> &cwk = S.cell_work[c];
			cl.clear(); cwk.clear();
			// (posterior order = h1 ascending, h2 >= h1 ascending, over the alleles that have haplotypes: src/LibHLA.cpp:1653-1691)
			S.present.clear();
			for (int h = 0; h < nh; h++) if (st[h + 1] > st[h]) S.present.push_back(h);
			for (size_t i1 = 0; i1 < S.present.size(); i1++) {
				const int h1 = S.present[i1];
				const uint64_t n1 = (uint64_t)(st[h1 + 1] - st[h1]);
				for (size_t i2 = i1; i2 < S.present.size(); i2++) {
					const int h2 = S.present[i2];
					const uint64_t n2 = (uint64_t)(st[h2 + 1] - st[h2]);
					cl.push_back((h1 << 16) | h2);
					cwk.push_back(h1 == h2 ? n1 * (n1 + 1) / 2 : n1 * n2);
				}
			}
			max_cells = std::max(max_cells, (int)cl.size());
		}
	}
	max_cells = (max_cells + SCAN_NB - 1) / SCAN_NB * SCAN_NB;

	// One device arena = [inputs | outputs | scratch]; the inputs travel in ONE copy from a pinned
	// staging buffer and the outputs come back in one: a growth step is a handful of small arrays,
	// and a dozen separate pageable copies cost more than the kernels.
	size_t o = 0;
	auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
	// (the driver's trainers keep the cohort's genotypes on the device: no bit planes per candidate, no range table per cell,
	// no per-sample position table -- BatchView "Round 6")
	bool use_dev = g.d_geno8 != nullptr;
	for (int c = 0; c < n_cand && use_dev; c++) use_dev = cand[c].snp >= 0 && cand[c].snp < g.geno8_snps;
	const size_t o_hb = take((size_t)nw * Hs * 4), o_cw = take(use_dev ? (size_t)n_cand * 4 : (size_t)n_cand * 2 * np * 4), o_start = take(S.start.size() * 4),
		o_cells = take((size_t)n_cand * max_cells * 4), o_cellb = take(use_dev ? 0 : (size_t)n_cand * max_cells * 16), o_seg = take((size_t)n_cand * (n_seg + 1) * 4),
		o_planes = take((size_t)2 * NW * np * 4), o_true = take((size_t)np * 4), o_wpos = take(use_dev ? (size_t)np * 4 : (size_t)n_cand * np * 4), o_hf = take(Hs * 8), in_end = o;
	const size_t b_best = (size_t)n_cand * 2 * np * 4, b_post = (size_t)n_cand * np * 8;
	const size_t o_best = take(b_best), o_post = take(b_post), out_end = o;
	const size_t o_cellv = take((size_t)n_cand * max_cells * np * 8);
	const double t_res0 = batch_now();
	reserve(SL.d, SL.cap_d, o, "hipMalloc(batch)");
	if (out_end > SL.cap_h) {
		if (SL.h) (void)hipHostFree(SL.h);
		SL.h = nullptr; SL.cap_h = 0;
		HIP_OK(hipHostMalloc(&SL.h, out_end * 2, hipHostMallocDefault), "hipHostMalloc(staging)");
		SL.cap_h = out_end * 2;
	}
	const double t_res1 = batch_now();
	char *d = (char *)SL.d, *h = (char *)SL.h;

	// Pass 2: every input array written where it travels from (the pinned staging area), once.
	uint32_t *const hb = (uint32_t *)(h + o_hb), *const cw = (uint32_t *)(h + o_cw), *const planes = (uint32_t *)(h + o_planes);
	double *const hf = (double *)(h + o_hf);
	int *const cells = (int *)(h + o_cells), *const cellb = (int *)(h + o_cellb), *const seg = (int *)(h + o_seg),
		*const true_cell = (int *)(h + o_true), *const wpos = (int *)(h + o_wpos);
	memcpy(h + o_start, S.start.data(), S.start.size() * 4);
	if (H == 0) { for (int w = 0; w < nw; w++) hb[w] = 0; hf[0] = 0.0; }
	S.base_w1.resize(np); S.base_w2.resize(np);                      // word `word` of every sample's base genotype
	for (int s = 0; s < n; s++) {
		S.base_w1[s] = (uint32_t)((uint64_t)base_geno[s].snp1[word >> 1] >> (32 * (word & 1)));
		S.base_w2[s] = (uint32_t)((uint64_t)base_geno[s].snp2[word >> 1] >> (32 * (word & 1)));
	}
	{
		size_t off = 0;
		for (int c = 0; c < n_cand; c++) {
			const PluginHaplotype *hp = cand[c].haplo;
			for (int i = 0; i < cand[c].n_haplo; i++) {
				for (int w = 0; w < nw; w++) {
					uint32_t v = (uint32_t)((uint64_t)hp[i].packed[w >> 1] >> (32 * (w & 1)));
					const int lo = 32 * w;
					if (n_snp < lo + 32) v &= (n_snp <= lo) ? 0u : ((1u << (n_snp - lo)) - 1);
					hb[(size_t)w * Hs + off + i] = v;
				}
				hf[off + i] = hp[i].freq;
			}
			off += (size_t)cand[c].n_haplo;
			if (use_dev) cw[c] = (uint32_t)cand[c].snp;                      // (the slot holds the candidates' SNP indices instead)
			else {
			// the candidate SNP's bit in the two planes of its word (branch-free: the compiler vectorises it)
			uint32_t *o1 = &cw[((size_t)c * 2) * np], *o2 = &cw[((size_t)c * 2 + 1) * np];
			const int32_t *col = cand[c].column;
			const uint32_t keep = ~(1u << bit);
			const uint32_t *b1p = S.base_w1.data(), *b2p = S.base_w2.data();
			for (int s = 0; s < n; s++) {
				const uint32_t v = (uint32_t)col[s];
				const uint32_t b1 = (uint32_t)(v == 1u) | (uint32_t)(v == 2u), b2 = (uint32_t)(v > 1u);   // TGenotype::_SetSNP, src/LibHLA.cpp:609-622
				o1[s] = (b1p[s] & keep) | (b1 << bit);
				o2[s] = (b2p[s] & keep) | (b2 << bit);
			}
			for (int s = n; s < np; s++) { o1[s] = 0u; o2[s] = 0xFFFFFFFFu; }      // padding lanes: all missing
			}
			// the cell list, the cells' haplotype ranges, the segments of equal work
			const int *st = &S.start[(size_t)c * (nh + 1)];
			const std::vector<int> &cl = S.cell_list[c];
			const std::vector<uint64_t> &cwk = S.cell_work[c];
			int *cc = cells + (size_t)c * max_cells, *cbv = cellb + (size_t)c * max_cells * 4;
			if (use_dev) {
				memcpy(cc, cl.data(), cl.size() * sizeof(int));
				for (size_t i = cl.size(); i < (size_t)max_cells; i++) cc[i] = -1;     // (never equal to a sample's packed true pair)
			} else {
				for (size_t i = 0; i < cl.size(); i++) {
					const int h1 = cl[i] >> 16, h2 = cl[i] & 0xFFFF;
					cc[i] = cl[i];
					cbv[4 * i] = st[h1]; cbv[4 * i + 1] = st[h1 + 1]; cbv[4 * i + 2] = st[h2]; cbv[4 * i + 3] = st[h2 + 1];
				}
				for (size_t i = cl.size(); i < (size_t)max_cells; i++) { cc[i] = 0; cbv[4 * i] = cbv[4 * i + 1] = cbv[4 * i + 2] = cbv[4 * i + 3] = 0; }
			}
			uint64_t total = 0;
			for (uint64_t w : cwk) total += w + 4;
			int *sg = seg + (size_t)c * (n_seg + 1);
			sg[0] = 0;
			uint64_t acc = 0;
			int k = 1;
			for (size_t i = 0; i < cl.size(); i++) {                        // equal work per segment, contiguous cells
				while (k < n_seg && acc * n_seg >= total * k) sg[k++] = (int)i;
				acc += cwk[i] + 4;
			}
			while (k <= n_seg) sg[k++] = (int)cl.size();
		}
	}

	// genotype planes and true pairs of the cohort (what upload_geno does, but into the same transfer)
	g.true1.assign(n, 0); g.true2.assign(n, 0);
	for (int w = 0; w < NW; w++) {
		uint32_t *p1 = planes + (size_t)w * np, *p2 = planes + (size_t)(NW + w) * np;
		for (int s = 0; s < n; s++) {
			p1[s] = (uint32_t)((uint64_t)base_geno[s].snp1[w >> 1] >> (32 * (w & 1)));
			p2[s] = (uint32_t)((uint64_t)base_geno[s].snp2[w >> 1] >> (32 * (w & 1)));
		}
		for (int s = n; s < np; s++) { p1[s] = 0u; p2[s] = 0xFFFFFFFFu; }
	}
	for (int s = 0; s < n; s++) {
		int a1 = base_geno[s].hla1, a2 = base_geno[s].hla2;
		if (a1 > a2) std::swap(a1, a2);
		if (a1 < 0 || a2 >= nh) build_throw("genotype with an invalid true HLA pair");
		g.true1[s] = a1; g.true2[s] = a2;
		true_cell[s] = a2 + a1 * (2 * nh - a1 - 1) / 2;
	}
	for (int s = n; s < np; s++) true_cell[s] = -1;

	// where each sample's true pair sits in each candidate's cell list -- or just the packed pair, which the scan compares itself
	if (use_dev) {
		for (int s = 0; s < n; s++) wpos[s] = (g.true1[s] << 16) | g.true2[s];
		for (int s = n; s < np; s++) wpos[s] = -1;
	}
	S.at.resize((size_t)nh * (nh + 1) / 2);
	for (int c = 0; c < n_cand && !use_dev; c++) {
		std::fill(S.at.begin(), S.at.end(), -1);
		const std::vector<int> &cl = S.cell_list[c];
		for (size_t i = 0; i < cl.size(); i++) {
			const int h1 = cl[i] >> 16, h2 = cl[i] & 0xFFFF;
			S.at[h2 + h1 * (2 * nh - h1 - 1) / 2] = (int)i;
		}
		int *wp = wpos + (size_t)c * np;
		for (int s = 0; s < n; s++) wp[s] = S.at[true_cell[s]];
		for (int s = n; s < np; s++) wp[s] = -1;
	}
	const double t1 = batch_now();
	g_batch_prof[4] += t1 - t_res1; g_batch_prof[5] += t_res1 - t_res0;
	BatchView B{nh, np, nw, n_cand, n_seg, word, (const uint32_t *)(d + o_hb), (const double *)(d + o_hf), (int)Hs,
		(const int *)(d + o_start), (const uint32_t *)(d + o_planes), (const uint32_t *)(d + o_cw), (const int *)(d + o_cells),
		(const int4 *)(d + o_cellb), (const int *)(d + o_seg), max_cells, (const int *)(d + o_true), (const int *)(d + o_wpos), (const double *)g.d_tab, (double *)(d + o_cellv),
		(int *)(d + o_best), (double *)(d + o_post), nullptr, nullptr, bit, nullptr};
	if (use_dev) {
		B.gdev = (const int8_t *)g.d_geno8; B.cand_snp = (const int *)(d + o_cw); B.true_pair = (const int *)(d + o_wpos);
		B.cand_w = nullptr; B.cellb = nullptr; B.wpos = nullptr;
	}
	// one operation (hibag_combine.h): the step's inputs in one copy, the two kernels -- alone, or fused with the scoring of
	// the other trainers' steps of the moment --, the results back in one; returns when they are in the staging area
	HibagOp op;
	op.kind = HIBAG_OP_EVAL;
	op.view = &B;
	op.up.push_back(HibagCopy{d, h, in_end});
	op.down.push_back(HibagCopy{h + o_best, d + o_best, out_end - o_best});
	hibag_combine_run(op);
	const double t2 = batch_now();
	SL.np = np; SL.o_best = o_best; SL.o_post = o_post; SL.t_launch = t2;
	g_batch_prof[0] += t1 - t0; g_batch_prof[1] += t2 - t1;
}

void hibag_build_eval_collect(int slot, int *acc_floor, int acc_oob[], double loss_ib[])
{
	if (slot < 0 || slot > 1) build_throw("build_eval_collect: slot must be 0 or 1");
	BuildState::Slot &SL = g.slot[slot];
	const int n_cand = SL.n_cand, np = SL.np;
	if (n_cand == 0) return;
	const double t2 = batch_now();                 // (the slot's operation was waited for by hibag_build_eval_launch)
	const char *h = (const char *)SL.h;
	const int *best = (const int *)(h + SL.o_best);
	const double *post = (const double *)(h + SL.o_post);
	const double t3 = batch_now();
	int run_max = *acc_floor;
	for (int c = 0; c < n_cand; c++) {
		int correct = 0;                                               // build_acc_oob
		for (int s : g.oob)
			correct += compare_hla(best[((size_t)c * 2) * np + s], best[((size_t)c * 2 + 1) * np + s], g.true1[s], g.true2[s]);
		acc_oob[c] = correct;
		loss_ib[c] = 0;
		if (correct < run_max) continue;                               // its loss is never looked at (src/LibHLA.cpp:2033-2034)
		if (correct > run_max) run_max = correct;
		double loglik = 0;                                             // build_acc_ib
		for (int s : g.inbag) loglik += g.boot[s] * std::log(post[(size_t)c * np + s]);
		loss_ib[c] = loglik * -2;
	}
	*acc_floor = run_max;
	g.evaluated = false;
	g_batch_prof[2] += t3 - t2; g_batch_prof[3] += batch_now() - t3;
}
