// hibag_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for HIBAG's
// attribute-bagging prediction hot path: the haplotype-pair posterior loop of
// CAlg_Prediction::_PostProb2 (src/LibHLA.cpp:1769-1830) and the ensemble step
// of CAttrBag_Model::_PredictHLA / PredictHLA (src/LibHLA.cpp:2317-2482).
//
// Mapping: LANE = SAMPLE.  A wavefront holds 64 samples and walks the model's
// loop nest (classifier -> allele pair -> haplotype pair) in the reference's
// order.  The nest depends only on the model, so control flow is wave-uniform,
// every haplotype word / frequency is a scalar (SMEM) load, and each lane
// reproduces the reference's rounding sequence for its own sample: results are
// bit-identical to the CPU kernels by construction, with no cross-lane
// reduction anywhere on the numeric path.
//
// The normalisation 1/sum of a classifier's posterior needs all of its cells,
// and holding 64 samples x P cells does not fit on chip, so the pair loop runs
// twice: pass 1 (k_total) produces the in-order total per (sample, classifier),
// pass 2 (k_accum) recomputes each cell, scales it and adds it to the ensemble
// sum kept in LDS.  Recomputing is cheaper than spilling 8*P bytes per
// (sample, classifier) to HBM (DESIGN.md "Why two passes").
//
// No MFMA: the pair weight 1e-5^d(i,j) does not factor over (i,j) at
// heterozygous SNPs, so there is no contraction to feed a matrix core.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA fusion: the
// reference multiplies and adds with separate roundings).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hibag_device.h"
#include "hibag_kernels.h"

#define NA_INTEGER (-2147483647 - 1)

// ---------------------------------------------------------------------------
// Per-lane genotype of one classifier, re-encoded from the (S1,S2) bit planes
// into three disjoint masks so that the distance of src/LibHLA.cpp:747-819
//     d = popc((H1^S1)&MASK) + popc((H2^S2)&MASK),
//     MASK = ((H1^S2)|(H2^S1)) & ~(S2&~S1)
// becomes, SNP by SNP (g=0: h1+h2, g=2: 2-h1-h2, g=1: [h1==h2], missing: 0),
//     d = popc((H1^T)&ZT) + popc((H2^T)&ZT) + popc(~(H1^H2)&E)
// with ZT = homozygous, T = g==2, E = heterozygous.  Same integer, fewer VALU
// ops, and ~(H1^H2) is wave-uniform so it runs on the scalar unit.
template <int NW>
struct LaneGeno {
	uint32_t zt[NW], t[NW], e[NW];
	int n_het;
};

template <int NW>
__device__ __forceinline__ void load_geno(const HibagBatchView &B, int row0, int s, LaneGeno<NW> &G)
{
	G.n_het = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		const uint32_t s1 = B.planes[(size_t)(row0 + 2 * w) * B.n_pad + s];
		const uint32_t s2 = B.planes[(size_t)(row0 + 2 * w + 1) * B.n_pad + s];
		G.zt[w] = ~(s1 ^ s2);
		G.t[w] = s1 & s2;
		G.e[w] = s1 & ~s2;
		G.n_het += __popc(G.e[w]);
	}
}

// Sum of one allele-pair cell in the reference's order
// (src/LibHLA.cpp:1781-1797 diagonal, :1804-1816 off-diagonal):
//   diagonal:  for a: cell += (f_a*f_a)*TAB[d(a,a)]; for b>a: cell += ((2 f_a)*f_b)*TAB[d(a,b)]
//   otherwise: for a in h1, b in h2: cell += ((2 f_a)*f_b)*TAB[d(a,b)]
// hb: this classifier's haplotype words [NW][H]; hf: frequencies; all uniform.
template <int NW>
__device__ __forceinline__ double cell_value(const uint32_t *__restrict__ hb,
	const double *__restrict__ hf, int H, int a0, int a1, int b0, int b1, bool diagonal,
	const LaneGeno<NW> &G, const double *tab_s)
{
	double cell = 0;
	for (int a = a0; a < a1; a++) {
		uint32_t A[NW];
		int ca = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) {
			A[w] = hb[w * H + a];
			ca += __popc((A[w] ^ G.t[w]) & G.zt[w]);
		}
		const double fa = hf[a];
		int b = b0;
		if (diagonal) {
			cell += (fa * fa) * tab_s[2 * ca + G.n_het];
			b = a + 1;
		}
		const double ff = 2 * fa;
		for (; b < b1; b++) {
			int d = ca;
#pragma unroll
			for (int w = 0; w < NW; w++) {
				const uint32_t Bw = hb[w * H + b];
				const uint32_t same = ~(A[w] ^ Bw);   // uniform -> SALU
				d += __popc((Bw ^ G.t[w]) & G.zt[w]) + __popc(same & G.e[w]);
			}
			cell += (ff * hf[b]) * tab_s[d];
		}
	}
	return cell;
}

__device__ __forceinline__ void stage_table(const HibagModelView &M, double *tab_s)
{
	for (int i = threadIdx.x; i < HIBAG_TAB_N; i += blockDim.x) tab_s[i] = M.tab[i];
	__syncthreads();
}

// ---------------------------------------------------------------------------
// k_pack: TGenotype::IntToSNP (src/LibHLA.cpp:662-706) for every (sample,
// classifier) plus the classifier weight from missingness
// (src/LibHLA.cpp:2418-2431).  grid (n_pad/64, C), block 64, lane = sample.
__global__ __launch_bounds__(HIBAG_WAVE) void k_pack(HibagModelView M, HibagBatchView B,
	const int32_t *__restrict__ geno)
{
	const int c = blockIdx.y;
	const int s = blockIdx.x * HIBAG_WAVE + threadIdx.x;
	const bool live = s < B.n_samp;
	const int k = M.n_snp_c[c];
	const int nw = M.n_word[c];
	const int *__restrict__ idx = M.snp_index + M.snp_off[c];
	const int32_t *__restrict__ row = geno + (size_t)(live ? s : 0) * M.n_snp;
	const int row0 = M.geno_row[c];
	int num = 0, den = 0;
	for (int w = 0; w < nw; w++) {
		uint32_t p1 = 0, p2 = 0xFFFFFFFFu;      // all missing: (S1,S2) = (0,1)
		const int lim = min(32, k - 32 * w);
		for (int j = 0; j < lim; j++) {
			const int snp = idx[32 * w + j];
			const int wt = M.snp_weight[snp];
			den += wt;
			const int g = live ? row[snp] : -1;
			const uint32_t bit = 1u << j;
			if (g >= 0 && g <= 2) {
				num += wt;
				if (g >= 1) p1 |= bit;
				if (g <= 1) p2 &= ~bit;
			}
		}
		B.planes[(size_t)(row0 + 2 * w) * B.n_pad + s] = p1;
		B.planes[(size_t)(row0 + 2 * w + 1) * B.n_pad + s] = p2;
	}
	B.cw[(size_t)c * B.n_pad + s] = (live && den > 0) ? ((double)num / den) : 0.0;
}

// k_unpack_tgeno: plugin path (predict_avg_prob): the host already packed one
// sample per classifier as TGenotype (48 bytes: int64 S1[2], S2[2], 16 bytes
// of book-keeping, inst/include/LibHLA_ext.h:311-352) and computed the
// weights.  One thread per classifier writes lane 0 of the planes; lanes 1..63
// are padding (missing, weight 0).
__global__ void k_unpack_tgeno(HibagModelView M, HibagBatchView B,
	const uint64_t *__restrict__ tgeno, const double *__restrict__ weight)
{
	const int c = blockIdx.x;
	const int lane = threadIdx.x;
	const int nw = M.n_word[c];
	const int row0 = M.geno_row[c];
	const uint64_t *g = tgeno + (size_t)c * 6;
	for (int w = 0; w < nw; w++) {
		uint32_t p1 = 0, p2 = 0xFFFFFFFFu;
		if (lane == 0) {
			p1 = (uint32_t)(g[w >> 1] >> (32 * (w & 1)));
			p2 = (uint32_t)(g[2 + (w >> 1)] >> (32 * (w & 1)));
		}
		B.planes[(size_t)(row0 + 2 * w) * B.n_pad + lane] = p1;
		B.planes[(size_t)(row0 + 2 * w + 1) * B.n_pad + lane] = p2;
	}
	B.cw[(size_t)c * B.n_pad + lane] = (lane == 0) ? weight[c] : 0.0;
}

// ---------------------------------------------------------------------------
// k_total (pass 1): in-order posterior total of one classifier for 64 samples:
// cells visited h1 ascending, h2 >= h1 ascending and added as produced
// (src/LibHLA.cpp:1776-1826).  Empty cells add +0.0 and are skipped.
// grid (n_pad/64, C) with the heaviest classifiers first, block 64.
template <int NW>
__device__ __forceinline__ double classifier_total(const HibagModelView &M, const HibagBatchView &B,
	int c, int s, const double *tab_s)
{
	LaneGeno<NW> G;
	load_geno<NW>(B, M.geno_row[c], s, G);
	const int H = M.n_hap[c];
	const uint32_t *__restrict__ hb = M.hbits + M.bits_off[c];
	const double *__restrict__ hf = M.hfreq + M.hap_off[c];
	const int *__restrict__ st = M.hla_start + (size_t)c * (M.n_hla + 1);
	const int n_hla = M.n_hla;
	double total = 0;
	for (int h1 = 0; h1 < n_hla; h1++) {
		const int a0 = st[h1], a1 = st[h1 + 1];
		if (a0 == a1) continue;
		total += cell_value<NW>(hb, hf, H, a0, a1, a0, a1, true, G, tab_s);
		for (int h2 = h1 + 1; h2 < n_hla; h2++) {
			const int b0 = st[h2], b1 = st[h2 + 1];
			if (b0 == b1) continue;
			total += cell_value<NW>(hb, hf, H, a0, a1, b0, b1, false, G, tab_s);
		}
	}
	return total;
}

__global__ __launch_bounds__(HIBAG_WAVE) void k_total(HibagModelView M, HibagBatchView B)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	stage_table(M, tab_s);
	const int c = M.c_order[blockIdx.y];
	const int s = blockIdx.x * HIBAG_WAVE + threadIdx.x;
	const size_t at = (size_t)c * B.n_pad + s;
	const bool active = B.cw[at] > 0;                 // src/LibHLA.cpp:2451
	if (__ballot(active) == 0) return;                // nobody needs this classifier
	double total;
	switch (M.n_word[c]) {
	case 1:  total = classifier_total<1>(M, B, c, s, tab_s); break;
	case 2:  total = classifier_total<2>(M, B, c, s, tab_s); break;
	case 3:  total = classifier_total<3>(M, B, c, s, tab_s); break;
	default: total = classifier_total<4>(M, B, c, s, tab_s); break;
	}
	B.tot[at] = total;
	B.inv[at] = 1 / total;                            // src/LibHLA.cpp:1827 (inf when total == 0)
}

// ---------------------------------------------------------------------------
// k_accum (pass 2): for a tile of allele-pair cells and 64 samples, walk the
// classifiers in order and do  S[p] += (cell * (1/total)) * w
// (src/LibHLA.cpp:1828 then :1497-1507) with S in LDS.  A classifier whose
// cell is structurally empty contributes (0*inv)*w = +0 unless inv is not
// finite (total == 0 or denormal), in which case the reference yields NaN/inf;
// the `poison` ballot keeps that case on the full path.
// grid n_tile * n_pad/64 (XCD-aware decode), block 64.
template <int NW, int T>
__device__ __forceinline__ void accumulate_classifier(const HibagModelView &M, const HibagBatchView &B,
	int c, int s, int tile, bool active, bool poison, double inv, double w,
	const double *tab_s, double (*acc)[HIBAG_WAVE])
{
	LaneGeno<NW> G;
	load_geno<NW>(B, M.geno_row[c], s, G);
	const int H = M.n_hap[c];
	const uint32_t *__restrict__ hb = M.hbits + M.bits_off[c];
	const double *__restrict__ hf = M.hfreq + M.hap_off[c];
	const int *__restrict__ st = M.hla_start + (size_t)c * (M.n_hla + 1);
	const int *__restrict__ cells = M.tile_cell + (size_t)tile * T;
	const int lane = threadIdx.x;
	for (int j = 0; j < T; j++) {
		const int p = cells[j];
		if (p < 0) break;
		const int h1 = M.cell_h1[p], h2 = M.cell_h2[p];
		const int a0 = st[h1], a1 = st[h1 + 1], b0 = st[h2], b1 = st[h2 + 1];
		if ((a0 == a1 || b0 == b1) && !poison) continue;
		const double cell = cell_value<NW>(hb, hf, H, a0, a1, b0, b1, h1 == h2, G, tab_s);
		if (active) acc[j][lane] += (cell * inv) * w;
	}
}

template <int T>
__global__ __launch_bounds__(HIBAG_WAVE) void k_accum(HibagModelView M, HibagBatchView B)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	__shared__ double acc[T][HIBAG_WAVE];
	stage_table(M, tab_s);

	// XCD-aware decode: workgroups are dealt round-robin over the 8 XCDs, so
	// give all tiles of one sample group the same (blockIdx % 8): the group's
	// planes / weights / totals are then fetched into one XCD's L2 only.
	const int n_group = B.n_pad / HIBAG_WAVE;
	const int b = blockIdx.x;
	int group, tile;
	{
		const int groups_full = n_group & ~7;          // groups covered by the swizzle
		if (b < groups_full * M.n_tile) {
			const int xcd = b & 7, j = b >> 3, jg = j / M.n_tile;
			group = jg * 8 + xcd; tile = j - jg * M.n_tile;
		} else {                                       // tail (< 8 groups): plain order
			const int r = b - groups_full * M.n_tile;
			group = groups_full + r / M.n_tile; tile = r % M.n_tile;
		}
	}
	const int lane = threadIdx.x;
	const int s = group * HIBAG_WAVE + lane;

#pragma unroll
	for (int j = 0; j < T; j++) acc[j][lane] = 0;

	for (int c = 0; c < M.n_classifier; c++) {
		const size_t at = (size_t)c * B.n_pad + s;
		const double w = B.cw[at];
		const bool active = w > 0;
		if (__ballot(active) == 0) continue;
		const double inv = B.inv[at];
		const bool poison = __ballot(active && !(fabs(inv) <= 1.79769313486231570815e+308)) != 0;
		switch (M.n_word[c]) {
		case 1:  accumulate_classifier<1, T>(M, B, c, s, tile, active, poison, inv, w, tab_s, acc); break;
		case 2:  accumulate_classifier<2, T>(M, B, c, s, tile, active, poison, inv, w, tab_s, acc); break;
		case 3:  accumulate_classifier<3, T>(M, B, c, s, tile, active, poison, inv, w, tab_s, acc); break;
		default: accumulate_classifier<4, T>(M, B, c, s, tile, active, poison, inv, w, tab_s, acc); break;
		}
	}

	const int *__restrict__ cells = M.tile_cell + (size_t)tile * T;
	for (int j = 0; j < T; j++) {
		const int p = cells[j];
		if (p < 0) break;
		B.part[(size_t)p * B.n_pad + s] = acc[j][lane];
	}
}

// ---------------------------------------------------------------------------
// k_vote_best (majority vote, pass 2 of vote_method = 2): per (sample,
// classifier) the first strict maximum of the NORMALISED posterior
// cell*(1/total) in cell order (src/LibHLA.cpp:2468 -> :1549-1566).
// grid (n_pad/64, C), block 64.  Writes the winning cell index or -1.
template <int NW>
__device__ __forceinline__ int classifier_best(const HibagModelView &M, const HibagBatchView &B,
	int c, int s, double inv, bool poison, const double *tab_s)
{
	LaneGeno<NW> G;
	load_geno<NW>(B, M.geno_row[c], s, G);
	const int H = M.n_hap[c];
	const uint32_t *__restrict__ hb = M.hbits + M.bits_off[c];
	const double *__restrict__ hf = M.hfreq + M.hap_off[c];
	const int *__restrict__ st = M.hla_start + (size_t)c * (M.n_hla + 1);
	const int n_hla = M.n_hla;
	double best = 0;
	int best_p = -1, p = 0;
	for (int h1 = 0; h1 < n_hla; h1++) {
		const int a0 = st[h1], a1 = st[h1 + 1];
		if (a0 == a1 && !poison) { p += n_hla - h1; continue; }
		for (int h2 = h1; h2 < n_hla; h2++, p++) {
			const int b0 = st[h2], b1 = st[h2 + 1];
			if ((a0 == a1 || b0 == b1) && !poison) continue;   // prob = +0 never beats max >= 0
			const double prob = cell_value<NW>(hb, hf, H, a0, a1, b0, b1, h1 == h2, G, tab_s) * inv;
			if (best < prob) { best = prob; best_p = p; }
		}
	}
	return best_p;
}

__global__ __launch_bounds__(HIBAG_WAVE) void k_vote_best(HibagModelView M, HibagBatchView B, int *__restrict__ best_cell)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	stage_table(M, tab_s);
	const int c = M.c_order[blockIdx.y];
	const int s = blockIdx.x * HIBAG_WAVE + threadIdx.x;
	const size_t at = (size_t)c * B.n_pad + s;
	const bool active = B.cw[at] > 0;
	if (__ballot(active) == 0) { best_cell[at] = -1; return; }
	const double inv = B.inv[at];
	const bool poison = __ballot(active && !(fabs(inv) <= 1.79769313486231570815e+308)) != 0;
	int bp;
	switch (M.n_word[c]) {
	case 1:  bp = classifier_best<1>(M, B, c, s, inv, poison, tab_s); break;
	case 2:  bp = classifier_best<2>(M, B, c, s, inv, poison, tab_s); break;
	case 3:  bp = classifier_best<3>(M, B, c, s, inv, poison, tab_s); break;
	default: bp = classifier_best<4>(M, B, c, s, inv, poison, tab_s); break;
	}
	best_cell[at] = active ? bp : -1;
}

// k_vote_tally: one-hot votes with weight 1.0 (src/LibHLA.cpp:2465-2475);
// counts are small integers, exact in any order.  thread = sample.
__global__ void k_vote_tally(HibagModelView M, HibagBatchView B, const int *__restrict__ best_cell)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= B.n_pad) return;
	for (int p = 0; p < M.n_cell; p++) B.part[(size_t)p * B.n_pad + s] = 0;
	for (int c = 0; c < M.n_classifier; c++) {
		const int p = best_cell[(size_t)c * B.n_pad + s];
		if (p >= 0) B.part[(size_t)p * B.n_pad + s] += 1.0;
	}
}

// ---------------------------------------------------------------------------
// k_scalars: per-sample ensemble scalars, classifiers in order:
//   part[P]   = sum of weights   (_Sum_Weight, src/LibHLA.cpp:1505; for the
//               majority vote the number of classifiers that produced a call)
//   part[P+1] = sum_matching = sum_c total_c * w_c        (:2458)
//   part[P+2] = num_matching = sum_c w_c                  (:2459)
__global__ void k_scalars(HibagModelView M, HibagBatchView B, const int *__restrict__ best_cell)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= B.n_pad) return;
	double sum_w = 0, sum_m = 0, num_m = 0;
	for (int c = 0; c < M.n_classifier; c++) {
		const size_t at = (size_t)c * B.n_pad + s;
		const double w = B.cw[at];
		if (!(w > 0)) continue;
		sum_m += B.tot[at] * w;
		num_m += w;
		if (best_cell) { if (best_cell[at] >= 0) sum_w += 1.0; }
		else sum_w += w;
	}
	const size_t P = (size_t)M.n_cell;
	B.part[(P + 0) * B.n_pad + s] = sum_w;
	B.part[(P + 1) * B.n_pad + s] = sum_m;
	B.part[(P + 2) * B.n_pad + s] = num_m;
}

// ---------------------------------------------------------------------------
// k_finish_call: NormalizeSumPostProb (src/LibHLA.cpp:1509-1518) in place,
// then BestGuessEnsemble (:1549-1566: first strict maximum, NA when nothing is
// positive), the called pair's probability (:2376-2382) and the matching
// proportion (:2480).  thread = sample; rows of `part` are coalesced.
__global__ void k_finish_call(HibagModelView M, HibagBatchView B, double *__restrict__ part,
	int32_t *__restrict__ H1, int32_t *__restrict__ H2, double *__restrict__ max_prob,
	double *__restrict__ matching)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= B.n_pad) return;
	const size_t P = (size_t)M.n_cell, np = (size_t)B.n_pad;
	const double sum_w = part[(P + 0) * np + s];
	const bool scale = sum_w > 0;
	const double ff = 1.0 / sum_w;
	double best = 0;
	int b1 = NA_INTEGER, b2 = NA_INTEGER;
	size_t p = 0;
	for (int h1 = 0; h1 < M.n_hla; h1++) {
		for (int h2 = h1; h2 < M.n_hla; h2++, p++) {
			double v = part[p * np + s];
			if (scale) { v *= ff; part[p * np + s] = v; }
			if (best < v) { best = v; b1 = h1; b2 = h2; }
		}
	}
	if (s < B.n_samp) {
		if (H1) { H1[s] = b1; H2[s] = b2; }
		if (max_prob) max_prob[s] = (b1 != NA_INTEGER) ? best : 0.0;
		if (matching) matching[s] = part[(P + 1) * np + s] / part[(P + 2) * np + s];
	}
}

// k_finish_dosage: expected allele dosage (src/LibHLA.cpp:2387-2402).  The
// reference scatters each cell into d[h1] and d[h2] while scanning cells in
// order; gathered per allele h that is  S[0,h], S[1,h], ..., then 2*S[h,h],
// then S[h,h+1], ...  added in that order.  thread = (sample, allele).
__global__ void k_finish_dosage(HibagModelView M, HibagBatchView B, const double *__restrict__ part,
	double *__restrict__ dosage)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	const int h = blockIdx.y;
	if (s >= B.n_samp) return;
	const int n = M.n_hla;
	const size_t np = (size_t)B.n_pad;
	double d = 0;
	for (int g = 0; g < h; g++) {
		const size_t p = (size_t)h + (size_t)g * (2 * n - g - 1) / 2;   // index of (g,h), src/LibHLA.cpp:1523
		d += part[p * np + s];
	}
	size_t p = (size_t)h + (size_t)h * (2 * n - h - 1) / 2;
	d += 2 * part[p * np + s];
	for (int g = h + 1; g < n; g++) { p++; d += part[p * np + s]; }
	dosage[(size_t)s * n + h] = d;
}

// k_finish_prob: posterior matrix out, [n_samp][P] sample-major
// (src/LibHLA.cpp:2403-2406); 64x64 transpose through LDS so that both the
// read of part[p][s] and the write of postprob[s][p] are coalesced.
__global__ __launch_bounds__(256) void k_finish_prob(HibagModelView M, HibagBatchView B,
	const double *__restrict__ part, double *__restrict__ postprob)
{
	__shared__ double tile[64][65];
	const int s0 = blockIdx.x * 64, p0 = blockIdx.y * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int P = M.n_cell;
	for (int r = ty; r < 64; r += 4) {
		const int p = p0 + r;
		tile[r][tx] = (p < P) ? part[(size_t)p * B.n_pad + s0 + tx] : 0.0;
	}
	__syncthreads();
	for (int r = ty; r < 64; r += 4) {
		const int s = s0 + r, p = p0 + tx;
		if (s < B.n_samp && p < P) postprob[(size_t)s * P + p] = tile[tx][r];
	}
}

// ---------------------------------------------------------------------------
// launchers (host side, no synchronisation, no allocation)

static inline dim3 grid1(int n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

void hibag_launch_pack(const HibagModelView &M, const HibagBatchView &B, const int32_t *d_geno, hipStream_t st)
{
	if (M.n_classifier == 0) return;
	hipLaunchKernelGGL(k_pack, dim3(B.n_pad / HIBAG_WAVE, M.n_classifier), dim3(HIBAG_WAVE), 0, st, M, B, d_geno);
}

void hibag_launch_unpack_tgeno(const HibagModelView &M, const HibagBatchView &B, const uint64_t *d_tgeno,
	const double *d_weight, hipStream_t st)
{
	if (M.n_classifier == 0) return;
	hipLaunchKernelGGL(k_unpack_tgeno, dim3(M.n_classifier), dim3(HIBAG_WAVE), 0, st, M, B, d_tgeno, d_weight);
}

void hibag_launch_total(const HibagModelView &M, const HibagBatchView &B, hipStream_t st)
{
	if (M.n_classifier == 0) return;
	hipLaunchKernelGGL(k_total, dim3(B.n_pad / HIBAG_WAVE, M.n_classifier), dim3(HIBAG_WAVE), 0, st, M, B);
}

void hibag_launch_accum(const HibagModelView &M, const HibagBatchView &B, hipStream_t st)
{
	const unsigned n = (unsigned)(B.n_pad / HIBAG_WAVE) * (unsigned)M.n_tile;
	if (n == 0) return;
	switch (M.tile_cells) {
	case 8:  hipLaunchKernelGGL(k_accum<8>, dim3(n), dim3(HIBAG_WAVE), 0, st, M, B); break;
	case 16: hipLaunchKernelGGL(k_accum<16>, dim3(n), dim3(HIBAG_WAVE), 0, st, M, B); break;
	default: hipLaunchKernelGGL(k_accum<32>, dim3(n), dim3(HIBAG_WAVE), 0, st, M, B); break;
	}
}

void hibag_launch_vote(const HibagModelView &M, const HibagBatchView &B, int *d_best_cell, hipStream_t st)
{
	if (M.n_classifier > 0)
		hipLaunchKernelGGL(k_vote_best, dim3(B.n_pad / HIBAG_WAVE, M.n_classifier), dim3(HIBAG_WAVE), 0, st,
			M, B, d_best_cell);
	hipLaunchKernelGGL(k_vote_tally, grid1(B.n_pad, 64), dim3(64), 0, st, M, B, (const int *)d_best_cell);
}

void hibag_launch_scalars(const HibagModelView &M, const HibagBatchView &B, const int *d_best_cell, hipStream_t st)
{
	hipLaunchKernelGGL(k_scalars, grid1(B.n_pad, 64), dim3(64), 0, st, M, B, d_best_cell);
}

void hibag_launch_finish(const HibagModelView &M, const HibagBatchView &B, double *d_part,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching,
	double *d_dosage, double *d_postprob, hipStream_t st)
{
	hipLaunchKernelGGL(k_finish_call, grid1(B.n_pad, 64), dim3(64), 0, st, M, B, d_part,
		d_H1, d_H2, d_max_prob, d_matching);
	if (d_dosage)
		hipLaunchKernelGGL(k_finish_dosage, dim3((B.n_pad + 63) / 64, M.n_hla), dim3(64), 0, st,
			M, B, (const double *)d_part, d_dosage);
	if (d_postprob)
		hipLaunchKernelGGL(k_finish_prob, dim3(B.n_pad / 64, (M.n_cell + 63) / 64), dim3(256), 0, st,
			M, B, (const double *)d_part, d_postprob);
}
