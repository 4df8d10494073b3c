// hibag_k_finish.h -- part of hibag_kernels.hip (included there, one translation unit: the walks are templates that inline into
// their kernels): k_scalars, k_nan_cells and the finish kernels: call, probability, matching, dosage, posterior matrix.
#ifndef HIBAG_K_FINISH_H_
#define HIBAG_K_FINISH_H_

// ---------------------------------------------------------------------------
// k_scalars: the per-sample ensemble scalars (ensemble_scalars above) where pass 2 is not k_accum -- the majority vote and
// models whose pass 2 only reads stored sums (k_accum_cells); k_accum's tile-0 workgroups form them themselves.
__global__ void k_scalars(HibagModelView M, HibagBatchView B, const int *__restrict__ best_cell)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= B.n_pad) return;
	ensemble_scalars(M, B, s, best_cell);
	poison_scalars_if_failed(M, B, s);
}

// k_nan_cells: the NaN terms of the structurally empty cells for the (sample, classifier) pairs pass 1 listed
// (note_infinite_reciprocal): S[p] += (0 * (1/total)) * w -- NaN absorbs, so the place of these terms in the order of the
// additions cannot show.  Launched behind pass 2; with an empty list (the normal case) every workgroup leaves after one load.
// Workgroup = tile, thread = one empty cell of the tile; a list that overflowed falls back to thread = sample, every
// classifier looked at.  (k_accum_cells, store_cells == 1, does this itself.)
__global__ __launch_bounds__(64) void k_nan_cells(HibagModelView M, HibagBatchView B)
{
	// (also the kernel behind k_accum that marks a batch whose hand-overs failed: k_accum forms the scalars itself)
	if ((int)(blockIdx.x * 64 + threadIdx.x) < B.n_pad) poison_scalars_if_failed(M, B, blockIdx.x * 64 + threadIdx.x);
	const uint32_t count = __hip_atomic_load(B.err_dev + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (count == 0) return;
	if (count <= HIBAG_NAN_CAP) {
		const int t = blockIdx.x;
		if (t >= M.n_tile) return;
		const unsigned long long *__restrict__ list = reinterpret_cast<const unsigned long long *>(B.err_dev + 4);
		for (uint32_t e = 0; e < count; e++) {
			const int c = (int)(list[e] >> 32), s = (int)(uint32_t)list[e];
			const size_t at = (size_t)c * B.n_pad + s;
			const double v = (0.0 * B.inv[at]) * B.cw[at];
			const uint32_t *__restrict__ meta = M.tile_meta + ((size_t)c * M.n_tile + t) * HIBAG_TILE_META;
			const int i = (int)meta[0] + (int)threadIdx.x;
			if (i < M.tile_n[t]) B.part[(size_t)(M.tile_p0[t] + (int)(meta[4 + i] >> 24)) * B.n_pad + s] += v;
			__syncthreads();                          // (two classifiers of one sample may meet in a cell)
		}
		return;
	}
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= B.n_pad) return;
	for (int c = 0; c < M.n_classifier; c++) {
		const size_t at = (size_t)c * B.n_pad + s;
		const double w = B.cw[at], inv = B.inv[at];
		if (!(w > 0) || fabs(inv) <= 1.79769313486231570815e+308) continue;
		const double v = (0.0 * inv) * w;
		for (int t = 0; t < M.n_tile; t++) {
			const uint32_t *__restrict__ meta = M.tile_meta + ((size_t)c * M.n_tile + t) * HIBAG_TILE_META;
			for (int i = (int)meta[0]; i < M.tile_n[t]; i++)
				B.part[(size_t)(M.tile_p0[t] + (int)(meta[4 + i] >> 24)) * B.n_pad + s] += v;
		}
	}
}

// ---------------------------------------------------------------------------
// The ensemble sums in `part` stay un-normalised; every consumer applies
// NormalizeSumPostProb (src/LibHLA.cpp:1509-1518: S *= 1/sum_w when sum_w > 0)
// on the fly, which rounds exactly like scaling in place first.
__device__ __forceinline__ double normalised(double v, bool scale, double ff) { return scale ? v * ff : v; }
// (The call and the dosage read the ensemble sums with the default cache policy: the second and third read come out of the
// caches.  Streamed (nt) they were measured 17 % slower for no gain elsewhere; k_finish_prob, the last reader, streams.)

// k_finish_call: BestGuessEnsemble (src/LibHLA.cpp:1549-1566: first strict
// maximum in cell order, NA when nothing is positive), the called pair's
// probability (:2376-2382) and the matching proportion (:2480).
// Block = 64 samples x FIN_SEG segments of the cell range; every thread scans
// its segment in order, then the segments are merged in order with the same
// strict comparison, which reproduces the sequential scan exactly.
#define FIN_SEG 16
__device__ __forceinline__ void finish_call(const HibagModelView &M, const HibagBatchView &B, int group,
	const double *__restrict__ part, int32_t *__restrict__ H1, int32_t *__restrict__ H2,
	double *__restrict__ max_prob, double *__restrict__ matching)
{
	__shared__ double best_s[FIN_SEG][64];
	__shared__ int cell_s[FIN_SEG][64];
	const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
	const int s = group * 64 + lane;
	const int P = M.n_cell;
	const size_t np = (size_t)B.n_pad;
	const double sum_w = part[(size_t)P * np + s];
	const bool scale = sum_w > 0;
	const double ff = 1.0 / sum_w;
	const int per = (P + FIN_SEG - 1) / FIN_SEG;
	const int lo = seg * per, hi = min(P, lo + per);
	double best = 0;
	int cell = -1;
	int p = lo;
	for (; p + 8 <= hi; p += 8) {                 // eight rows in flight, compared in cell order
		double v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = part[(size_t)(p + j) * np + s];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const double x = normalised(v[j], scale, ff);
			if (best < x) { best = x; cell = p + j; }
		}
	}
	for (; p < hi; p++) {
		const double v = normalised(part[(size_t)p * np + s], scale, ff);
		if (best < v) { best = v; cell = p; }
	}
	best_s[seg][lane] = best;
	cell_s[seg][lane] = cell;
	__syncthreads();
	if (seg != 0 || s >= B.n_samp) return;
	for (int g = 1; g < FIN_SEG; g++)
		if (best < best_s[g][lane]) { best = best_s[g][lane]; cell = cell_s[g][lane]; }
	if (sum_w != sum_w) { cell = -1; best = sum_w; }     // poisoned batch (k_scalars): NA call, NaN probability and matching
	int b1 = NA_INTEGER, b2 = NA_INTEGER;
	if (cell >= 0) {
		// invert p = h2 + h1*(2n-h1-1)/2 (src/LibHLA.cpp:1523)
		int h1 = 0, row = M.n_hla, rem = cell;
		while (rem >= row) { rem -= row; row--; h1++; }
		b1 = h1; b2 = h1 + rem;
	}
	if (H1) { H1[s] = b1; H2[s] = b2; }
	if (max_prob) max_prob[s] = (cell >= 0 || sum_w != sum_w) ? best : 0.0;
	if (matching) matching[s] = part[(size_t)(P + 1) * np + s] / part[(size_t)(P + 2) * np + s];
}

__global__ __launch_bounds__(64 * FIN_SEG) void k_finish_call(HibagModelView M, HibagBatchView B,
	const double *__restrict__ part, int32_t *__restrict__ H1, int32_t *__restrict__ H2,
	double *__restrict__ max_prob, double *__restrict__ matching)
{
	finish_call(M, B, blockIdx.x, part, H1, H2, max_prob, matching);
}

// finish_dosage: expected allele dosage (src/LibHLA.cpp:2387-2402).  The
// reference scatters each cell into d[h1] and d[h2] while scanning cells in
// order; gathered per allele h that is  S[0,h], S[1,h], ..., then 2*S[h,h],
// then S[h,h+1], ...  added in that order.  thread = (sample, allele).
__device__ __forceinline__ void finish_dosage(const HibagModelView &M, const HibagBatchView &B, int s, int h,
	const double *__restrict__ part, double *__restrict__ dosage)
{
	const int n = M.n_hla;
	if (s >= B.n_samp || h >= n) return;
	const size_t np = (size_t)B.n_pad;
	const double sum_w = part[(size_t)M.n_cell * np + s];
	const bool scale = sum_w > 0;
	const double ff = 1.0 / sum_w;
	double d = 0;
	// term g of allele h: the cell (g, h) for g < h, (h, g) for g >= h -- index p = h2 + h1 (2n - h1 - 1) / 2 (src/LibHLA.cpp:1523);
	// eight cells in flight, added in order (the diagonal cell counts twice)
	auto cell_of = [&](int g) {
		const int h1 = g < h ? g : h, h2 = g < h ? h : g;
		return (size_t)h2 + (size_t)h1 * (2 * n - h1 - 1) / 2;
	};
	int g = 0;
	for (; g + 8 <= n; g += 8) {
		double v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = part[cell_of(g + j) * np + s];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const double x = normalised(v[j], scale, ff);
			d += g + j == h ? 2 * x : x;
		}
	}
	for (; g < n; g++) {
		const double x = normalised(part[cell_of(g) * np + s], scale, ff);
		d += g == h ? 2 * x : x;
	}
	dosage[(size_t)s * n + h] = sum_w != sum_w ? sum_w : d;      // (NaN weight sum: poisoned batch, see k_scalars)
}

// k_finish: the call and the dosage in ONE launch -- two independent readers of the ensemble sums, which as two kernels ran
// one behind the other (20 + 33 us of the benchmark step's 1,340).  The first n_pad / 64 workgroups are k_finish_call's, the
// others take 64 samples x FIN_SEG alleles each.
__global__ __launch_bounds__(64 * FIN_SEG) void k_finish(HibagModelView M, HibagBatchView B,
	const double *__restrict__ part, int32_t *__restrict__ H1, int32_t *__restrict__ H2,
	double *__restrict__ max_prob, double *__restrict__ matching, double *__restrict__ dosage)
{
	const int n_group = B.n_pad / 64;
	if ((int)blockIdx.x < n_group) {
		finish_call(M, B, blockIdx.x, part, H1, H2, max_prob, matching);
	} else {
		const int j = (int)blockIdx.x - n_group;
		finish_dosage(M, B, (j % n_group) * 64 + (int)(threadIdx.x & 63), (j / n_group) * FIN_SEG + (int)(threadIdx.x >> 6), part, dosage);
	}
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
	const double sum_w = part[(size_t)P * B.n_pad + s0 + tx];
	const bool scale = sum_w > 0;
	const double ff = 1.0 / sum_w;
	for (int r = ty; r < 64; r += 4) {
		const int p = p0 + r;
		// (the last read of the sums, and a matrix as large as they are going out: both streamed -- with the default policy
		// the 100 MB written here pushed pass 1's rows and operands out of the caches, and pass 1 of the NEXT batch took 14 % longer)
		tile[r][tx] = (p < P) ? (sum_w != sum_w ? sum_w : normalised(__builtin_nontemporal_load(&part[(size_t)p * B.n_pad + s0 + tx]), scale, ff)) : 0.0;   // (NaN weight sum: poisoned batch)
	}
	__syncthreads();
	for (int r = ty; r < 64; r += 4) {
		const int s = s0 + r, p = p0 + tx;
		if (s < B.n_samp && p < P) __builtin_nontemporal_store(tile[tx][r], &postprob[(size_t)s * P + p]);
	}
}

#endif
