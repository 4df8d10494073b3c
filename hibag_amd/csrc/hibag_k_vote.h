// hibag_k_vote.h -- part of hibag_kernels.hip (included there, one translation unit: the walks are templates that inline into
// their kernels): the majority vote: k_vote_pick, k_vote_scan, k_vote_best_valu, k_vote_tally.
#ifndef HIBAG_K_VOTE_H_
#define HIBAG_K_VOTE_H_

// ---------------------------------------------------------------------------
// k_vote_best (majority vote, vote_method = 2): per (sample, classifier) the
// first strict maximum of the NORMALISED posterior cell*(1/total) in cell order
// (src/LibHLA.cpp:2468 -> :1549-1566).  Empty cells give +0 or NaN, neither of
// which can replace a maximum that starts at 0, so they are skipped.
// grid as k_total.  Writes the winning cell index or -1.
template <int NWP>
__device__ __forceinline__ int classifier_best(const HibagModelView &M, const HibagBatchView &B,
	int c, int s, double inv, const double *tab_s)
{
	LaneMask<NWP> L;
	load_masks<NWP>(B, M.mask_row[c], s, L);
	const uint32_t *__restrict__ cnt = M.cls_cnt + M.cls_off[c];
	const uint32_t *__restrict__ cell_p = M.cls_cell + M.cls_off[c];
	const uint32_t *__restrict__ cp = M.stream + M.stream_off[c];
	const int ncell = M.cls_n[c];
	double best = 0;
	int best_p = -1;
	for (int i = 0; i < ncell; i++) {
		const double prob = cell_sum<NWP>(cnt[i], cp, L, tab_s) * inv;
		if (best < prob) { best = prob; best_p = (int)cell_p[i]; }
	}
	return best_p;
}

// k_vote_best_valu: the majority vote's choice for the classifiers of the VALU engine (more than 112 SNPs) -- their pairs are
// walked a second time, with 1/total in hand.  (Every other classifier: k_vote_pick / k_vote_scan below, no second walk.)
// grid (group quads, classifiers); writes the winning cell index or -1.
__global__ __launch_bounds__(BLOCK_THREADS) void k_vote_best_valu(HibagModelView M, HibagBatchView B, int *__restrict__ best_cell)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	const int c = M.c_order[blockIdx.y];
	if (M.engine[c] != HIBAG_ENGINE_VALU) return;
	stage_table(M, tab_s);
	const int group = blockIdx.x * BLOCK_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (group * HIBAG_WAVE >= B.n_pad) return;
	const int s = group * HIBAG_WAVE + (threadIdx.x & 63);
	const size_t at = (size_t)c * B.n_pad + s;
	const bool active = B.cw[at] > 0;
	if (__ballot(active) == 0) { best_cell[at] = -1; return; }
	const double inv = B.inv[at];
	int bp;
#define CALL(N) bp = classifier_best<N>(M, B, c, s, inv, tab_s)
	HIBAG_DISPATCH_NWP(M.nwp[c], CALL)
#undef CALL
	best_cell[at] = active ? bp : -1;
}

// k_vote_pick: the cell a matrix-engine classifier of one K step votes for, from the records pass 1 logged
// (HibagBatchView::vrec): the first strict maximum of cell * (1/total) in cell order (src/LibHLA.cpp:2468 -> :1549-1566) is
// the EARLIEST record whose product equals the last record's.  Where 1/total is infinite every positive cell's product is
// infinite and the first one wins; a NaN reciprocal wins nothing.  thread = (sample, classifier).
__global__ __launch_bounds__(64) void k_vote_pick(HibagModelView M, HibagBatchView B, int *__restrict__ best_cell)
{
	const int c = blockIdx.y, s = blockIdx.x * 64 + threadIdx.x;
	if (M.engine[c] == HIBAG_ENGINE_VALU || M.n_step[c] > 1) return;      // k_vote_best_valu / k_vote_scan
	const size_t at = (size_t)c * B.n_pad + s;
	int pick = -1;
	if (B.cw[at] > 0) {
		const uint4 *__restrict__ rec = B.vrec + (size_t)c * 8 * B.n_pad + s;
		const uint4 h = rec[0];
		const double vmax = __hiloint2double((int)h.y, (int)h.x), inv = B.inv[at];
		const int n = (int)h.z;
		if (n > 0 && inv == inv) {
			const uint4 f = rec[(size_t)B.n_pad];
			if (!(fabs(inv) <= 1.79769313486231570815e+308)) pick = (int)f.z;
			else {
				const double pm = vmax * inv;
				int best = 0x7FFFFFFF;
				if (__hiloint2double((int)f.y, (int)f.x) * inv == pm) best = (int)f.z;
				const int nr = min(n - 1, 6);             // ring entries that belong to this batch: slots 2 .. 1 + nr
				for (int j = 0; j < nr; j++) {
					const uint4 r = rec[(size_t)(2 + j) * B.n_pad];
					if (__hiloint2double((int)r.y, (int)r.x) * inv == pm) best = min(best, (int)r.z);
				}
				pick = best;                              // (the last record itself always qualifies)
			}
		}
	}
	best_cell[at] = pick < 0 ? -1 : (int)M.cls_cell[M.cls_off[c] + pick];
}

// k_vote_scan: the same choice for the FP4 classifiers of several K steps, whose cell sums pass 1 stores one and all
// (k_total_wide): the reference's scan itself over the stored sums, thread = sample, sixteen loads in flight.
__global__ __launch_bounds__(64) void k_vote_scan(HibagModelView M, HibagBatchView B, int *__restrict__ best_cell)
{
	const int c = M.wide_cls[blockIdx.y], s = blockIdx.x * 64 + threadIdx.x;
	const size_t at = (size_t)c * B.n_pad + s;
	const bool active = B.cw[at] > 0;
	if (__ballot(active) == 0) { best_cell[at] = -1; return; }            // (pass 1 skipped the classifier: its rows are stale)
	const double *__restrict__ rows = cell_rows(M, B, c, s >> 6) + (s & 63);
	const double inv = B.inv[at];
	const int n = M.cls_n[c];
	double best = 0;
	int bi = -1, i = 0;
	for (; i + 16 <= n; i += 16) {
		double v[16];
#pragma unroll
		for (int j = 0; j < 16; j++) v[j] = rows[(size_t)(i + j) * HIBAG_WAVE];
#pragma unroll
		for (int j = 0; j < 16; j++) { const double prob = v[j] * inv; if (best < prob) { best = prob; bi = i + j; } }
	}
	for (; i < n; i++) { const double prob = rows[(size_t)i * HIBAG_WAVE] * inv; if (best < prob) { best = prob; bi = i; } }
	best_cell[at] = active && bi >= 0 ? (int)M.cls_cell[M.cls_off[c] + bi] : -1;
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

#endif
