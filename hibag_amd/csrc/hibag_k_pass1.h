// hibag_k_pass1.h -- part of hibag_kernels.hip (included there, one translation unit: the walks are templates that inline into
// their kernels): pass 1: k_total, k_total_wide, k_total_scan -- the in-order posterior total per (sample, classifier) -- with the hand-overs between chunked work items and the ensemble scalars.
#ifndef HIBAG_K_PASS1_H_
#define HIBAG_K_PASS1_H_

// ---------------------------------------------------------------------------
// k_total (pass 1): in-order posterior total of one classifier for 64 samples:
// cells visited h1 ascending, h2 >= h1 ascending and added as produced
// (src/LibHLA.cpp:1776-1826).  Empty cells add +0.0 and are skipped.
// grid (ceil(groups/4), C) with the heaviest classifiers first; each of the 4
// wavefronts of a block owns one group of 64 samples.
// The cell sums pass 1 stores for pass 2: per 64-sample group one row of 64 doubles per stored cell of the model
// (classifier after classifier, cell order inside), the groups back to back -- a wavefront writes its classifier's
// cells as one sequential stream, pass 2 reads a tile's cells of a classifier as one contiguous piece, and a row's
// address needs nothing but its number (HibagModelView::cell_row[c] + position) and the group.
__device__ __forceinline__ double *cell_rows(const HibagModelView &M, const HibagBatchView &B, int c, int group)
{
	return B.cells + ((size_t)group * (size_t)M.cell_row[M.n_classifier] + (size_t)M.cell_row[c]) * HIBAG_WAVE;
}

template <int NWP>
__device__ __forceinline__ double classifier_total(const HibagModelView &M, const HibagBatchView &B,
	int c, int s, int i0, int i1, int chunk0, double *__restrict__ rows, const double *tab_s)
{
	LaneMask<NWP> L;
	load_masks<NWP>(B, M.mask_row[c], s, L);
	const uint32_t *__restrict__ cnt = M.cls_cnt + M.cls_off[c];
	const uint32_t *__restrict__ cp = M.stream + M.stream_off[c] + (size_t)chunk0 * HIBAG_CHUNK_DWORDS(NWP);
	double total = 0;
	uint32_t n = cnt[i0];
	for (int i = i0; i < i1; i++) {
		const uint32_t n_next = cnt[i + 1];           // fetched while this cell is evaluated
		const double cell = cell_sum<NWP>(n, cp, L, tab_s);
		if (rows) __builtin_nontemporal_store(cell, &rows[(size_t)i * HIBAG_WAVE + (s & 63)]);   // pass 2 reads the cells back (or k_total_scan, for a split classifier)
		total += cell;
		n = n_next;
	}
	return total;
}

// ---- hand-overs -----------------------------------------------------------------------------------
// A pass is a few thousand work items of similar length on ~1,000 resident workgroups, so its last round runs
// mostly empty (10k samples: 3,200 items of pass 2 on 1,024 slots = 3.1 rounds, the chip idle for most of the
// fourth).  The items of the last round or two are therefore cut into K chunks along their classifier sequence,
// each chunk a workgroup of its own: the last round is then made of pieces a K-th as long.  A chunk continues the
// sums of the one before it -- parked in the output rows and announced by a flag -- so the additions and their
// order are those of the undivided item.  All first chunks are dispatched before all second chunks, and so on: a
// workgroup only ever waits for one that was dispatched (a whole round of chunks) earlier, and no cycle can form.
//
// Visibility.  Chunks of one item run on one XCD (workgroups b and b + 8 share an XCD, and an item's chunks sit a
// multiple of 8 apart), so the hand-over goes through that XCD's L2: the parked sums are sc1 stores (store_parked), complete
// once the storing wavefront's vmcnt is 0; the flag follows behind a workgroup barrier
// as an sc1 store too; the reader polls it with sc1 loads and fetches the sums with sc1 loads, which
// bypass its CU's L1.  No cache is flushed or invalidated (agent-scope fences cost 2-7 us each here and
// evict everybody's L1).  The dispatch order is observed behaviour, not a contract: every flag carries the XCD
// number of its writer, and a reader on another XCD reports the launch as failed instead of using the sums.
__device__ __forceinline__ unsigned xcc_id()
{
	unsigned x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	return x & 15u;
}

__device__ __forceinline__ void handover_post(unsigned long long *flag, uint32_t epoch, uint32_t progress, bool drop = false)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wavefront's parked sums have reached L2
	__syncthreads();
	if (threadIdx.x == 0 && !drop)                              // (drop: fault injection, HibagBatchView::drop_post)
		__hip_atomic_store(flag, ((unsigned long long)epoch << 32) | (xcc_id() << 24) | progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait for the chunk before this one.  A flag that never comes (B.spin_limit polls: scaled with the model's longest
// work item, hibag_predict.hip make_batch) or that was written on another XCD is an error the caller must see: the host-mapped
// word for the host (sticky model status), the device word for k_scalars, which poisons the batch's outputs.
__device__ __forceinline__ void handover_wait(unsigned long long *flag, const HibagBatchView &B, uint32_t progress)
{
	if (threadIdx.x == 0) {
		const unsigned long long want = ((unsigned long long)B.epoch << 32) | progress;
		unsigned spins = 0;
		int bad = 0;
		for (;;) {
			const unsigned long long v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((v & ~(15ull << 24)) == want) {
				if (((unsigned)(v >> 24) & 15u) != xcc_id()) bad = 2;     // written on another XCD: not coherent through L2
				break;
			}
			__builtin_amdgcn_s_sleep(16);
			if (++spins > B.spin_limit) { bad = 1; break; }     // give up rather than hang the device
		}
		if (bad) {
			*B.err = bad;
			__hip_atomic_store(B.err_dev, B.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	__syncthreads();
}

// A (sample, classifier) whose total is 0 or so small that 1/total is not finite: the reference's `cell * (1/total)` then
// turns the classifier's structurally EMPTY cells into 0 * inf = NaN too (src/LibHLA.cpp:1826-1828).  Pass 2 only visits
// cells that have haplotype pairs, so pass 1 lists these rare pairs here and k_nan_cells adds the NaN terms afterwards.
// List: HibagBatchView::err_dev -- [2] = count (reset by the host before pass 1), entries of 8 bytes from byte 16 on.
#define HIBAG_NAN_CAP 2040
__device__ __forceinline__ void note_infinite_reciprocal(const HibagBatchView &B, int c, int s, double w, double inv)
{
	if (w > 0 && !(fabs(inv) <= 1.79769313486231570815e+308)) {
		const uint32_t i = atomicAdd(B.err_dev + 2, 1u);
		if (i < HIBAG_NAN_CAP) reinterpret_cast<unsigned long long *>(B.err_dev + 4)[i] = ((unsigned long long)(uint32_t)c << 32) | (uint32_t)s;
	}
}

// a parked sum: read past the CU's L1 (global_load_dwordx2 ... sc1), and WRITTEN sc1 too -- producer and consumer then match
// the first row of MI355X_MICROARCH.md's hand-off table ("stores, all sc1 / loads, all sc1", one lane's sc1 flag store behind
// every storing wave's vmcnt(0) and the workgroup barrier, an sc1 poll).  A handful of 8-byte stores per chunk: no measurable cost.
__device__ __forceinline__ double load_parked(const double *p)
{
	return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void store_parked(double *p, double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// first classifier of a tile whose cost prefix reaches `target` (acc_cum row of the tile: C + 1 entries)
__device__ __forceinline__ int chunk_bound(const uint32_t *__restrict__ cum, int C, uint64_t target)
{
	int a = 0, b = C;
	while (a < b) {
		const int mid = (a + b) >> 1;
		if (cum[mid] >= target) b = mid; else a = mid + 1;
	}
	return a;
}

// Workgroup b < n_whole: item b (items run classifier-major: item = b / gx, group quad = b % gx).  Behind them the
// other `rest` items in K chunks along their block lists, all first chunks, then all second chunks ... ("hand-overs"
// above; `stride` = rest rounded up to a multiple of 8, so that the chunks of an item share an XCD).  Only
// matrix-engine items are cut; a chunk hands over the lane's running total and the sum of the cell it stopped in,
// parked in the classifier's tot / inv rows.
// STORE: every cell sum goes to HibagBatchView::cells for pass 2 to read back (models whose pass 2 streams, see
// k_accum_cells); otherwise only a split VALU-engine classifier stores its cells (for k_total_scan).
// FP4ONLY: every work item is a one-step FP4 classifier (HibagModelView::all_fp4) -- the build for six workgroups per CU
// carries that loop alone: at 80 registers the int8 and VALU-engine loops would spill, the FP4 loop does not.
// VOTE (majority vote; never together with STORE, whose sums only pass 2 reads, and never with chunked items): the walk
// logs the records of its cell sums for k_vote_pick (HibagBatchView::vrec).
// WALK (the builds a launch chooses from): 0 = every engine, the one-step FP4 classifiers on prebuilt rows where the model has
// them; 1 / 2 = FP4ONLY builds, whose work items are all one-step FP4 classifiers -- 1 generates their rows from the haplotype
// table, 2 reads the prebuilt ones (HibagModelView::p1_prebuilt) and nothing else: 68 vector registers, 2 % faster than
// a build that carries both walks.
template <bool STORE, int OCC, int WALK, bool VOTE = false>
__global__ __launch_bounds__(BLOCK_THREADS, OCC) void k_total(HibagModelView M, HibagBatchView B, int gx, int n_whole, int rest, int stride, int K)
{
	static_assert(!(STORE && VOTE), "the majority vote has no second pass to store cell sums for");
	constexpr bool FP4ONLY = WALK != 0;
	__shared__ double tab_s[HIBAG_TAB_N];
	int li = blockIdx.x, k = 0;
	if (li >= n_whole) {
		const int jj = li - n_whole;
		k = jj / stride;
		if (jj - k * stride >= rest) return;
		li = n_whole + jj - k * stride;
	}
	const int *__restrict__ item = M.item + 4 * (li / gx);
	const int c = item[0];
	const int nkb = FP4ONLY ? HIBAG_ENGINE_FP4 : M.engine[c];      // matrix-engine variant, 0 = VALU engine
	// blocks [b0, b1) of the classifier's list
	int b0 = 0, b1 = nkb > 0 ? M.cls_nblk[c] : 0;
	const bool chunked = blockIdx.x >= n_whole && nkb > 0;
	if (blockIdx.x >= n_whole) {
		if (nkb > 0) {
			// (readfirstlane: the division runs on the vector ALU, and a list offset that lives in a vector register
			// turns every list load of the walk into a waterfall loop)
			const long long nb = b1;
			b0 = __builtin_amdgcn_readfirstlane((int)(nb * k / K)); b1 = __builtin_amdgcn_readfirstlane((int)(nb * (k + 1) / K));
			if (b0 >= b1 && !(k == K - 1 && nb == 0)) return;      // (fewer blocks than chunks: an empty list still needs its total written)
		} else if (k > 0) return;                                  // VALU-engine items are not cut
	}
	const bool first = b0 == 0, last = !chunked || k == K - 1;
	stage_table(M, tab_s);
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int group = (li % gx) * BLOCK_WAVES + wave;
	const int s = group * HIBAG_WAVE + (threadIdx.x & 63);
	const size_t at = (size_t)c * B.n_pad + s;
	// a wavefront beyond the batch, or one none of whose samples uses the classifier (src/LibHLA.cpp:2451), has nothing
	// to do in any chunk
	bool live = group * HIBAG_WAVE < B.n_pad;
	if (live) live = __ballot(B.cw[at] > 0) != 0;
	unsigned long long *flag = B.sync_total + (li - n_whole);
	if (!first) handover_wait(flag, B, (uint32_t)b0);
	if (live) {
		double total = 0;
		const bool split = M.n_split > 0 && M.split_row[c] >= 0;    // a split classifier: k_total_scan adds its cells in order
		// this classifier's stored cell sums, one row each (a VALU-engine classifier stores all or none)
		double *__restrict__ rows = STORE || split ? cell_rows(M, B, c, group) : nullptr;
		if (nkb > 0) {
			double cell = 0;
			if (!first) { total = load_parked(&B.tot[at]); cell = load_parked(&B.inv[at]); }
			// cells closed by earlier chunks = the row this one starts at
			int row = !STORE || first ? 0 : (int)M.blk_close[(M.blk_off[c] - M.p1_base) / HIBAG_PLIST_DWORDS + (uint64_t)b0];
			const int lane = threadIdx.x & 63;
			// (a store issued where the cell closes; parking the sums in LDS and sending them a block later, so that
			// the vmcnt waits of the look-ahead gathers never include a young store, measured 10 % slower)
			// (stores through a raw buffer with the row as a scalar offset -- no 64-bit address on the vector ALU -- measured 2 % slower)
			// majority vote: the lane's records so far -- the largest cell sum, how many records, the log slot of the next one --
			// and (wave-uniform) the position of the closing cell in the classifier's list of non-empty cells
			double vmax = 0;
			int vcnt = 0, vslot = 1, ci = 0;
			uint4 *const vlog = VOTE ? B.vrec + (size_t)c * 8 * B.n_pad + s : nullptr;
			auto fin = [&](double v, bool stored, int) {
#ifdef HIBAG_STORE_PLAIN      // (variant: write-back stores instead of streaming ones)
				if (STORE && stored) { rows[(size_t)row * HIBAG_WAVE + lane] = v; row++; }
#elif defined(HIBAG_ABL_STX4)      // (timing ablation: as many stores, 16 bytes per lane each -- overlapping: the same memory lines)
				if (STORE && stored) { __builtin_nontemporal_store(f64x2{v, v}, (f64x2 *)&rows[(size_t)row * HIBAG_WAVE + lane]); row++; }
#elif defined(HIBAG_ABL_STHALF)    // (timing ablation: every second stored cell is written)
				if (STORE && stored) { if (!(row & 1)) __builtin_nontemporal_store(v, &rows[(size_t)row * HIBAG_WAVE + lane]); row++; }
#else
				if (STORE && stored) { __builtin_nontemporal_store(v, &rows[(size_t)row * HIBAG_WAVE + lane]); row++; }
#endif
				total += v;
				asm("" : "+v"(total));                    // keeps the cell end a scalar branch
				if (VOTE) {
					if (v > vmax) {                       // a record (NaN is none, like `best < prob` in the reference)
						vmax = v;
						vlog[(size_t)vslot * B.n_pad] = uint4{(uint32_t)__double2loint(v), (uint32_t)__double2hiint(v), (uint32_t)ci, 0u};
						vslot = vslot == 1 || vslot == 7 ? 2 : vslot + 1;
						vcnt++;
					}
					ci++;
				}
			};
#define CALLX(E, PRE) { LaneOperand T;                                                                                         \
			constexpr bool OWN = PRE && E == HIBAG_ENGINE_FP4 && TOTAL_OWN && !VOTE;   /* the walk without lane swaps (walk_blocks); the vote build has no registers for it */ \
			if (OWN) load_operand_own_sample(B, M.bt_row[c], group, lane, T); else load_operand_row<E>(B, M.bt_row[c], c, group, lane, T); \
			ListCursor cur;                                                                                                \
			walk_blocks<E, TOTAL_G, PRE, OWN>(M, M.blk_off[c] + (uint64_t)b0 * HIBAG_PLIST_DWORDS, b1 - b0, lane, cur,       \
				hap_rsrc(M, M.hap_off[c]), M.n_snp_c[c], T, WideSrc(), tab_s, cell, fin); }
#define CALL(E) CALLX(E, false)
			// one-step FP4 classifiers of a model small enough for prebuilt A-operand rows walk those (HibagModelView::parow)
			if (WALK == 2) CALLX(HIBAG_ENGINE_FP4, true)
			else if (WALK == 1) CALLX(HIBAG_ENGINE_FP4, false)
			else if (nkb == HIBAG_ENGINE_FP4) { if (M.p1_prebuilt) CALLX(HIBAG_ENGINE_FP4, true) else CALLX(HIBAG_ENGINE_FP4, false) }
			else if (nkb == HIBAG_ENGINE_I8) CALL(HIBAG_ENGINE_I8)
			else CALL(HIBAG_ENGINE_I8S)
#undef CALL
#undef CALLX
			if (!last) { store_parked(&B.tot[at], total); store_parked(&B.inv[at], cell); }
			if (VOTE) vlog[0] = uint4{(uint32_t)__double2loint(vmax), (uint32_t)__double2hiint(vmax), (uint32_t)vcnt, 0u};
		} else if (!FP4ONLY) {
#define CALL(N) total = classifier_total<N>(M, B, c, s, item[1], item[2], item[3], rows, tab_s)
			HIBAG_DISPATCH_NWP(M.nwp[c], CALL)
#undef CALL
			if (split) return;
		}
		if (last) {
			const double inv = 1 / total;                 // src/LibHLA.cpp:1827 (inf when total == 0)
			B.tot[at] = total;
			B.inv[at] = inv;
			// (beside the weight k_pack left there: pass 2 reads both in one load -- and reads 0 instead of 1/total where the sample
			// does not use the classifier: its term there is (cell * 0) * 0 = +0, which pass 2 would otherwise select per block)
			B.winv[2 * at + 1] = B.cw[at] > 0 ? inv : 0.0;
			note_infinite_reciprocal(B, c, s, B.cw[at], inv);
		}
	}
	if (!last) handover_post(flag, B.epoch, (uint32_t)b1, B.drop_post == 1 && li == n_whole && k == 0);
}

// k_total_wide: pass 1 of the FP4 classifiers with several K steps (33 .. 112 SNPs) -- a kernel of their own, started
// beside k_total on a second stream: their walk needs a dozen registers more than k_total's 96.  Their lists come in
// segments of whole cells (HibagModelView::wide_seg), one workgroup per segment and group quad; every cell sum is
// stored (pass 2 reads them back whatever the model's other classifiers do) and k_total_scan adds them in order.
// grid (group quads, segments).
// WHOLE: every segment is a whole classifier (a model with many of them): the walk forms the in-order total itself.
// (three workgroups per CU: the walk keeps the further steps' B operands and a block's images of every step in registers,
// 24 + 32 of them -- at four per CU, 128 registers, it spilled)
template <bool WHOLE>
#ifndef HIBAG_WIDE_OCC
#define HIBAG_WIDE_OCC 3
#endif
__global__ __launch_bounds__(BLOCK_THREADS, HIBAG_WIDE_OCC) void k_total_wide(HibagModelView M, HibagBatchView B)
{
	__shared__ double tab_s[HIBAG_WIDE_TAB ? HIBAG_WIDE_TAB_N : HIBAG_TAB_N];     // (HIBAG_WIDE_TAB: the bank-interleaved variant, hibag_k_engine.h)
#if defined(HIBAG_WIDE_OLDWALK) || !HIBAG_WIDE_TAB
	stage_table(M, tab_s);
#else
	stage_table_wide(M, tab_s);
#endif
	const int *__restrict__ seg = M.wide_seg + 4 * blockIdx.y;
	const int c = seg[0];
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int group = blockIdx.x * BLOCK_WAVES + wave;
	if (group * HIBAG_WAVE >= B.n_pad) return;
	const int s = group * HIBAG_WAVE + lane;
	if (__ballot(B.cw[(size_t)c * B.n_pad + s] > 0) == 0) return;          // nobody needs this classifier (src/LibHLA.cpp:2451)
	double cell = 0, total = 0;
	double *__restrict__ rows = cell_rows(M, B, c, group);
	int row = seg[1];
	auto fin = [&](double v, bool, int) {
		if (!ABL_WIDE_NOSTORE) { __builtin_nontemporal_store(v, &rows[(size_t)row * HIBAG_WAVE + lane]); row++; }
		if (WHOLE) { total += v; asm("" : "+v"(total)); }   // (the asm keeps the cell end a scalar branch)
	};
	const WideSrc wide = wide_src(B, M.bt_row[c], M.n_step[c], group);
	LaneOperand T;
	load_operand_row<HIBAG_ENGINE_FP4W>(B, M.bt_row[c], c, group, lane, T);
	ListCursor cur;
	walk_blocks<HIBAG_ENGINE_FP4W, TOTAL_G, false>(M, M.wide_seg_off[blockIdx.y], seg[2], lane, cur, hap_rsrc(M, M.hap_off[c]),
		M.n_snp_c[c] - HIBAG_FP4_STEP_SNPS * (wide.nstep - 1), T, wide, tab_s, cell, fin);
	if (WHOLE) {
		const size_t at = (size_t)c * B.n_pad + s;
		const double inv = 1 / total;                 // src/LibHLA.cpp:1827 (inf when total == 0)
		B.tot[at] = total;
		B.inv[at] = inv;
		B.winv[2 * at + 1] = B.cw[at] > 0 ? inv : 0.0;
		note_infinite_reciprocal(B, c, s, B.cw[at], inv);
	}
}

// k_total_scan: the in-order total of a split classifier from its stored cell sums; thread = sample.
// Thirty-two loads in flight, then the thirty-two additions in cell order (with one dependent load per addition the
// kernel would be pure memory latency: a few hundred cells, one wavefront per 64 samples).
__global__ void k_total_scan(HibagModelView M, HibagBatchView B)
{
	const int c = M.split_cls[blockIdx.y];
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= B.n_pad) return;
	const double *__restrict__ rows = cell_rows(M, B, c, s >> 6) + (s & 63);
	const int n = M.cls_n[c];
	double total = 0;
	int i = 0;
	for (; i + 32 <= n; i += 32) {
		double v[32];
#pragma unroll
		for (int j = 0; j < 32; j++) v[j] = rows[(size_t)(i + j) * HIBAG_WAVE];
#pragma unroll
		for (int j = 0; j < 32; j++) total += v[j];
	}
	for (; i < n; i++) total += rows[(size_t)i * HIBAG_WAVE];
	B.tot[(size_t)c * B.n_pad + s] = total;
	B.inv[(size_t)c * B.n_pad + s] = 1 / total;
	B.winv[2 * ((size_t)c * B.n_pad + s) + 1] = B.cw[(size_t)c * B.n_pad + s] > 0 ? 1 / total : 0.0;
	note_infinite_reciprocal(B, c, s, B.cw[(size_t)c * B.n_pad + s], 1 / total);
}

// The per-sample ensemble scalars, classifiers in order (k_scalars, or the tile-0 workgroups of k_accum):
//   part[P]   = sum of weights   (_Sum_Weight, src/LibHLA.cpp:1505; for the
//               majority vote the number of classifiers that produced a call)
//   part[P+1] = sum_matching = sum_c total_c * w_c        (:2458)
//   part[P+2] = num_matching = sum_c w_c                  (:2459)
template <int NB = 16>
__device__ __forceinline__ void ensemble_scalars(const HibagModelView &M, const HibagBatchView &B, int s, const int *__restrict__ best_cell)
{
	double sum_w = 0, sum_m = 0, num_m = 0;
	for (int c0 = 0; c0 < M.n_classifier; c0 += NB) {
		// sixteen classifiers' loads in flight, then the sums in classifier order (one thread per sample:
		// with dependent loads this would be pure memory latency)
		double wv[NB], tv[NB];
		int bv[NB];
#pragma unroll
		for (int j = 0; j < NB; j++) {
			const bool in = c0 + j < M.n_classifier;
			const size_t at = (size_t)(in ? c0 + j : c0) * B.n_pad + s;
			wv[j] = in ? B.cw[at] : 0.0;
			tv[j] = B.tot[at];
			bv[j] = best_cell ? best_cell[at] : 0;
		}
#pragma unroll
		for (int j = 0; j < NB; j++) {
			const double w = wv[j];
			if (!(w > 0)) continue;
			sum_m += tv[j] * w;
			num_m += w;
			if (best_cell) { if (bv[j] >= 0) sum_w += 1.0; }
			else sum_w += w;
		}
	}
	const size_t P = (size_t)M.n_cell;
	B.part[(P + 0) * B.n_pad + s] = sum_w;
	B.part[(P + 1) * B.n_pad + s] = sum_m;
	B.part[(P + 2) * B.n_pad + s] = num_m;
}

// A hand-over of this batch failed (see handover_wait): its sums are not to be trusted.  The weight sum is never NaN
// otherwise, so NaN there is the in-band mark every k_finish_* kernel (and a merge of partial sums) recognises.
__device__ __forceinline__ void poison_scalars_if_failed(const HibagModelView &M, const HibagBatchView &B, int s)
{
	if (__hip_atomic_load(B.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != B.epoch) return;
	const size_t P = (size_t)M.n_cell;
	for (int q = 0; q < 3; q++) B.part[(P + q) * B.n_pad + s] = __builtin_nan("");
}

#endif
