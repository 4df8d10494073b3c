// hibag_k_pass2.h -- part of hibag_kernels.hip (included there, one translation unit: the walks are templates that inline into
// their kernels): pass 2: k_accum (the block stream: small cells evaluated again, stored sums read back) and k_accum_cells (every cell read back).
#ifndef HIBAG_K_PASS2_H_
#define HIBAG_K_PASS2_H_

// ---------------------------------------------------------------------------
// k_accum (pass 2): for one tile of allele-pair cells and 64 samples, go through the classifiers in order and do
//     S[p] += (cell * (1/total)) * w          (src/LibHLA.cpp:1828 then :1497-1507)
// with the tile's S in LDS (one row of 64 doubles per cell, conflict-free).  What a classifier contributes to a tile is
// either evaluated again from its haplotype pairs (the cells with few pairs) or read back from the sums pass 1 stored
// (HibagModelView::store_cells); both arrive here as ONE STREAM OF BLOCKS per tile (hibag_device.h, "E-stream"): the
// blocks of classifier 0, 1, 2 ... that have anything for the tile, each block 32 pair slots -- their prebuilt A-operand rows --
// plus a 32-byte header that holds everything scalar about it besides the slots' factors: its end-of-cell mask, the tile rows
// its cells close into, up to seven stored sums to add, how many of its slots are worth evaluating, and what names the NEXT
// block's classifier (-> weight and 1/total rows), operand row and stored sums (hibag_device.h).  Round 2 walked (classifier, tile) "visits" -- mostly one short
// block each -- with a scalar prologue per visit (record, descriptors, engine dispatch) and nothing of the next visit in
// flight while the current one ran: 0.73 us of SIMD time per visit against 0.35 us of instructions.  As a stream the loop
// body is one block, and behind its matrix instructions EVERYTHING of block b + 1 is requested -- A rows, B operand, weight,
// 1/total, stored sums (its header and first factors at the top already) -- so a whole block's evaluation covers each
// latency, across classifier boundaries too.  Only one-step FP4 classifiers are evaluated here; every other engine has all
// its cells stored by pass 1 (their blocks carry stored sums only).
//
// A cell that is structurally empty in a classifier contributes (0 * inv) * w = +0 and is absent from the stream; where
// 1/total is not finite the reference's 0 * inf = NaN is added by k_scalars afterwards (NaN absorbs: the order of that
// addition cannot show).
//
// grid = 8 x (n_whole + K * (items per XCD - n_whole)): per XCD first the undivided items, then the others' first
// chunks, second chunks, ... ("hand-overs" above).
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

#ifndef ACCUM_OCC
#define ACCUM_OCC 5                         // workgroups per CU pass 2 is compiled for (LDS: HIBAG_TILE in hibag_device.h; registers: HIBAG_STORED_PER_VISIT)
#endif

// What a block needs that is requested a block ahead and is still in use while the NEXT block's is in flight: its header
// and its first factors (scalar registers) and the lane's weight and 1/total.  The loop body exists twice
// (A -> B, B -> A): the two sets take turns, nothing is moved from a "next" register to a "current" one.
struct AccumAhead {
	u32x8 hv;           // the E-stream header (hibag_device.h): end-of-cell mask, stored sums, the next block's request words, tile rows, groups worth evaluating
	FactorGroup<ACCUM_G>::type F;   // the first ACCUM_G factors
	f64x2 winv;         // {weight, 1/total} of the block's classifier for this lane's sample
};

__global__ __launch_bounds__(ACCUM_WAVES * HIBAG_WAVE, ACCUM_OCC) void k_accum(HibagModelView M, HibagBatchView B, int n_whole, int K)
{
	// (pass 2 evaluates one-step FP4 classifiers only: distances up to 2 * 30, the first 64 table entries)
	__shared__ double tab_s[ACCUM_TAB_N];
	__shared__ double acc_s[ACCUM_WAVES][HIBAG_TILE][HIBAG_WAVE];
#ifdef HIBAG_ACCUM_STAMPS
	__shared__ unsigned long long stamp_s[ACCUM_STAMP_N];
	if (threadIdx.x < ACCUM_STAMP_N) stamp_s[threadIdx.x] = 0;
	unsigned long long stamp_t = 0;
#endif

	// Work item = (XCD, four sample groups, one tile); the four wavefronts of a workgroup take the four groups.
	// They read the same blocks at about the same time, so those
	// come from the CU's L1 for three of them.  Workgroups are dealt round-robin over the 8 XCDs, so sample
	// group g goes to XCD g % 8 with all its tiles: its operands / weights / totals are fetched into one XCD's L2 only.
	const int n_group = B.n_pad / HIBAG_WAVE;
	const int n_gq = ((n_group + 7) / 8 + ACCUM_WAVES - 1) / ACCUM_WAVES;      // group quads per XCD
	const int C = M.n_classifier;
	const int n_item_x = n_gq * M.n_tile;             // items of one XCD: (group quad, tile), tile fastest
	const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	double (*acc)[HIBAG_WAVE] = acc_s[wave];

	// this workgroup's classifiers [cb, ce) of its item: cut where the tile's block count is split evenly
	int item = wx, cb = 0, ce = C;
	if (wx >= n_whole) {
		const int rest = n_item_x - n_whole, k = (wx - n_whole) / rest;
		item = n_whole + (wx - n_whole) - k * rest;
		const uint32_t *__restrict__ cum = M.etile_cstart + (size_t)(item % M.n_tile) * (C + 1);
		const uint64_t total = cum[C];
		if (k > 0) cb = chunk_bound(cum, C, (total * (uint64_t)k + K - 1) / K);
		if (k < K - 1) ce = chunk_bound(cum, C, (total * (uint64_t)(k + 1) + K - 1) / K);
		if (cb >= ce) return;                         // (fewer classifiers than chunks)
	}
	stage_table(M, tab_s, ACCUM_TAB_N);
	const int jq = item / M.n_tile, tile = item - jq * M.n_tile;
	const int group = (jq * ACCUM_WAVES + wave) * 8 + xcd;
	unsigned long long *flag = B.sync + (size_t)xcd * n_item_x + item;
	if (cb > 0) handover_wait(flag, B, (uint32_t)cb);
#ifdef HIBAG_ACCUM_STAMPS
	int bb_diag = 0, be_diag = 0;
#endif
	if (group < n_group) {
	const int s = group * HIBAG_WAVE + lane;
	const int ncell = M.tile_n[tile];
	const int p0 = M.tile_p0[tile];

	if (cb > 0) {                                     // continue the parked sums (all loads in flight together)
		double v[HIBAG_TILE];
#pragma unroll
		for (int j = 0; j < HIBAG_TILE; j++) v[j] = load_parked(&B.part[(size_t)(p0 + (j < ncell ? j : 0)) * B.n_pad + s]);
#pragma unroll
		for (int j = 0; j < HIBAG_TILE; j++) acc[j][lane] = v[j];
	} else {
#pragma unroll
		for (int j = 0; j < HIBAG_TILE; j++) acc[j][lane] = 0;
	}

	const ConstPtr<uint32_t> cst = as_const(M.etile_cstart) + (size_t)tile * (C + 1);
	const int bb = __builtin_amdgcn_readfirstlane((int)cst[cb]), be = __builtin_amdgcn_readfirstlane((int)cst[ce]);
#ifdef HIBAG_ACCUM_STAMPS
	if (wave == 0) { bb_diag = bb; be_diag = be; }
#endif
	if (bb < be && !(ABL2_NOLOOP && B.n_pad >= 0)) {
		// the tile's blocks [bb, be): their prebuilt A-operand rows as a raw buffer rebased at block bb (no 4 GB limit on the stream)
		const uint64_t blk0 = as_const(M.etile_blk0)[tile] + (uint64_t)bb;
		auto bytes32 = [](size_t n) { return n > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)(uint32_t)n; };
		// (every descriptor gets a flags word of its own: sharing one constant register between the four made the compiler keep
		// two copies of three of them -- twelve scalar registers, and three moves per block)
		auto rsrc_flags = [] { int f = 0x00020000; asm volatile("" : "+s"(f)); return f; };
		const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(M.parow + blk0 * 64), 0,
			bytes32((size_t)(M.parow_blocks - blk0) * 1024u), rsrc_flags());
		ConstPtr<double> fac = as_const(M.pfac) + blk0 * HIBAG_PLIST_DWORDS;                 // the slots' frequency factors
		ConstPtr<u32x8> eh = (ConstPtr<u32x8>)(as_const(M.ehdr) + blk0 * 8);                 // the blocks' 8-dword headers (scalar loads)
		typedef FactorGroup<ACCUM_G>::type AFG;
		// the batch's operand / {weight, 1/total} rows and this group's stored sums as raw buffers too: a row is then a scalar
		// offset (classifier or row number times the row size, SALU) added to one constant per-lane offset -- no 64-bit address
		// arithmetic on the vector ALU.  (hibag_predict.hip batch_limit keeps every one of these arrays below 4 GB.)
		// (every descriptor ends where its array ends: a request past it -- a look-ahead through a header that names more than
		// exists -- reads zeros instead of faulting)
		// (both rebased at this wavefront's 64 samples: the lane's offset into a row is then the same 16 * lane as into a block's rows)
		const size_t g_bytes = (size_t)group * HIBAG_WAVE * 16u;
		const __amdgpu_buffer_rsrc_t r_bt = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)B.bt + g_bytes), 0,
			bytes32((size_t)B.bt_rows * B.n_pad * 16u - g_bytes), rsrc_flags());
		const __amdgpu_buffer_rsrc_t r_wi = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)B.winv + g_bytes), 0,
			bytes32((size_t)C * B.n_pad * 16u - g_bytes), rsrc_flags());
		const __amdgpu_buffer_rsrc_t r_sv = __builtin_amdgcn_make_buffer_rsrc(
			(void *)(B.cells + (size_t)group * (size_t)as_const(M.cell_row)[C] * HIBAG_WAVE), 0,
			bytes32((size_t)as_const(M.cell_row)[C] * HIBAG_WAVE * 8u), rsrc_flags());
		const int vo_a = lane * 16, vo_row = vo_a, vo_sv = lane * 8;
		const uint32_t row_stride = (uint32_t)B.n_pad * 16u;          // bytes per operand row and per classifier's {weight, 1/total} row
		constexpr int NS = HIBAG_STORED_PER_VISIT;

		// per-lane data of the block in hand, requested a block ahead into the registers the block before has just finished with
		v4i arow, t0, t1;                             // the A-operand row, the B operand (two sample halves)
		double sv[NS];                                // the stored sums
		double cell = 0;
		// the stored sums of a block (word 1 of its header: first row | count << 25).  Their number differs from block to block,
		// so a wait that leaves them in flight would have to be a counted one the compiler cannot get right; ALWAYS requesting
		// HIBAG_STORED_PER_VISIT of them (the ones a block lacks out of the buffer's range: no memory access) so that every wait
		// is exact, and adding them at the end of the block, was measured: pass 2 +15 % -- the loads that fetch nothing still cost
		// their issue (profiles/r05_pass2_notes.txt).
		auto request_sv = [&](uint32_t w1) {
			const int ns = abl2_stored(w1);
			if (ns > 0) {
				const int sr = (int)(abl2_stored_row(w1) * (uint32_t)(HIBAG_WAVE * 8));
				int vo = vo_sv;
				asm volatile("" : "+v"(vo));                  // (kept out of the loop-invariant code: vo + 512 i in six registers instead of the instructions' offset fields)
#pragma unroll
				for (int i = 0; i < NS; i++) {
					if (i >= ns) break;
#ifdef HIBAG_ABL2_SVNOLOAD                    // (timing ablation: the stored sums are added, never loaded)
					asm volatile("" : "=v"(sv[i]));
#elif defined(HIBAG_ABL2_SVX4)                // (timing ablation: as many loads, 16 bytes per lane each -- overlapping: the same memory lines)
					sv[i] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r_sv, vo + i * HIBAG_WAVE * 8, sr, 2))[0];
#elif defined(HIBAG_ABL2_SVHALF)              // (timing ablation: every second stored sum is loaded)
					if (i & 1) asm volatile("" : "=v"(sv[i]));
					else sv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_sv, vo + i * HIBAG_WAVE * 8, sr, 2));
#else
#ifdef HIBAG_SV_PLAIN                         // (correct variant: default cache policy instead of nt)
					sv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_sv, vo + i * HIBAG_WAVE * 8, sr, 0));
#else
					sv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_sv, vo + i * HIBAG_WAVE * 8, sr, 2));   // (read once: nt; the row's distance as the instruction's immediate offset)
#endif
#endif
				}
			}
		};
		// the request words of a header (its own -- words 7, 1 -- or the next block's, words 2, 3): classifier | operand row << 16, stored row | stored sums << 25
		auto request_lane = [&](uint32_t w0, uint32_t w1, int soff_a, f64x2 &winv) {
			arow = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(pr, vo_a, soff_a, 0));
			const int sb = (int)((w0 >> 16) * row_stride);
			t0 = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r_bt, vo_row, sb, 0));
			t1 = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r_bt, vo_row, sb + (int)row_stride, 0));
			if (ABL2_NOWINV) winv = f64x2{1.0 + (double)w0, 2.0};
			else winv = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r_wi, vo_row, (int)((w0 & 0xFFFFu) * row_stride), 0));
			request_sv(w1);
		};
		// One block: `cur` = what it needs (arrived: requested a block ago), `nxt` = where the next block's goes.
		// Order: the stored sums are added, the matrix instructions issued -- which frees this block's rows, operand and stored
		// sums' registers -- then EVERYTHING of block b + 1 is requested, and only then the long part, the pairs' accumulation,
		// runs: it covers every latency.  No load of the loop is waited for with a count: at the top of a block everything in
		// flight is that block's.
		auto one_block = [&](const int rel, AccumAhead &cur, AccumAhead &nxt) {
#if defined(HIBAG_ACCUM_PRIO) && HIBAG_ACCUM_PRIO == 1      // (measured variant: the head of a block -- waits, stored sums, matrix instructions, requests -- at raised priority)
			__builtin_amdgcn_s_setprio(1);
#elif defined(HIBAG_ACCUM_PRIO) && HIBAG_ACCUM_PRIO == 2    // (measured variant: the pairs' accumulation at raised priority)
			__builtin_amdgcn_s_setprio(0);
#endif
			const double w_c = cur.winv[0];
			const bool active = w_c > 0;
			// (as integers in scalar registers -- a count of lanes is one scalar instruction; a bool that lives across the requests
			// below ends up in a vector register and back)
			const int any = __builtin_popcountll(__ballot(active));   // 0: nobody in the group uses the classifier (src/LibHLA.cpp:2451): nothing to add
			// inactive lanes (weight 0) must keep their sums: with 1/total replaced by 0 their term is
			// (cell * 0) * 0 = +0 and a + 0 == a, which spares a select per closed cell -- and pass 1 has written that 0 (winv)
			const double inv_e = cur.winv[1];
			const uint32_t endmask = cur.hv[0];
			// the groups of four records worth evaluating, 0..8, are the top bits of word 6: "group g has any" is one compare of the
			// word with a constant
			const uint32_t gword = cur.hv[6];
			auto live4 = [&](int g) { return g == 0 || gword >= ((uint32_t)(g + 1) << 28); };      // (asked only where the block has any: `eval`)
			// (this block's scalar data has arrived: the block before waited for it at its end, see the last statement below)
			__builtin_amdgcn_sched_barrier(0);
			nxt.hv = eh[rel + 1];
			nxt.F = *(ConstPtr<AFG>)(fac + (size_t)(rel + 1) * HIBAG_PLIST_DWORDS);
			// The other three 64-byte lines of block b + 1's factors are touched a block ahead, so that the scalar loads of its
			// later groups hit the scalar cache (-5 % on the kernel): one dword each, volatile so that the loads stay HERE, "used"
			// at the end of this block (the compiler waits for them there, where they are long done).
			typedef const volatile __attribute__((address_space(4))) uint32_t *TouchPtr;
			const TouchPtr touch = (TouchPtr)(uintptr_t)(fac + (size_t)(rel + 1) * HIBAG_PLIST_DWORDS);
			const uint32_t tch0 = touch[16], tch1 = touch[32], tch2 = touch[48];
			ACCUM_STAMP(0);
			// (`eval` and `eval_b`: the same number in two scalar registers the compiler cannot tell are equal -- one condition used in
			// two places became a lane mask parked in a vector register between them)
			int eval = ABL2_NOEVAL ? 0 : (any != 0 ? (int)(gword >> 28) : 0), eval_b = eval;
			asm volatile("" : "+s"(eval), "+s"(eval_b));
			// ---- the sums pass 1 stored for this block's classifier:   S[p] += (cell * (1/total)) * w
			{
				const int ns = abl2_stored(cur.hv[1]);
				// (a classifier nobody in the group uses is passed over: its blocks leave `cell` as it was, and the next block's
				// header bit describes a stream in which they were evaluated -- so the sum goes back to zero here)
				if (!any) asm volatile("v_mov_b64 %0, 0" : "+v"(cell));    // (cell = 0, in its own register: as an assignment it cost the common path two moves)
				else {
					asm volatile("");                         // (two scalar branches, not one over a combined lane mask)
					if (ns > 0) {
						uint32_t jps = cur.hv[6];
#pragma unroll
						for (int i = 0; i < NS; i++) {
							if (i >= ns) break;
#ifdef HIBAG_ABL2_SVNOADD                     // (timing ablation: the stored sums are loaded and waited for, never added)
							asm volatile("" :: "v"(sv[i]));
#else
							__hip_atomic_fetch_add(&acc[(int)(jps & 15)][lane], (sv[i] * inv_e) * w_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
							jps >>= 4;
						}
					}
				}
			}
			__builtin_amdgcn_sched_barrier(0);
			ACCUM_STAMP(1);
			// ---- distances on the matrix pipe (their operands have arrived with everything else of the block)
			v16i D0, D1;
			if (eval) {
				v16f d0, d1;
#pragma unroll
				for (int r = 0; r < 16; r++) { d0[r] = 0.0f; d1[r] = 0.0f; }
				const v8i a8 = {arow[0], arow[1], arow[2], arow[3], 0, 0, 0, 0};
				const v8i b0 = {t0[0], t0[1], t0[2], t0[3], 0, 0, 0, 0};
				const v8i b1 = {t1[0], t1[1], t1[2], t1[3], 0, 0, 0, 0};
				const int sbs = lane >= 32 ? HIBAG_FP4_SCALE_B_HI : HIBAG_FP4_SCALE_B_LO;
				d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b0, d0, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, sbs);
				d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b1, d1, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, sbs);
				D0 = __builtin_bit_cast(v16i, d0);
				D1 = __builtin_bit_cast(v16i, d1);
			}
			__builtin_amdgcn_sched_barrier(0);
			ACCUM_STAMP(2);
			// ---- everything of block b + 1, into the registers this block has finished with
			request_lane(cur.hv[2], cur.hv[3], (rel + 1) * 1024, nxt.winv);
			__builtin_amdgcn_sched_barrier(0);
			ACCUM_STAMP(3);
#if defined(HIBAG_ACCUM_PRIO) && HIBAG_ACCUM_PRIO == 1
			__builtin_amdgcn_s_setprio(0);
#elif defined(HIBAG_ACCUM_PRIO) && HIBAG_ACCUM_PRIO == 2
			__builtin_amdgcn_s_setprio(1);
#endif
			// ---- every lane its own sample's distances, then cell += prod * TAB[d] in order
			if (eval_b) {
				block_own_sample(D0, D1, [&](int g) { return live4(2 * g); });
#ifdef HIBAG_ACCUM_STAMPS
				asm volatile("" :: "v"(D0[0]), "v"(D1[0]));
				ACCUM_STAMP(4);
#endif
				// S[p] += v as one LDS floating-point add (ds_add_f64: the same IEEE addition, no register for the old
				// sum, nothing to wait for).  The tile row of the cell that closes at slot i (odd) is field i / 2 of the header's
				// words 4, 5: a fixed place per slot, so one bit-field extract, no shifting along of a packed list.
				const uint32_t jp_lo = cur.hv[4], jp_hi = cur.hv[5];
				auto fin = [&](double c, bool, int slot) {
					uint32_t row;                             // (s_bfe_u32 by hand: the compiler's shift-and-mask form takes two instructions)
					asm volatile("s_bfe_u32 %0, %1, %2" : "=s"(row) : "s"(slot < 16 ? jp_lo : jp_hi), "n"(4 * ((slot >> 1) & 7) | (4 << 16)) : "scc");
					const double v = (c * inv_e) * w_c;
#ifdef HIBAG_ABL2_NOADD                       // (timing ablation: the product is made, the LDS addition is not)
					asm volatile("" :: "v"(v), "s"(row));
#else
					__hip_atomic_fetch_add(&acc[(int)row][lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
				};
				// what the block's first record starts from is a bit of its header (the record before it in the stream closed a cell)
				uint32_t fresh = (cur.hv[1] >> 29) & 1u;
				block_accumulate<ACCUM_G, ACCUM_AHEAD>(fac + (size_t)rel * HIBAG_PLIST_DWORDS, cur.F, endmask, 0u, live4, D0, D1, cell, fresh, tab_s, fin);
			}
			// (the header's word 7 counts as in use to the end of the block: otherwise its register -- free as far as the compiler
			// can see, but still to be written by the header load in flight -- is handed to one of the loads above, which then has to wait for that load)
			asm volatile("" :: "s"(tch0), "s"(tch1), "s"(tch2), "s"(nxt.hv[7]));
			ACCUM_STAMP(5);
		};

		AccumAhead A, Bn;
		A.hv = eh[0];
		A.F = *(ConstPtr<AFG>)fac;
		request_lane(A.hv[7], A.hv[1], 0, A.winv);
#ifdef HIBAG_ACCUM_STAMPS
		stamp_t = __builtin_readcyclecounter();
#endif
		const int nb = be - bb;
		for (int rel = 0;;) {
			one_block(rel, A, Bn);
			if (++rel >= nb) break;
			one_block(rel, Bn, A);
			if (++rel >= nb) break;
		}
		// (a walk that ends on a closed cell leaves `cell` unused: nothing to materialise -- every cell of a tile closes inside the tile's stream)
	}

	// the item's sums, or -- parked -- what the workgroup behind continues from
	// (parked sums leave as sc1 stores, like the loads that pick them up: load_parked / store_parked)
	if (ce < C) for (int j = 0; j < ncell; j++) store_parked(&B.part[(size_t)(p0 + j) * B.n_pad + s], acc[j][lane]);
	else for (int j = 0; j < ncell; j++) B.part[(size_t)(p0 + j) * B.n_pad + s] = acc[j][lane];
	// The workgroup that ends tile 0 of its sample groups also forms their ensemble scalars (k_scalars' loop, classifiers in
	// order): one kernel and its launch gap less on the step.
	if (tile == 0 && ce == C) ensemble_scalars<8>(M, B, s, nullptr);      // (eight loads in flight: sixteen would set the kernel's register count)
	}
#ifdef HIBAG_ACCUM_STAMPS
	__syncthreads();
	if (threadIdx.x < ACCUM_STAMP_N)
		atomicAdd(reinterpret_cast<unsigned long long *>(B.err_dev + 4) + 2000 + threadIdx.x, stamp_s[threadIdx.x]);
	if (threadIdx.x == ACCUM_STAMP_N) atomicAdd(reinterpret_cast<unsigned long long *>(B.err_dev + 4) + 2000 + ACCUM_STAMP_N, (unsigned long long)(be_diag - bb_diag) * ACCUM_WAVES);
#endif
	if (ce < C) handover_post(flag, B.epoch, (uint32_t)ce, B.drop_post == 2 && blockIdx.x == 8 * n_whole);
}

// ---------------------------------------------------------------------------
// k_accum_cells (pass 2, cells read back): S[p] += (cell * (1/total)) * w over the classifiers in order
// (src/LibHLA.cpp:1828 then :1497-1507) with the cell sums pass 1 stored -- 8 bytes per sample, classifier and
// non-empty cell instead of a second evaluation of every haplotype pair; bound by HBM reads.
// Wavefront = (tile of up to HIBAG_TILE cells, 64 samples), the tile's sums in LDS; the four wavefronts of a
// workgroup take four tiles of one sample group (its weights and 1/totals then come from L1 for three of them), and a
// group's workgroups all go to XCD group % 8, so those rows stay in one L2.
#define CELLS_WAVES 4
#ifndef CELLS_OCC
#define CELLS_OCC 4                         // workgroups per CU k_accum_cells is compiled for
#endif
__global__ __launch_bounds__(CELLS_WAVES * HIBAG_WAVE, CELLS_OCC) void k_accum_cells(HibagModelView M, HibagBatchView B)
{
	constexpr int CELLS_V = (HIBAG_TILE + 3) / 4 * 4;     // the cell sums of a visit in registers: requested four at a time
	__shared__ double acc_s[CELLS_WAVES][HIBAG_TILE][HIBAG_WAVE];
	const int n_group = B.n_pad / HIBAG_WAVE;
	const int tq = (M.n_tile + CELLS_WAVES - 1) / CELLS_WAVES;
	const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int group = (jb / tq) * 8 + xcd, tile = (jb % tq) * CELLS_WAVES + wave;
	if (group >= n_group || tile >= M.n_tile) return;
	const int s = group * HIBAG_WAVE + lane;
	const int C = M.n_classifier;
	const int ncell = M.tile_n[tile];
	double (*acc)[HIBAG_WAVE] = acc_s[wave];
#pragma unroll
	for (int q = 0; q < HIBAG_TILE; q++) acc[q][lane] = 0;

	typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
	const ConstPtr<u32x8> ct = as_const(reinterpret_cast<const u32x8 *>(M.ctile)) + tile;
	const double *__restrict__ const group_rows = B.cells + (size_t)group * (size_t)as_const(M.cell_row)[C] * HIBAG_WAVE + lane;
	struct Visit { u32x8 rec; double w, inv; };
	// what classifier c contributes to the tile: its record (one s_load_dwordx8), the lane's weight and 1/total.  Requested two classifiers ahead, so that nothing below waits for a load it has just issued.
	auto visit = [&](int c) {
		Visit x;
		x.rec = ct[(size_t)c * M.n_tile];
		x.w = B.cw[(size_t)c * B.n_pad + s];
		x.inv = B.inv[(size_t)c * B.n_pad + s];
		return x;
	};
	// request the tile's n non-empty cells of the classifier (rows k0 .. k0 + n - 1 of the group's cells), four at a time.
	// (Also where pass 1 skipped the classifier because no sample of the group uses it: the rows then hold stale
	// numbers, which `add` never looks at.)
	auto fetch = [&](const Visit &x, double (&v)[CELLS_V]) {
		const int n = (int)((x.rec[0] >> 8) & 31u);
		const double *__restrict__ rows = group_rows + (size_t)(x.rec[5] & 0x7FFFFFFu) * HIBAG_WAVE;
#pragma unroll
		for (int g = 0; g < HIBAG_TILE; g += 4) {
			if (g >= n) break;
#pragma unroll
			for (int i = g; i < g + 4; i++) v[i] = __builtin_nontemporal_load(rows + (size_t)(i < n ? i : n - 1) * HIBAG_WAVE);
		}
	};
	// S[p] += (cell * (1/total)) * w for those cells, rows in the order of the tile's non-empty list
	auto add = [&](int c, const Visit &x, const double (&v)[CELLS_V]) {
		const bool active = x.w > 0;
		if (__ballot(active) == 0) return;           // nobody in the group uses the classifier (src/LibHLA.cpp:2451)
		const bool poison = __ballot(active && !(fabs(x.inv) <= 1.79769313486231570815e+308)) != 0;
		const double inv_e = active ? x.inv : 0.0;   // inactive lanes keep their sums: (cell * 0) * 0 = +0
		const int n = (int)((x.rec[0] >> 8) & 31u);
		uint64_t jp = ((uint64_t)x.rec[7] << 32) | x.rec[6];
#pragma unroll
		for (int i = 0; i < HIBAG_TILE; i++) {
			if (i >= n) break;
			__hip_atomic_fetch_add(&acc[(int)(jp & 15)][lane], (v[i] * inv_e) * x.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			jp >>= 4;
		}
		if (poison) {                                // empty cells: (0 * inv) * w is NaN where inv is not finite
			const uint32_t *__restrict__ meta = M.tile_meta + ((size_t)c * M.n_tile + tile) * HIBAG_TILE_META;
			for (int i = n; i < ncell; i++) {
				const double t = (0.0 * x.inv) * x.w;
				acc[meta[4 + i] >> 24][lane] += active ? t : 0.0;
			}
		}
	};

	// two classifiers per turn: while classifier c is added, the cells of c + 1 and the records of c + 2 are in flight
	double va[CELLS_V], vb[CELLS_V];
	Visit x0 = visit(0), x1 = visit(C > 1 ? 1 : 0);
	fetch(x0, va);
	for (int c = 0; c < C; c += 2) {
		const Visit x2 = visit(c + 2 < C ? c + 2 : C - 1);
		if (c + 1 < C) fetch(x1, vb);
		add(c, x0, va);
		const Visit x3 = visit(c + 3 < C ? c + 3 : C - 1);
		if (c + 2 < C) fetch(x2, va);
		if (c + 1 < C) add(c + 1, x1, vb);
		x0 = x2; x1 = x3;
	}
	const int p0 = M.tile_p0[tile];
	for (int q = 0; q < ncell; q++) B.part[(size_t)(p0 + q) * B.n_pad + s] = acc[q][lane];
}

#endif
