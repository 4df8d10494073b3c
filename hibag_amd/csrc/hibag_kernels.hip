// hibag_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for HIBAG's
// attribute-bagging prediction hot path: the haplotype-pair posterior loop of
// CAlg_Prediction::_PostProb2 (src/LibHLA.cpp:1769-1830) and the ensemble step
// of CAttrBag_Model::_PredictHLA / PredictHLA (src/LibHLA.cpp:2317-2482).
//
// Mapping: LANE = SAMPLE.  A wavefront holds 64 samples and walks the model's
// loop nest (classifier -> allele pair -> haplotype pair) in the reference's
// order.  The nest depends only on the model, so control flow is wave-uniform, and each lane reproduces the
// reference's rounding sequence for its own sample: results are bit-identical
// to the CPU kernels by construction, with no cross-lane reduction anywhere on
// the numeric path.
//
// Two engines compute the distance d of a pair (bit-identical results):
//   matrix engine 8 d = A . B over 2k+1 positions (h1+h2 against the genotype's signs, h1&h2 -- on the FP4 path a stand-in
//                 for it that is a plain sum of two images -- against [g = 1], an offset term): a small GEMM of 32 records
//                 x 64 samples per block --
//                 up to 30 SNPs on the FP4 matrix path (v_mfma_scale_f32_32x32x64_f8f6f4, e2m1 operands: all 64 K
//                 positions in one instruction per sample half; block scales of 2^-73 make the f32 result the
//                 denormal 8 d * 2^-149, whose bit pattern IS the integer 8 d), 31..32 SNPs on
//                 v_mfma_i32_32x32x32_i8 (two K blocks) -- plus 16 v_permlane32_swap to give every lane its own
//                 sample's column.  The records are generated: each lane gathers its pair's two haplotype
//                 images (nibbles / bytes) from an O(H) table through a 4-byte index pair; what is the same for all lanes
//                 -- the pair's frequency factor, the block's end-of-cell masks -- comes through the scalar cache into
//                 scalar registers.  33..112 SNPs: FP4 again, 28 SNPs per K step, chained through the accumulator.  All real classifiers; the default.
//   VALU engine   d = sum_w popc((W[w] ^ T'[w]) & M'[w]): v_bitop3_b32 + v_bcnt_u32_b32 per 32-bit
//                 word of the stored 3k-bit pair string (W uniform in SGPRs, T'/M' the lane's genotype
//                 masks); classifiers with more than 112 SNPs.  (The per-sample plugin route has kernels of its own:
//                 hibag_sample.hip.)
// In both, what the contract fixes stays on the vector ALU, per lane and in the reference's order:
//     cell += prod * TAB[d]          ds_read_b64 (table in LDS), v_mul_f64 (prod a scalar-register operand), v_add_f64
//
// The normalisation 1/sum of a classifier's posterior needs all of its cells,
// and 64 samples x P cells do not fit on chip, so the pair loop runs twice:
// pass 1 (k_total) produces the in-order total per (sample, classifier) and stores the sums of the cells with many
// pairs, pass 2 (k_accum) evaluates the other cells again (or reads the stored sums back), scales each cell and adds it to
// the ensemble sum of its tile of cells, held in LDS (DESIGN.md "Why two passes").
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA fusion: the
// reference multiplies and adds with separate roundings).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "hibag_device.h"
#include "hibag_kernels.h"
#include "hibag_ablation.h"

#define NA_INTEGER (-2147483647 - 1)
#define CH HIBAG_CHUNK
#ifndef HIBAG_GATHER_DEPTH
#define HIBAG_GATHER_DEPTH 1                // blocks of look-ahead of the haplotype-entry gathers
#endif
#define BLOCK_WAVES HIBAG_BLOCK_WAVES         // wavefronts per workgroup (each on its own work item; hibag_device.h)
#define BLOCK_THREADS (BLOCK_WAVES * HIBAG_WAVE)
#ifndef TOTAL_G
#define TOTAL_G 8                           // records of pass 1 whose look-ups are in flight together (and whose factors arrive in one scalar load)
#endif
// Workgroups per CU pass 1 is compiled for, twice: 5 (96 VGPRs: every engine) and 6 (80 VGPRs: the one-step FP4 block loop
// fits, the int8 and VALU-engine loops would spill, so that build carries the FP4 loop alone and serves models whose work
// items are all one-step FP4 classifiers -- k_total's FP4ONLY; no kernel of the path has a private segment).  Six wavefronts per SIMD are 1.5 - 4 % faster once a pass is two rounds of resident
// workgroups or more (10,000 samples of the HLA-B shape: 0.80 -> 0.775 ms; 8,192 of DRB1: 14.1 -> 13.6 ms), and slower below that,
// where a pass lasts as long as its longest work item, which six wavefronts sharing a SIMD stretch (4,096 samples of DRB1:
// 7.1 -> 7.9 ms): the launcher chooses by the number of work items (hibag_launch_total).  (4 -> 128 VGPRs: measured slower.)
#ifndef HIBAG_TOT_OCC
#define HIBAG_TOT_OCC 5
#endif
#ifndef HIBAG_TOT_OCC_MANY
#define HIBAG_TOT_OCC_MANY 6
#endif
#ifndef TOTAL_OWN
#define TOTAL_OWN false                     // true: pass 1 over prebuilt rows with four matrix instructions per block and NO lane swaps (walk_blocks, OWN) -- bit-identical, measured 2-5 % slower than two and sixteen swaps
#endif
#ifndef ACCUM_AHEAD
#define ACCUM_AHEAD true                    // pass 2: the next group's table look-ups requested before this group is added up (-0.6 %, six registers; false: the wait right behind the look-ups)
#endif
#ifndef ACCUM_G
#define ACCUM_G 4                           // the same for pass 2
#endif
#define ACCUM_TAB_N 64                      // table entries pass 2 stages: it evaluates one-step FP4 classifiers only
static_assert(2 * HIBAG_FP4_MAX_SNPS + 1 <= ACCUM_TAB_N, "pass 2's table must cover every distance of a one-step FP4 classifier");
#define ACCUM_WAVES HIBAG_ACCUM_WAVES         // wavefronts per workgroup of pass 2 (sample groups that share a tile's blocks in L1)

// The device code, in the order of a step (each file says what it holds):
#include "hibag_k_engine.h"
#include "hibag_k_pack.h"
#include "hibag_k_pass1.h"
#include "hibag_k_pass2.h"
#include "hibag_k_vote.h"
#include "hibag_k_finish.h"

// ---------------------------------------------------------------------------
// launchers (host side, no synchronisation, no allocation)

static inline dim3 grid1(int n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

void hibag_launch_pack(const HibagModelView &M, const HibagBatchView &B, const int32_t *d_geno, int row_len,
	const int32_t *d_col, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st)
{
	if (M.n_classifier == 0 || M.n_snp == 0) return;
	hipLaunchKernelGGL(k_codes, dim3(B.n_pad / 64, (M.n_snp + 63) / 64), dim3(256), 0, st, M, B, d_geno,
		d_col ? row_len : M.n_snp, d_col, d_flip, d_codes);
	hipLaunchKernelGGL(k_pack, dim3(B.n_pad / HIBAG_WAVE, (M.n_classifier + PACK_WAVES - 1) / PACK_WAVES), dim3(PACK_WAVES * HIBAG_WAVE), 0, st, M, B,
		(const uint8_t *)d_codes);
}

// the genotypes as a SNP-major matrix int32 [rows][ld] (k_codes_rows)
void hibag_launch_pack_rows(const HibagModelView &M, const HibagBatchView &B, const int32_t *d_geno, size_t ld,
	const int32_t *d_col, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st)
{
	if (M.n_classifier == 0 || M.n_snp == 0) return;
	hipLaunchKernelGGL(k_codes_rows, dim3((B.n_pad + 255) / 256, M.n_snp), dim3(256), 0, st, M, B, d_geno, ld, d_col, d_flip, d_codes);
	hipLaunchKernelGGL(k_pack, dim3(B.n_pad / HIBAG_WAVE, (M.n_classifier + PACK_WAVES - 1) / PACK_WAVES), dim3(PACK_WAVES * HIBAG_WAVE), 0, st, M, B,
		(const uint8_t *)d_codes);
}

void hibag_launch_pack_bed(const HibagModelView &M, const HibagBatchView &B, const uint8_t *d_bed, int mode,
	size_t stride, int samp0, const int32_t *d_snp_row, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st)
{
	if (M.n_classifier == 0 || M.n_snp == 0) return;
	hipLaunchKernelGGL(k_bed_codes, dim3(B.n_pad / 64, (M.n_snp + 3) / 4), dim3(256), 0, st, M, B, d_bed, mode, stride,
		samp0, d_snp_row, d_flip, d_codes);
	hipLaunchKernelGGL(k_pack, dim3(B.n_pad / HIBAG_WAVE, (M.n_classifier + PACK_WAVES - 1) / PACK_WAVES), dim3(PACK_WAVES * HIBAG_WAVE), 0, st, M, B,
		(const uint8_t *)d_codes);
}

void hibag_launch_bed_geno(const uint8_t *d_bed, int mode, size_t stride, int n_samp, int n_save,
	const int32_t *d_sel, int32_t *d_geno, hipStream_t st)
{
	if (n_samp <= 0 || n_save <= 0) return;
	hipLaunchKernelGGL(k_bed_geno, dim3((n_samp + 63) / 64, (n_save + 63) / 64), dim3(256), 0, st, d_bed, mode, stride,
		n_samp, n_save, d_sel, d_geno);
}

// resident workgroups of a kernel on the current device (0 = unknown)
template <class F>
static int resident_blocks(F kernel, int threads, size_t dyn_lds = 0)
{
	int per_cu = 0, cus = 0, dev = 0;
	(void)hipGetDevice(&dev);
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds) != hipSuccess) return 0;
	return per_cu > 0 && cus > 0 ? per_cu * cus : 0;
}

#ifdef HIBAG_ABL2_LDSPAD                      // (timing ablation, hibag_ablation.h: pass 2 at fewer resident workgroups, same code)
#define ACCUM_DYN_LDS HIBAG_ABL2_LDSPAD
#else
#define ACCUM_DYN_LDS 0
#endif

// chunks per item of the last rounds of passes 1 and 2 ("hand-overs"; HIBAG_TAIL_K=1: undivided items only).
// A hand-over costs about as much as a tenth of a block list of the benchmark model, so 4 chunks there (2 and 8 measure 1-3 %
// worse); items of several thousand blocks (the DRB1 shape: 3,900) take 8 - 16 (-4 % against 4): one chunk per
// `blocks_per_item` / 256, between 4 and 12.
// `forced` = HibagBatchView::tail_k: 1 after a failed hand-over on the model (its launches then have no hand-overs at all).
// (Pass 2, whose items are a tile's ~85 short blocks: 2 chunks -- 0.522-0.527 ms against 0.532 with 4, same box, three runs each.)
static int tail_chunks(int forced, long long blocks_per_item = 0, int at_least = 4)
{
	if (forced > 0) return forced;
	static const int k = getenv("HIBAG_TAIL_K") ? std::max(1, std::min(64, atoi(getenv("HIBAG_TAIL_K")))) : 0;
	return k ? k : (int)std::max<long long>(at_least, std::min(12ll, blocks_per_item / 256));
}

// Resident workgroups of the chunked kernels on the current device; the model keeps them (HibagModelView::slots_*), so
// that models on different devices, and the two instantiations of k_total, each get their own figure.
void hibag_query_slots(int total[4], int *accum)
{
	// [STORE][many]: k_total<false, 5>, <false, 6>, <true, 5>, <true, 6>
	total[0] = resident_blocks(k_total<false, HIBAG_TOT_OCC, 0>, BLOCK_THREADS);
	total[1] = resident_blocks(k_total<false, HIBAG_TOT_OCC_MANY, 2>, BLOCK_THREADS);
	total[2] = resident_blocks(k_total<true, HIBAG_TOT_OCC, 0>, BLOCK_THREADS);
	total[3] = resident_blocks(k_total<true, HIBAG_TOT_OCC_MANY, 2>, BLOCK_THREADS);
	*accum = resident_blocks(k_accum, ACCUM_WAVES * HIBAG_WAVE, ACCUM_DYN_LDS);
}

void hibag_launch_total(const HibagModelView &M, const HibagBatchView &B, hipStream_t st, const HibagSideStream &side, bool vote)
{
	if (M.n_classifier == 0) return;
	const bool wide = M.n_wide > 0;
	if (wide) {
		// the classifiers of several K steps: their own kernel, beside k_total (more registers than k_total's hot loop may have)
		const unsigned gq = (unsigned)((B.n_pad / HIBAG_WAVE + BLOCK_WAVES - 1) / BLOCK_WAVES);
		HibagModelView W = M;                         // k_total_scan over the classifiers of several K steps that were cut into segments
		W.split_cls = M.wide_scan;
		const hipStream_t ws = side.stream ? side.stream : st;
		if (side.stream) {
			(void)hipEventRecord(side.fork, st);
			(void)hipStreamWaitEvent(side.stream, side.fork, 0);
		}
		// (a classifier of several K steps without haplotypes has no segment: its total still has to be written)
		// (segments are either all whole classifiers or none: hibag_model.hip finalize_model)
		if (M.n_wide_seg > 0 && M.n_wide_scan < M.n_wide) hipLaunchKernelGGL(k_total_wide<true>, dim3(gq, M.n_wide_seg), dim3(BLOCK_THREADS), 0, ws, M, B);
		else if (M.n_wide_seg > 0) hipLaunchKernelGGL(k_total_wide<false>, dim3(gq, M.n_wide_seg), dim3(BLOCK_THREADS), 0, ws, M, B);
		if (M.n_wide_scan > 0) hipLaunchKernelGGL(k_total_scan, dim3(B.n_pad / 64, M.n_wide_scan), dim3(64), 0, ws, W, B);
		if (side.stream) (void)hipEventRecord(side.join, side.stream);
	}
	if (M.n_item_whole == 0) {                        // (every classifier is one of those)
		if (wide && side.stream) (void)hipStreamWaitEvent(st, side.join, 0);
		return;
	}
	const unsigned gx = (unsigned)((B.n_pad / HIBAG_WAVE + BLOCK_WAVES - 1) / BLOCK_WAVES);
	// A classifier far heavier than the rest (VALU engine, > 112 SNPs) is only worth cutting up when its
	// single-wavefront walk would outlast the rest of the pass, i.e. for small batches.
	HibagModelView V = M;
	const double groups = (double)(B.n_pad / HIBAG_WAVE);
	// (the heavy walk starts with the pass, so it only matters once it lasts about as long as everything else)
	const bool split = M.n_split > 0 && 1.25 * M.split_heavy_ns > M.split_rest_ns * groups / 4096.0;
	V.item = split ? M.item_split : M.item_whole;
	V.n_item = split ? M.n_item_split : M.n_item_whole;
	if (!split) V.n_split = 0;
	// more items than resident workgroups: the last, incomplete round and the full round before it go in K chunks each
	const unsigned n = gx * (unsigned)V.n_item;
	// two rounds of the denser build's resident workgroups or more: six workgroups per CU, otherwise five (above)
	static const int occ_env = getenv("HIBAG_TOT_OCC") ? atoi(getenv("HIBAG_TOT_OCC")) : 0;      // (diagnostic: 5 or 6)
	const int sbase = (M.store_cells && !vote) ? 2 : 0;       // (the vote's build stores no cell sums: the occupancy figures of the non-storing one)
	const int slots_many = M.slots_total[sbase + 1];
	// (the denser build holds the one-step FP4 loop only: models with other work items always take the general one)
	const bool many = M.all_fp4 && !split && (occ_env ? occ_env == HIBAG_TOT_OCC_MANY : (slots_many > 0 && n >= 2u * (unsigned)slots_many));
	const int slots = M.slots_total[sbase + (many ? 1 : 0)];
	unsigned n_whole = n, rest = 0, stride = 8, K = 1;
	static const int k1_env = getenv("HIBAG_TAIL_K1") ? std::max(1, std::min(64, atoi(getenv("HIBAG_TAIL_K1")))) : 0;     // (diagnostic: pass 1 only)
	const int k_pass1 = (k1_env && B.tail_k == 0) ? k1_env : tail_chunks(B.tail_k, M.p1_blocks / std::max(M.n_classifier, 1));
	// (the majority vote's record log is not handed over between chunks: its items stay whole)
	if (!vote && k_pass1 > 1 && slots > 0 && n > (unsigned)slots) {
		K = (unsigned)k_pass1;
		rest = n % (unsigned)slots + (unsigned)slots;
		n_whole = n - rest;
		stride = (rest + 7) / 8 * 8;
	}
	const dim3 grid(n_whole + (rest ? K * stride : 0));
	// the build: storing (models whose pass 2 reads cell sums back) / plain / voting (no second pass: nothing stored, records
	// logged) x the walk -- every engine at five workgroups per CU, or one of the two FP4-only builds at six
#define LAUNCH_TOTAL(STORE, OCC, WALK, VOTE) hipLaunchKernelGGL((k_total<STORE, OCC, WALK, VOTE>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K)
#define LAUNCH_WALK(STORE, VOTE) do { if (!many) LAUNCH_TOTAL(STORE, HIBAG_TOT_OCC, 0, VOTE);                                 \
		else if (M.p1_prebuilt) LAUNCH_TOTAL(STORE, HIBAG_TOT_OCC_MANY, 2, VOTE); else LAUNCH_TOTAL(STORE, HIBAG_TOT_OCC_MANY, 1, VOTE); } while (0)
	if (vote) LAUNCH_WALK(false, true);
	else if (M.store_cells) LAUNCH_WALK(true, false);
	else LAUNCH_WALK(false, false);
#undef LAUNCH_WALK
#undef LAUNCH_TOTAL
	if (split)
		hipLaunchKernelGGL(k_total_scan, dim3(B.n_pad / 64, V.n_split), dim3(64), 0, st, V, B);
	if (wide && side.stream) (void)hipStreamWaitEvent(st, side.join, 0);
}

void hibag_launch_accum(const HibagModelView &M, const HibagBatchView &B, hipStream_t st)
{
	if (M.store_cells == 1 && M.n_classifier > 0) {    // pass 1 stored every cell: read them back
		const unsigned groups_x = (unsigned)((B.n_pad / HIBAG_WAVE + 7) / 8), tq = (unsigned)((M.n_tile + CELLS_WAVES - 1) / CELLS_WAVES);
		hipLaunchKernelGGL(k_accum_cells, dim3(8 * groups_x * tq), dim3(CELLS_WAVES * HIBAG_WAVE), 0, st, M, B);
		return;
	}
	const unsigned n_group = (unsigned)(B.n_pad / HIBAG_WAVE);
	const unsigned n = 8u * (((n_group + 7) / 8 + ACCUM_WAVES - 1) / ACCUM_WAVES) * (unsigned)M.n_tile;   // work items, see k_accum
	if (n == 0 || M.n_classifier == 0) {
		// no classifier: the ensemble sums are all zero (src/LibHLA.cpp:1491-1495)
		(void)hipMemsetAsync(B.part, 0, (size_t)M.n_cell * B.n_pad * sizeof(double), st);
		return;
	}
	// more items than resident workgroups: the last, incomplete round and the full round before it go in K chunks each
	const int slots = M.slots_accum;
	const unsigned nx = n / 8, sx = (unsigned)slots / 8;
	unsigned n_whole = nx, K = 1;
	if (tail_chunks(B.tail_k, 0, 2) > 1 && sx > 0 && nx > sx) {
		K = (unsigned)tail_chunks(B.tail_k, 0, 2);
		n_whole = nx - (nx % sx + sx);
	}
	hipLaunchKernelGGL(k_accum, dim3(8 * (n_whole + K * (nx - n_whole))), dim3(ACCUM_WAVES * HIBAG_WAVE), ACCUM_DYN_LDS, st, M, B, (int)n_whole, (int)K);
}

void hibag_launch_vote(const HibagModelView &M, const HibagBatchView &B, int *d_best_cell, hipStream_t st)
{
	if (M.n_classifier > 0) {
		// pass 1 (hibag_launch_total with vote = true) has logged the records of every one-step matrix-engine classifier
		hipLaunchKernelGGL(k_vote_pick, dim3(B.n_pad / 64, M.n_classifier), dim3(64), 0, st, M, B, d_best_cell);
		if (M.n_wide > 0) hipLaunchKernelGGL(k_vote_scan, dim3(B.n_pad / 64, M.n_wide), dim3(64), 0, st, M, B, d_best_cell);
		if (M.n_valu > 0) {
			const unsigned gx = (unsigned)((B.n_pad / HIBAG_WAVE + BLOCK_WAVES - 1) / BLOCK_WAVES);
			hipLaunchKernelGGL(k_vote_best_valu, dim3(gx, M.n_classifier), dim3(BLOCK_THREADS), 0, st, M, B, d_best_cell);
		}
	}
	hipLaunchKernelGGL(k_vote_tally, grid1(B.n_pad, 64), dim3(64), 0, st, M, B, (const int *)d_best_cell);
}

void hibag_launch_scalars(const HibagModelView &M, const HibagBatchView &B, const int *d_best_cell, hipStream_t st)
{
	if (!d_best_cell && M.store_cells != 1 && M.n_classifier > 0 && M.n_tile > 0) {
		// pass 2 was k_accum: its tile-0 workgroups have written the scalars; what is left is the rare NaN work and the mark
		// of a failed hand-over
		hipLaunchKernelGGL(k_nan_cells, dim3((unsigned)std::max(M.n_tile, B.n_pad / 64)), dim3(64), 0, st, M, B);
		return;
	}
	hipLaunchKernelGGL(k_scalars, grid1(B.n_pad, 64), dim3(64), 0, st, M, B, d_best_cell);
}

void hibag_launch_finish(const HibagModelView &M, const HibagBatchView &B, double *d_part,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching,
	double *d_dosage, double *d_postprob, hipStream_t st)
{
	// (One read of the ensemble sums instead of three -- call and dosage from LDS chunks of 128 cells, 32 samples x 32 threads per
	// workgroup -- was built and measured in round 5: 76 us against 47.  The three-read kernel has 785 workgroups of independent
	// loads and runs at 6.5 TB/s; the one-read one has 314 workgroups with a barrier per chunk.  profiles/r05_pass2_notes.txt item 9.)
	if (d_dosage) {
		const unsigned n_group = (unsigned)(B.n_pad / 64);
		hipLaunchKernelGGL(k_finish, dim3(n_group * (1u + (unsigned)((M.n_hla + FIN_SEG - 1) / FIN_SEG))), dim3(64 * FIN_SEG), 0, st,
			M, B, (const double *)d_part, d_H1, d_H2, d_max_prob, d_matching, d_dosage);
	} else {
		hipLaunchKernelGGL(k_finish_call, dim3(B.n_pad / 64), dim3(64 * FIN_SEG), 0, st, M, B, (const double *)d_part,
			d_H1, d_H2, d_max_prob, d_matching);
	}
	if (d_postprob)
		hipLaunchKernelGGL(k_finish_prob, dim3(B.n_pad / 64, (M.n_cell + 63) / 64), dim3(256), 0, st,
			M, B, (const double *)d_part, d_postprob);
}
