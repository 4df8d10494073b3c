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
#ifndef BLOCK_WAVES
#define BLOCK_WAVES 4                       // wavefronts per workgroup (each on its own work item)
#endif
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
#ifndef ACCUM_G
#define ACCUM_G 4                           // the same for pass 2
#endif
#define ACCUM_TAB_N 64                      // table entries pass 2 stages: it evaluates one-step FP4 classifiers only
static_assert(2 * HIBAG_FP4_MAX_SNPS + 1 <= ACCUM_TAB_N, "pass 2's table must cover every distance of a one-step FP4 classifier");
#ifndef ACCUM_WAVES
#define ACCUM_WAVES 4                       // wavefronts per workgroup of pass 2 (sample groups that share a tile's lists in L1)
#endif

// The lane's genotype for one classifier: XOR mask x (= T') and AND mask m (= M').
template <int NWP>
struct LaneMask {
	uint32_t x[NWP], m[NWP];
};

template <int NWP>
__device__ __forceinline__ void load_masks(const HibagBatchView &B, int row0, int s, LaneMask<NWP> &L)
{
#pragma unroll
	for (int w = 0; w < NWP; w++) {
		L.x[w] = B.masks[(size_t)(row0 + w) * B.n_pad + s];
		L.m[w] = B.masks[(size_t)(row0 + NWP + w) * B.n_pad + s];
	}
	// The masks are used by every instruction of the loops that follow: make the
	// loads complete here (an empty asm that passes the registers through) instead
	// of leaving one s_waitcnt vmcnt per mask inside the loop.  It must be a plain
	// asm: a "memory" clobber, asm volatile or the s_waitcnt builtin all make the
	// compiler assume the stream may have been written, and the stream loads then
	// become per-lane VMEM instead of scalar s_load.
#pragma unroll
	for (int w = 0; w < NWP; w++) asm("" : "+v"(L.x[w]), "+v"(L.m[w]));
}

// popc(x) + acc in one VALU op.  Written as asm so that the compiler keeps the
// distance a single chained sum (it otherwise scales every partial count by 8
// for the table address, one shift per word).
__device__ __forceinline__ int bcnt_acc(uint32_t x, int acc)
{
	int r;
	asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
	return r;
}

// One chunk of CH pair records at cp (wave-uniform: s_load into SGPRs):
//   cell += prod_r * TAB[d_r], r in order -- the statement
//   sum += (2*f1*f2) * TAB[hamm_d]   of src/LibHLA.cpp:1786-1813 (ADD_FREQ_MUTANT, src/LibHLA.h:223).
// The CH table look-ups are independent, so their LDS latencies overlap.
template <int NWP>
__device__ __forceinline__ double chunk_apply(double cell, const uint32_t *__restrict__ cp,
	const LaneMask<NWP> &L, const double *tab_s)
{
	double t[CH];
#pragma unroll
	for (int r = 0; r < CH; r++) {
		int d = __popc((cp[r] ^ L.x[0]) & L.m[0]);
#pragma unroll
		for (int w = 1; w < NWP; w++) d = bcnt_acc((cp[w * CH + r] ^ L.x[w]) & L.m[w], d);
		t[r] = tab_s[d];
	}
	const double *__restrict__ pr = reinterpret_cast<const double *>(cp + NWP * CH);
#pragma unroll
	for (int r = 0; r < CH; r++) cell += pr[r] * t[r];
	return cell;
}

// The strictly ordered sum of one allele-pair cell: n consecutive chunks at cp.
template <int NWP>
__device__ __forceinline__ double cell_sum(uint32_t n, const uint32_t *__restrict__ &cp,
	const LaneMask<NWP> &L, const double *tab_s)
{
	double cell = 0;
	for (; n > 0; n--) {
		cell = chunk_apply<NWP>(cell, cp, L, tab_s);
		cp += HIBAG_CHUNK_DWORDS(NWP);
	}
	return cell;
}

// ---------------------------------------------------------------------------
// MFMA engine.  The distance of src/LibHLA.cpp:747-819 is, SNP by SNP, |g - h1 - h2| for a
// called genotype g and 0 for a missing one:
//     g = 0: h1 + h2      g = 2: 2 - h1 - h2      g = 1: [h1 == h2] = 1 - h1 - h2 + 2 h1 h2
// i.e. an integer dot product over 2k + 1 positions (K layout in hibag_device.h),
//     8 d = sum_s (h1_s + h2_s) * 8 t_s  +  sum_s (h1_s & h2_s) * 16 [g_s == 1]  +  8 * (2 #[g == 2] + #[g == 1]),
// t_s = +1 / -1 / -1 / 0 for g_s = 0 / 1 / 2 / missing: D[record][sample] = A[record][:] . B[:][sample] is a
// small int8 GEMM with K = 32 * nkb (nkb = 2 for 16..31 SNPs).  One v_mfma_i32_32x32x32_i8 gives the exact
// distances of 32 records x 32 samples (scaled by 8: the byte offset of TAB[d]); two (sample halves) cover
// the wavefront's 64 samples, and 16 v_permlane32_swap move every lane's own-sample column into its
// registers.  The A rows are not stored anywhere: lane l builds the row of record l % 32 (K half l / 32)
// from the two haplotype words of its pair, fetched from the model's O(H) haplotype table through the
// 4-byte index pair of the slot, and the frequency factor (2 f1) f2 with the reference's rounding
// (src/LibHLA.cpp:1786-1813).  The FP64 accumulation below is per lane, in the reference's order, so
// results stay bit-identical to the CPU kernels.
// Used for classifiers with at most 112 SNPs; wider ones use the VALU engine above.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// Model tables read through the constant address space: a wave-uniform load from it is a scalar load (s_load) wherever
// it stands.  Through a plain pointer the compiler only uses scalar loads while no store or atomic of the kernel can
// precede them -- with the hand-over flags in the kernels, the per-classifier record loads of pass 2 had silently become
// vector loads + eight v_readfirstlane each, waited for on the spot.  (The model is never written by a kernel.)
template <class T> using ConstPtr = const __attribute__((address_space(4))) T *;
template <class T> __device__ __forceinline__ ConstPtr<T> as_const(const T *p) { return (ConstPtr<T>)(uintptr_t)p; }

// Template tag of the walk for FP4 classifiers with more than one K step (33 .. 112 SNPs); HibagModelView::engine says
// HIBAG_ENGINE_FP4 for them, n_step > 1.
#define HIBAG_ENGINE_FP4W 4

// Where the further K steps of such a classifier find their B operands (all wave-uniform: the lane's address is only
// formed where a step needs it): step j, sample half n = bt[((bt_row + 2 j + n) * n_group + group) * 64 + lane]
struct WideSrc {
	const uint4 *bt = nullptr;
	size_t n_group = 0;
	int bt_row = 0, group = 0;
	int nstep = 1;
};

struct LaneOperand {
	v4i b[2][2];        // B operand of sample half n, K block kb (MFMA lane layout); the FP4 engine uses b[n][0] only
	int bias[2];        // I8S (32 SNPs) only: the lane's distance offset (times 8) for each sample half
};

// ENG = HIBAG_ENGINE_FP4 / _I8 / _I8S
template <int ENG>
__device__ __forceinline__ void load_operand_row(const HibagBatchView &B, int bt_row, int c, int group,
	int lane, LaneOperand &T)
{
	constexpr int NKB = (ENG == HIBAG_ENGINE_FP4 || ENG == HIBAG_ENGINE_FP4W) ? 1 : 2;
	const size_t n_group = (size_t)(B.n_pad / HIBAG_WAVE);
#pragma unroll
	for (int n = 0; n < 2; n++) {
#pragma unroll
		for (int kb = 0; kb < NKB; kb++) {
			const uint4 v = B.bt[((size_t)(bt_row + n * NKB + kb) * n_group + group) * HIBAG_WAVE + lane];
			T.b[n][kb] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
		}
		T.bias[n] = ENG == HIBAG_ENGINE_I8S ? B.bias[((size_t)(2 * c + n) * n_group + group) * HIBAG_WAVE + lane] : 0;
	}
}

__device__ __forceinline__ WideSrc wide_src(const HibagBatchView &B, int bt_row, int nstep, int group)
{
	WideSrc w;
	w.bt = B.bt; w.n_group = (size_t)(B.n_pad / HIBAG_WAVE); w.bt_row = bt_row; w.group = group; w.nstep = nstep;
	return w;
}

// 16 bits -> 16 bytes (bit i -> byte i = 0/1): per nibble (n * 0x00204081) & 0x01010101
__device__ __forceinline__ v4i expand_bits16(uint32_t x)
{
	v4i r;
#pragma unroll
	for (int q = 0; q < 4; q++) r[q] = (int)((((x >> (4 * q)) & 0xFu) * 0x00204081u) & 0x01010101u);
	return r;
}

// 8 bits -> 8 nibbles (bit i -> nibble i = 0/1)
__device__ __forceinline__ uint32_t expand_bits8_nibbles(uint32_t x)
{
	x = (x | (x << 12)) & 0x000F000Fu;
	x = (x | (x << 6)) & 0x03030303u;
	x = (x | (x << 3)) & 0x11111111u;
	return x;
}

// The lane's constant part of an FP4 A row (K layout in hibag_device.h): lanes 0..31 own the K positions 0..31
// (nibbles k, k+1 = 1, 4 -> codes 2, 6), lanes 32..63 the positions 32..63 (nibbles k, k+1 = 4, 4 -> 6, 6).
__device__ __forceinline__ v4i fp4_offset_term(int k, int lane)
{
	const unsigned __int128 c = (unsigned __int128)(lane < 32 ? 0x62u : 0x66u) << (4 * k);
	return v4i{(int)(uint32_t)c, (int)(uint32_t)(c >> 32), (int)(uint32_t)(c >> 64), (int)(uint32_t)(c >> 96)};
}

// Issue the MFMAs of one block: acc_n[r] of lane l = 8 x distance of record
// 8(r/4) + 4(l/32) + r%4 to sample (l%32) of sample half n.
// e1, e2 = this lane's 16 bytes of the two haplotypes' images of record (lane % 32):
//   I8 / I8S  bytes 16 (lane / 32) .. + 15 of the byte images; K block 0 = e1 + e2, K block 1 = e1 & e2; the value 8
//             at K position 31 (byte 15 of the upper K half of block 0) meets the sample's offset term
//   FP4       one K step: this lane's nibble image (lanes 0..31 the sum image, lanes 32..63 the pair image), A = e1 + e2;
//             several K steps: the one nibble image (codes 0 / 2), lanes 0..31 carry e1 + e2, lanes 32..63 e1 & e2, each plus
//             its constant nibbles `cterm`; the f32 result is the denormal 8 d * 2^-149, i.e. its bits are the integer 8 d
// One K step of the FP4 distance: d_n += A x B_n for the two sample halves, A built from this lane's images.
__device__ __forceinline__ void fp4_step(const v4i &e1, const v4i &e2, int lane, const v4i &cterm, const v4i &b0v, const v4i &b1v,
	v16f &d0, v16f &d1)
{
	const bool upper = lane >= 32;
	v4i a;
	if (upper) {
#pragma unroll
		for (int d = 0; d < 4; d++) a[d] = (e1[d] & e2[d]) | cterm[d];
	} else {
#pragma unroll
		for (int d = 0; d < 4; d++) a[d] = e1[d] + e2[d] + cterm[d];       // nibbles 0 / 2 / 4 and the constants: no carry
	}
	const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
	const v8i b0 = {b0v[0], b0v[1], b0v[2], b0v[3], 0, 0, 0, 0};
	const v8i b1 = {b1v[0], b1v[1], b1v[2], b1v[3], 0, 0, 0, 0};
	const int sb = upper ? HIBAG_FP4_SCALE_B_HI : HIBAG_FP4_SCALE_B_LO;
	d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b0, d0, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, sb);
	d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b1, d1, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, sb);
}

template <int ENG>
__device__ __forceinline__ void block_mfma(const v4i &e1, const v4i &e2, int lane, const v4i &cterm, const LaneOperand &T,
	v16i &acc0, v16i &acc1)
{
	const bool upper = lane >= 32;
	if (ENG == HIBAG_ENGINE_FP4) {
		// one K step: each lane has fetched ITS image of the two haplotypes -- lanes 0..31 the "sum" image, whose nibbles add up
		// to the A row of the lower K half (h1 + h2; the offset digits' constants 1, 4 as 0.5 + 0.5, 1.5 + 1.5), lanes 32..63 the
		// "pair" image, whose nibbles add up to w = 0 / 1.5 / 4 (constants 4, 4) -- so the row is ONE add per dword for all lanes
		v16f d0, d1;
#pragma unroll
		for (int r = 0; r < 16; r++) { d0[r] = 0.0f; d1[r] = 0.0f; }
		v4i a;
#pragma unroll
		for (int d = 0; d < 4; d++) a[d] = e1[d] + e2[d];
		const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
		const v8i b0 = {T.b[0][0][0], T.b[0][0][1], T.b[0][0][2], T.b[0][0][3], 0, 0, 0, 0};
		const v8i b1 = {T.b[1][0][0], T.b[1][0][1], T.b[1][0][2], T.b[1][0][3], 0, 0, 0, 0};
		const int sb = upper ? HIBAG_FP4_SCALE_B_HI : HIBAG_FP4_SCALE_B_LO;
		if (ABL_NOMFMA) { abl_fake_distances(a8, b0, b1, sb, d0, d1); }
		else {
			d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b0, d0, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, sb);
			d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b1, d1, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, sb);
		}
		acc0 = __builtin_bit_cast(v16i, d0);
		acc1 = __builtin_bit_cast(v16i, d1);
		return;
	}
	if (ENG == HIBAG_ENGINE_I8S) {
		// 32 SNPs: no K position is left for the offset term, it starts the accumulators.
		// The empty asm makes the offsets opaque per block: otherwise the two 16-register splats are
		// hoisted out of the block loop and cost 32 VGPRs for its whole duration.
		int b0 = T.bias[0], b1 = T.bias[1];
		asm("" : "+v"(b0), "+v"(b1));
#pragma unroll
		for (int r = 0; r < 16; r++) { acc0[r] = b0; acc1[r] = b1; }
	} else {
#pragma unroll
		for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }      // folds into the MFMA's inline-constant C operand
	}
	const int off3 = (upper && ENG != HIBAG_ENGINE_I8S) ? (8 << 24) : 0;   // K position 31
	v4i a0 = e1 + e2;                                                   // bytes 0/1/2: no carry between bytes
	const v4i both = e1 & e2;
	a0[3] |= off3;
	acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, T.b[0][0], acc0, 0, 0, 0);
	acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, T.b[1][0], acc1, 0, 0, 0);
	acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(both, T.b[0][1], acc0, 0, 0, 0);
	acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(both, T.b[1][1], acc1, 0, 0, 0);
}

// Swap the upper lanes of half 0 with the lower lanes of half 1: afterwards every
// lane holds its OWN sample: record i = 8g + q  ->  q < 4 ? D0[4g + q] : D1[4g + q - 4].
// Only the register groups that hold valid records are moved (a partly filled last block).
__device__ __forceinline__ void block_own_sample(v16i &D0, v16i &D1, int n_valid)    // in place: (acc0, acc1) -> (D0, D1)
{
#pragma unroll
	for (int g = 0; g < 4; g++) {
		if (8 * g >= n_valid) break;
		if (ABL_NOSWAP) continue;
#pragma unroll
		for (int r = 4 * g; r < 4 * g + 4; r++) {
			const auto sw = __builtin_amdgcn_permlane32_swap(D0[r], D1[r], false, false);
			D0[r] = sw[0]; D1[r] = sw[1];
		}
	}
}

// cell += prod_i * TAB[d_i] for the first n_valid records of a block, in order;
// `fin(cell, stored)` at every record that closes a cell (end mask, store mask; cells are padded to
// an even number of records, so only odd positions can close one).
// The factors prod_i are wave-uniform: they come from HibagModelView::pfac through the SCALAR cache, G at a time
// (one s_load), and multiply as scalar-register operands -- no LDS traffic, no vector register, no instruction to
// make them.  `fac` = the block's 32 factors, `F` = the first G of them, requested by the caller at the top of the
// block.  Scalar loads share the lgkmcnt counter with the table look-ups and return out of order, so the wait for a
// group's look-ups also waits for every scalar load in flight: the NEXT group's factors are therefore requested
// right behind that wait (the first product), and have this group's arithmetic and the next group's look-ups to arrive.
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x8 __attribute__((ext_vector_type(8)));
template <int G> struct FactorGroup;
template <> struct FactorGroup<4> { typedef f64x4 type; };
template <> struct FactorGroup<8> { typedef f64x8 type; };

// G = records whose table look-ups are in flight together
//
// Where a cell closes the sum does not go back to zero and take the next product on top -- `cell = 0; cell += x` -- it simply
// STARTS with the next product, `cell = x`: the same value bit for bit (0 + x = x for every x the path can produce: x is +0,
// positive or NaN, never -0), one instruction instead of three (a move, a multiplication into a temporary, an addition).  Inside a
// group the choice is part of the branch that closes the cell; across groups and blocks the wave-uniform `fresh` says that the
// record before closed one (a scalar register; the walk that ends on it materialises the zero).
template <int G, class Fin>
__device__ __forceinline__ void block_accumulate(ConstPtr<double> fac, typename FactorGroup<G>::type F, uint32_t endmask, uint32_t storemask, int n_valid,
	const v16i &D0, const v16i &D1, double &cell, bool &fresh, const double *tab_s, Fin &&fin)
{
	typedef typename FactorGroup<G>::type FG;
#pragma unroll
	for (int g = 0; g < 32 / G; g++) {
		if (G * g >= n_valid) break;
		double t[G];
#pragma unroll
		for (int q = 0; q < G; q++) {         // D = 8*d: already the byte offset into the table
			const int i = G * g + q;          // record i = 8 m + r  ->  r < 4 ? D0[4 m + r] : D1[4 m + r - 4]
			const int off = (i & 7) < 4 ? D0[4 * (i >> 3) + (i & 3)] : D1[4 * (i >> 3) + (i & 3)];
			t[q] = table_value(tab_s, off);
		}
		// the look-ups are waited for HERE (a use of the first one; LDS returns in order, and a scalar load in flight makes it a
		// wait for everything), and only then are the next group's factors requested: they have this group's arithmetic to arrive.
		// (Requesting the NEXT group's look-ups here as well, before this group is added up, was measured twice -- round 4 and on
		// this loop: +-1 %, eight registers.)
		asm volatile("" : "+v"(t[0]));
		__builtin_amdgcn_sched_barrier(0);
		FG Fn = F;
		if (!ABL_NOFAC && g + 1 < 32 / G) Fn = *(ConstPtr<FG>)(fac + G * (g + 1));
		__builtin_amdgcn_sched_barrier(0);
		if (fresh) { cell = F[0] * t[0]; asm volatile("" : "+v"(cell)); }     // (the asm keeps this a scalar branch, not a select)
		else cell += F[0] * t[0];
		fresh = false;
#pragma unroll
		for (int q = 1; q < G; q += 2) {      // cells are padded to an even number of records: only odd positions close one
			cell += F[q] * t[q];
			const bool end = (endmask & (1u << (G * g + q))) != 0;
			const bool stored = (storemask & (1u << (G * g + q))) != 0;
			if (q + 1 < G) {
				if (end) { fin(cell, stored); cell = F[q + 1] * t[q + 1]; }
				else cell += F[q + 1] * t[q + 1];
			} else if (end) { fin(cell, stored); fresh = true; }
		}
		F = Fn;
	}
}

// (the LDS staging area of round 2 -- the factors parked by lanes 0..31 and read back as broadcasts -- is gone)

// The image part of a haplotype-table entry {image(s), ff, f} through a raw buffer; `vo` = the entry's byte offset + this
// lane's offset into the image.  (ff and f stay in the entry for the per-sample route, hibag_sample.hip; the walks below
// take the product ff * f of a pair from HibagModelView::pfac.)
__device__ __forceinline__ v4i load_hap_image(__amdgpu_buffer_rsrc_t hp, uint32_t vo)
{
	const auto v = __builtin_amdgcn_raw_buffer_load_b128(hp, (int)vo, 0, 0);
	return v4i{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}

// What a walk has already fetched of the list behind its last block: the slot words of the next two
// blocks.  Pass 2 reads one tile's segments classifier after classifier through contiguous memory, so the
// look-ahead of one walk is the prologue of the next.
struct ListCursor {
	uint64_t at = ~(uint64_t)0;      // dword offset of the block `idx` belongs to (~0: nothing fetched)
	uint32_t idx = 0, idx_n = 0;     // this lane's slot word of that block and of the one behind it
};

// a block's header {end-of-cell mask, stored-cell mask, slots worth evaluating, 0} (HibagModelView::phdr)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Walk `nblk` consecutive blocks of a pair list starting at dword offset `at` (a multiple of 32); `cell` = the sum of the
// cell the first record belongs to so far (0 at a cell boundary), on return that of the cell the walk ended in.
//
// Latency plan.  What differs from lane to lane travels as per-lane vector loads, software-pipelined over the blocks:
//   at the top of block b   the haplotype images of block b+1 are gathered (their slot words
//                           arrived during block b-1) and the slot words of block b+2 are requested,
// so that a whole block's evaluation covers their latency.  Lane l (and l+32: the other K half of
// the same row) turns its pair (i1, i2) into the A-operand row (the sum of its two images).
// What is the same for all lanes -- the block's header and the records' frequency factors ff[i1] * f[i2], both made by
// the host (hibag_model.hip finalize_model) -- comes through the scalar cache into scalar registers: the header of block
// b+1 and the first factors of block b are requested at the top of block b, before the matrix instructions; the other
// factors group by group inside block_accumulate (which explains how they avoid the table look-ups' waits).  The number
// of slots worth evaluating follows from the last slot that closes a cell or has a non-zero factor (a zero factor adds
// +0.0: skipping it is exact).  The lists are padded so that every look-ahead stays in bounds.
// PRE (one-step FP4 only): the A-operand rows are PREBUILT (HibagModelView::parow, 1 KB per block): one coalesced 16-byte load
// per lane and block, requested a block ahead right behind the matrix instructions that consumed the current rows -- no
// slot words, no gathers from the haplotype table, no address arithmetic, no additions.
template <int ENG, int G, bool PRE, class Fin>
__device__ __forceinline__ void walk_blocks(const HibagModelView &M, uint64_t at, int nblk, int lane, ListCursor &cur,
	__amdgpu_buffer_rsrc_t hp, int k, const LaneOperand &T, const WideSrc &wide, const double *tab_s, double &cell, Fin &&fin)
{
	static_assert(!PRE || ENG == HIBAG_ENGINE_FP4, "prebuilt rows exist for one-step FP4 classifiers only");
	if (nblk <= 0) return;
	typedef typename FactorGroup<G>::type FG;
	bool fresh = false;                              // block_accumulate: the record before closed a cell
	ConstPtr<double> fac = as_const(M.pfac) + at;                            // this segment's factors and headers
	ConstPtr<u32x4> hdr = (ConstPtr<u32x4>)(as_const(M.phdr) + at / HIBAG_PLIST_DWORDS * 4);
	if (PRE) {
		const uint64_t blk = at / HIBAG_PLIST_DWORDS;
		const uint64_t left = (M.parow_blocks - blk) * 1024u;
		const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(M.parow + blk * 64), 0,
			left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)left, 0x00020000);
		const int vo = lane * 16;
		v4i arow = load_hap_image(pr, (uint32_t)vo);
		u32x4 H_n = hdr[0];
		FG F_n = *(ConstPtr<FG>)fac;
		uint32_t soff = 1024;
		for (int b = 0; b < nblk; b++) {
			const u32x4 H = H_n;
			const FG F = F_n;
			const uint32_t endmask = abl_endmask(H[0]), storemask = abl_storemask(H[1]);
			const int n_valid = (int)H[2];
			// (the header is waited for HERE, before the next scalar loads are issued: a wait behind them would be for them too)
			asm volatile("" :: "s"(n_valid));
			__builtin_amdgcn_sched_barrier(0);
			H_n = hdr[b + 1];
			F_n = *(ConstPtr<FG>)(fac + (size_t)(b + 1) * HIBAG_PLIST_DWORDS);
			v4i a = arow;
			asm volatile("" : "+v"(a));                   // (this block's rows have arrived: requested a block ago)
			if (n_valid > 0) {
				v16i D0, D1;
				block_mfma<ENG>(a, v4i{0, 0, 0, 0}, lane, v4i{0, 0, 0, 0}, T, D0, D1);
				__builtin_amdgcn_sched_barrier(0);
				arow = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(pr, vo, (int)soff, 0));    // the next block's, behind the instructions that read this one's
				__builtin_amdgcn_sched_barrier(0);
				block_own_sample(D0, D1, n_valid);
				block_accumulate<G>(fac + (size_t)b * HIBAG_PLIST_DWORDS, F, endmask, storemask, n_valid, D0, D1, cell, fresh, tab_s, fin);
			} else {
				arow = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(pr, vo, (int)soff, 0));
			}
			soff += 1024;
		}
		if (fresh) cell = 0;
		return;
	}
	// FP4W: an FP4 classifier of `wide.nstep` K steps; `k` = SNPs of its LAST step, the others have HIBAG_FP4_STEP_SNPS
	constexpr bool FP4W = ENG == HIBAG_ENGINE_FP4W;
	const uint32_t ES = FP4W ? 4u * (uint32_t)HIBAG_FP4_ENTRY_DWORDS(wide.nstep)
	                         : 4u * (uint32_t)HIBAG_ENGINE_HAP_DWORDS(ENG);   // bytes per table entry (one-step FP4 and int8: 48)
	const int vo_i = (lane & 31) * 4;                // this lane's slot inside a block
	// this lane's 16 bytes of an entry: the K half's bytes (int8), the K half's nibble image (one-step FP4: the "sum" image
	// for lanes 0..31, the "pair" image for lanes 32..63), the one nibble image (FP4 of several steps)
	const uint32_t img = FP4W ? 0u : (uint32_t)(lane >> 5) * 16u;
	const v4i cterm = FP4W ? fp4_offset_term(HIBAG_FP4_STEP_SNPS, lane) : v4i{0, 0, 0, 0};   // (of K step 0)
	const uint32_t BB = 4 * HIBAG_PLIST_DWORDS;      // bytes per block
	// The list is addressed as a raw buffer rebased at this segment, so that the 32-bit offsets inside
	// the descriptor never limit the model size.
	const uint64_t left = (M.plist_dwords - at) * 4;
	const __amdgpu_buffer_rsrc_t pl = __builtin_amdgcn_make_buffer_rsrc((void *)(M.plist + at), 0,
		left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)left, 0x00020000);
	uint32_t soff = 0;
	if (cur.at != at) {                              // nothing usable fetched: slot words of blocks 0 and 1
		cur.idx = __builtin_amdgcn_raw_buffer_load_b32(pl, vo_i, soff, 0);
		cur.idx_n = __builtin_amdgcn_raw_buffer_load_b32(pl, vo_i, soff + BB, 0);
	}
	uint32_t idx_c = cur.idx, idx_n = cur.idx_n;
	// One address per haplotype: entry * size + this lane's offset into the image(s)
	uint32_t o1 = (idx_c & 0xFFFFu) * ES + img, o2 = ((idx_c >> 16) & 0x3FFFu) * ES + img;
	v4i e1 = load_hap_image(hp, o1), e2 = load_hap_image(hp, o2);
	u32x4 H_n = hdr[0];
	FG F_n = *(ConstPtr<FG>)fac;
	for (int b = 0; b < nblk; b++) {
		// this block's records: header, first factors, images
		const u32x4 H = H_n;
		const FG F = F_n;
		v4i a1 = e1, a2 = e2;
		if (ENG == HIBAG_ENGINE_FP4) {
			// one K step: the A row is made right away, so that the images' registers are free for the next block's loads
			// (otherwise the loop ends in eight register moves)
			a1 = e1 + e2; a2 = v4i{0, 0, 0, 0};
			asm volatile("" : "+v"(a1));
		}
		const uint32_t endmask = abl_endmask(H[0]), storemask = abl_storemask(H[1]);
		const int n_valid = (int)H[2];
		const uint32_t ob1 = o1, ob2 = o2;           // (FP4W: where this block's entries are, for their further images)
		// (the header is waited for HERE, before the next scalar loads are issued: a wait behind them would be for them too)
		asm volatile("" :: "s"(n_valid));
		__builtin_amdgcn_sched_barrier(0);
		// look-ahead: header and first factors of block b+1 (carried around the loop: requested inside the branch below they
		// would be waited for at once); entries of block b+1, slot words of block b+2
		H_n = hdr[b + 1];
		F_n = *(ConstPtr<FG>)(fac + (size_t)(b + 1) * HIBAG_PLIST_DWORDS);
		idx_c = idx_n;
		o1 = (idx_c & 0xFFFFu) * ES + img; o2 = ((idx_c >> 16) & 0x3FFFu) * ES + img;
		e1 = load_hap_image(hp, o1); e2 = load_hap_image(hp, o2);
		idx_n = __builtin_amdgcn_raw_buffer_load_b32(pl, vo_i, soff + 2 * BB, 0);
		if (n_valid > 0) {
			v16i D0, D1;
			if (FP4W) {
				// K step 0 like a one-step classifier, then the further steps: their images and B operands are fetched here
				// (no look-ahead: a classifier this wide is rare, and its registers would be everybody's), chained through the
				// accumulators
				v16f d0, d1;
#pragma unroll
				for (int r = 0; r < 16; r++) { d0[r] = 0.0f; d1[r] = 0.0f; }
				fp4_step(a1, a2, lane, cterm, T.b[0][0], T.b[1][0], d0, d1);
				for (int j = 1; j < wide.nstep; j++) {
					const v4i s1 = load_hap_image(hp, ob1 + 16u + 16u * (uint32_t)j), s2 = load_hap_image(hp, ob2 + 16u + 16u * (uint32_t)j);
					const uint4 *row = wide.bt + ((size_t)(wide.bt_row + 2 * j) * wide.n_group + wide.group) * HIBAG_WAVE;
					const uint4 u0 = row[lane], u1 = row[wide.n_group * HIBAG_WAVE + lane];
					const v4i cj = fp4_offset_term(j == wide.nstep - 1 ? k : HIBAG_FP4_STEP_SNPS, lane);
					fp4_step(s1, s2, lane, cj, v4i{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w}, v4i{(int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w}, d0, d1);
				}
				D0 = __builtin_bit_cast(v16i, d0);
				D1 = __builtin_bit_cast(v16i, d1);
			} else {
				block_mfma<ENG>(a1, a2, lane, cterm, T, D0, D1);
			}
			block_own_sample(D0, D1, n_valid);
			block_accumulate<G>(fac + (size_t)b * HIBAG_PLIST_DWORDS, F, endmask, storemask, n_valid, D0, D1, cell, fresh, tab_s, fin);
		}
		soff += BB;
	}
	if (fresh) cell = 0;
	cur.at = at + (uint64_t)nblk * HIBAG_PLIST_DWORDS;
	cur.idx = idx_c; cur.idx_n = idx_n;
}

// raw-buffer descriptor of a classifier's haplotype table (gfx9 word 3: 32-bit data format, no swizzle;
// reads past the end return 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hap_rsrc(const HibagModelView &M, uint32_t first_dword)
{
	// exact bound: the look-ahead of a walk runs into the next segment's index pairs, whose entries may lie
	// past the end of the table (out-of-range raw-buffer reads return 0 instead of faulting)
	const uint64_t left = (uint64_t)(M.hap_dwords - first_dword) * 4u;
	return __builtin_amdgcn_make_buffer_rsrc((void *)(M.hap + first_dword), 0, left > 0x7FFFFFF0ull ? 0x7FFFFFF0 : (int)left, 0x00020000);
}

// matrix-engine variant of a classifier -> template instance
#define HIBAG_DISPATCH_ENGINE(code, CALL)              \
	switch (code) {                                    \
	case HIBAG_ENGINE_FP4: { CALL(HIBAG_ENGINE_FP4); } break;  \
	case HIBAG_ENGINE_I8:  { CALL(HIBAG_ENGINE_I8); } break;   \
	default:               { CALL(HIBAG_ENGINE_I8S); } break;  \
	}
// ... where FP4 classifiers of several K steps can turn up (k_total_wide, k_vote_best: the hot kernels never see them --
// their extra registers would cost every classifier a spill in the block loop)
#define HIBAG_DISPATCH_ENGINE_WIDE(code, nstep, CALL)  \
	switch (code) {                                    \
	case HIBAG_ENGINE_FP4: if ((nstep) > 1) { CALL(HIBAG_ENGINE_FP4W); } else { CALL(HIBAG_ENGINE_FP4); } break;  \
	case HIBAG_ENGINE_I8:  { CALL(HIBAG_ENGINE_I8); } break;   \
	default:               { CALL(HIBAG_ENGINE_I8S); } break;  \
	}

// Record widths the kernels are specialised for; the host rounds a classifier's
// ceil(3k/32) up to the next of these (padding words carry AND mask 0).
#define HIBAG_DISPATCH_NWP(nwp, CALL)      \
	switch (nwp) {                         \
	case 1:  { CALL(1); } break;           \
	case 2:  { CALL(2); } break;           \
	case 3:  { CALL(3); } break;           \
	case 4:  { CALL(4); } break;           \
	case 6:  { CALL(6); } break;           \
	case 8:  { CALL(8); } break;           \
	case 10: { CALL(10); } break;          \
	default: { CALL(12); } break;          \
	}

__device__ __forceinline__ void stage_table(const HibagModelView &M, double *tab_s, int n = HIBAG_TAB_N)
{
	for (int i = threadIdx.x; i < n; i += blockDim.x) tab_s[i] = M.tab[i];
	__syncthreads();
}

// ---------------------------------------------------------------------------
// k_codes: the raw genotype matrix int32 [n_samp][row_len] (sample-major, the
// memory of R's SNP x sample matrix) -> byte codes [n_snp][n_pad] with the
// sample index fastest: 0/1/2 = genotype, 3 = missing (anything outside 0..2,
// incl. NA_integer_, src/LibHLA.cpp:662-665).  64x64 transpose through LDS:
// reads are coalesced along SNPs, writes along samples.
// With `col` the matrix is the cohort's own (row_len SNPs in the cohort's order):
// model SNP k is read from column col[k] (-1 = the cohort lacks it -> missing) and
// flip[k] != 0 reverses its allele count, g -> 2 - g: the SNP selection and strand /
// allele-order fix-up of hlaPredict (R/HIBAG.R:640-676) done while packing instead
// of on the host.  col == nullptr: the matrix already is in model order (row_len = n_snp).
__global__ __launch_bounds__(256) void k_codes(HibagModelView M, HibagBatchView B,
	const int32_t *__restrict__ geno, int row_len, const int32_t *__restrict__ col, const int32_t *__restrict__ flip,
	uint8_t *__restrict__ codes)
{
	__shared__ uint8_t tile[64][65];
	const int s0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int k = k0 + tx;
	int c = -1, f = 0;
	if (k < M.n_snp) {
		c = col ? col[k] : k;
		f = (col && flip) ? flip[k] : 0;
	}
	for (int r = ty; r < 64; r += 4) {
		const int s = s0 + r;
		int g = -1;
		if (s < B.n_samp && c >= 0) g = geno[(size_t)s * row_len + c];
		uint8_t v = (g >= 0 && g <= 2) ? (uint8_t)g : (uint8_t)3;
		if (f && v != 3) v = (uint8_t)(2 - v);
		tile[r][tx] = v;
	}
	__syncthreads();
	for (int r = ty; r < 64; r += 4) {
		const int kk = k0 + r;
		if (kk < M.n_snp) codes[(size_t)kk * B.n_pad + s0 + tx] = tile[tx][r];
	}
}

// ---------------------------------------------------------------------------
// PLINK BED sources (HIBAG_ConvBED, src/HIBAG.cpp:1094-1191).  `bed` is the
// payload after the 3-byte prefix: rows of `stride` bytes, 4 two-bit codes per
// byte, lowest bits first.  mode 0 = individual-major (row = sample, column =
// SNP), otherwise SNP-major.  Code -> genotype {2, NA, 1, 0} (:1135), returned
// here as the byte code 0/1/2 or 3 = missing.
__device__ __forceinline__ uint32_t bed_code(const uint8_t *__restrict__ bed, int mode, size_t stride, int snp_row, int samp)
{
	const size_t row = mode == 0 ? (size_t)samp : (size_t)snp_row;
	const int col = mode == 0 ? snp_row : samp;
	const uint32_t two = ((uint32_t)bed[row * stride + (size_t)(col >> 2)] >> (2 * (col & 3))) & 3u;
	return (0x0132u >> (4 * two)) & 0xFu;
}

// k_bed_codes: BED payload -> the byte codes [n_snp][n_pad] k_pack consumes,
// skipping the int32 matrix.  snp_row[k] = row (SNP-major) / column
// (individual-major) of model SNP k inside `bed`, or -1 if the cohort lacks it
// (-> missing); flip[k] != 0 swaps the allele count, g -> 2 - g (the strand /
// allele-order fix-up of hlaPredict, R/HIBAG.R:640-676).  Block = 4 wavefronts,
// one SNP each, lane = sample: SNP-major rows are read as 16 contiguous bytes
// per wavefront and written as 64 contiguous codes.
__global__ __launch_bounds__(256) void k_bed_codes(HibagModelView M, HibagBatchView B,
	const uint8_t *__restrict__ bed, int mode, size_t stride, int samp0,
	const int32_t *__restrict__ snp_row, const int32_t *__restrict__ flip, uint8_t *__restrict__ codes)
{
	const int k = blockIdx.y * 4 + (threadIdx.x >> 6);
	const int s = blockIdx.x * 64 + (threadIdx.x & 63);
	if (k >= M.n_snp) return;
	uint32_t g = 3;
	const int r = snp_row[k];
	if (s < B.n_samp && r >= 0) {
		g = bed_code(bed, mode, stride, r, samp0 + s);
		if (flip[k] && g != 3) g = 2 - g;
	}
	codes[(size_t)k * B.n_pad + s] = (uint8_t)g;
}

// k_bed_geno: HIBAG_ConvBED itself -- the int32 matrix [n_samp][n_save]
// (sample-major = R's n_save x n_samp matrix) of the selected SNPs, NA_integer_
// for the missing code.  64 x 64 tiles; SNP-major sources go through an LDS
// transpose so that both the byte reads (along samples) and the int32 writes
// (along SNPs) are contiguous.
__global__ __launch_bounds__(256) void k_bed_geno(const uint8_t *__restrict__ bed, int mode, size_t stride,
	int n_samp, int n_save, const int32_t *__restrict__ sel, int32_t *__restrict__ geno)
{
	__shared__ uint8_t tile[64][65];
	const int s0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	if (mode != 0) {
		for (int r = ty; r < 64; r += 4) {           // r = SNP, tx = sample
			const int j = j0 + r, s = s0 + tx;
			tile[r][tx] = (j < n_save && s < n_samp) ? (uint8_t)bed_code(bed, mode, stride, sel[j], s) : (uint8_t)3;
		}
		__syncthreads();
	}
	for (int r = ty; r < 64; r += 4) {               // r = sample, tx = SNP
		const int s = s0 + r, j = j0 + tx;
		if (s >= n_samp || j >= n_save) continue;
		const uint32_t g = mode != 0 ? tile[tx][r] : bed_code(bed, mode, stride, sel[j], s);
		geno[(size_t)s * n_save + j] = g == 3 ? (int32_t)0x80000000 : (int32_t)g;
	}
}

// k_pack: TGenotype::IntToSNP (src/LibHLA.cpp:662-706) for every (sample, classifier), plus the
// classifier weight from missingness (src/LibHLA.cpp:2418-2431).  grid (n_pad/64, C / 4), one
// wavefront per classifier, lane = sample: every code load is one coalesced 64-byte row segment.
// Matrix-engine classifiers (at most 112 SNPs) get the sample's column of the B operand (int8 bytes or FP4 nibbles)
// in the K layout of hibag_device.h, written to the two lanes (K halves) that own it in the MFMA layout.
// VALU-engine classifiers get the lane masks of the packed 3k-bit pair string
//   bits [0,k)   first haplotype : x = [g==2], m = [g in {0,2}]
//   bits [k,2k)  second haplotype: same
//   bits [2k,3k) ~(H1^H2)        : x = 0,      m = [g==1]
// (missing SNPs have m = 0 everywhere).
#define PACK_WAVES 4        // classifiers per workgroup (one wavefront each)
__global__ __launch_bounds__(PACK_WAVES * HIBAG_WAVE) void k_pack(HibagModelView M, HibagBatchView B,
	const uint8_t *__restrict__ codes)
{
	__shared__ uint32_t pack_s[PACK_WAVES][3][4][HIBAG_WAVE];
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) B.err_dev[2] = 0;     // the batch's list of totals without a finite reciprocal (pass 1 -> k_nan_cells)
	const int c = blockIdx.y * PACK_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (c >= M.n_classifier) return;
	const int lane = threadIdx.x & 63;
	const int s = blockIdx.x * HIBAG_WAVE + lane;
	const int k = M.n_snp_c[c];
	const int nwp = M.nwp[c];
	const int *__restrict__ idx = M.snp_index + M.snp_off[c];
	const int row0 = M.mask_row[c];
	const int nkb = M.engine[c];                      // matrix-engine variant, 0 = VALU engine
	int num = 0, den = 0;
	if (nkb > 0) {
	// one pass over the SNPs of each K step (independent byte loads, several in flight): all k <= 32 of them, or 28 per
	// step of a multi-step FP4 classifier
	const int steps = M.n_step[c];
	for (int st = 0; st < steps; st++) {
		const int j0 = steps > 1 ? HIBAG_FP4_STEP_SNPS * st : 0, kj = steps > 1 ? min(HIBAG_FP4_STEP_SNPS, k - j0) : k;
		uint32_t X = 0, Z = 0, E = 0;          // bit j: g == 2, g == 0, g == 1 at SNP j0 + j
#pragma unroll 8
		for (int j = 0; j < kj; j++) {
			const int snp = idx[j0 + j];
			const uint32_t g = codes[(size_t)snp * B.n_pad + s];
			const int wt = M.snp_weight[snp];
			den += wt;
			if (g != 3) num += wt;
			X |= (uint32_t)(g == 2) << j;
			Z |= (uint32_t)(g == 0) << j;
			E |= (uint32_t)(g == 1) << j;
		}
		const uint32_t offset = 2u * (uint32_t)__popc(X) + (uint32_t)__popc(E);     // <= 64
		const int n = lane >> 5;
		if (nkb == HIBAG_ENGINE_FP4) {
			// e2m1 codes: +1 -> 0x2, -1 -> 0xA, 2 -> 0x4, 3 -> 0x5, 4 -> 0x6.  K half 0 (positions 0..31): the signs of the SNPs,
			// then the offset's two low base-4 digits at k, k+1; K half 1 (positions 32..63): [g == 1] of the SNPs, then offset bits 4, 5.
			const uint32_t neg = X | E;
#pragma unroll
			for (int h = 0; h < 2; h++) {
				uint32_t a[4];
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t z8 = (Z >> (8 * q)) & 0xFFu, n8 = (neg >> (8 * q)) & 0xFFu, e8 = (E >> (8 * q)) & 0xFFu;
					// (one K step: the upper half of A is w = 0 / 1.5 / 4, not the AND, so g = 1 counts -1 - 3 = -4 = code 0xE here)
					a[q] = h == 0 ? (expand_bits8_nibbles(z8 | n8) << 1) | (expand_bits8_nibbles(n8) << 3) | (steps == 1 ? expand_bits8_nibbles(e8) << 2 : 0u)
					              : expand_bits8_nibbles(e8) << 1;
				}
				// the offset (<= 60) in four digits: (offset & 3) and ((offset >> 2) & 3) as the values 0 / 1 / 2 / 3 (e2m1 codes
				// 0, 2, 4, 5) against A = 1 and A = 4 in the lower K half; bit 4 as the value 2 (code 4) and bit 5 as the value 4
				// (code 6), both against A = 4, in the upper half, which counts twice: 4 * 2 * 2 = 16, 4 * 4 * 2 = 32
				const uint32_t code4 = 0x5420u;             // value v -> e2m1 code
				const uint32_t digits = h == 0 ? ((code4 >> (4 * (offset & 3u))) & 0xFu) | (((code4 >> (4 * ((offset >> 2) & 3u))) & 0xFu) << 4)
				                               : (((offset >> 4) & 1u) * 0x4u) | (((offset >> 5) & 1u) * 0x60u);
				const unsigned __int128 d128 = (unsigned __int128)digits << (4 * kj);
#pragma unroll
				for (int q = 0; q < 4; q++) a[q] |= (uint32_t)(d128 >> (32 * q));
				B.bt[((size_t)(M.bt_row[c] + 2 * st + n) * gridDim.x + blockIdx.x) * HIBAG_WAVE + h * 32 + (lane & 31)] =
					uint4{a[0], a[1], a[2], a[3]};
			}
		} else {
			const uint64_t pos64 = Z, neg64 = X | E, e64 = (uint64_t)E << 32;
#pragma unroll
			for (int m = 0; m < 2; m++) {
				const uint32_t pw = (uint32_t)(pos64 >> (32 * m)), nw = (uint32_t)(neg64 >> (32 * m)), ew = (uint32_t)(e64 >> (32 * m));
#pragma unroll
				for (int h = 0; h < 2; h++) {
					const v4i pos = expand_bits16((pw >> (16 * h)) & 0xFFFFu), neg = expand_bits16((nw >> (16 * h)) & 0xFFFFu),
						one = expand_bits16((ew >> (16 * h)) & 0xFFFFu);
					uint32_t a[4];
#pragma unroll
					for (int q = 0; q < 4; q++) a[q] = (uint32_t)pos[q] * 0x08u | (uint32_t)neg[q] * 0xF8u | (uint32_t)one[q] * 0x10u;
					if (m == 0 && h == 1 && k < 32) a[3] |= offset << 24;       // K position 31 meets the A operand's 8
					B.bt[((size_t)(M.bt_row[c] + n * 2 + m) * gridDim.x + blockIdx.x) * HIBAG_WAVE + h * 32 + (lane & 31)] =
						uint4{a[0], a[1], a[2], a[3]};
				}
			}
			if (k == 32) {                                                      // no K position left: the offset starts the accumulators
				const size_t at = ((size_t)(2 * c + n) * gridDim.x + blockIdx.x) * HIBAG_WAVE + (lane & 31);
				B.bias[at] = 8 * (int)offset;
				B.bias[at + 32] = 8 * (int)offset;
			}
		}
	}
	} else {
		// VALU engine (more than 112 SNPs): one pass over the k <= 128 SNPs builds the
		// three k-bit fields [g == 2], [g in {0, 2}], [g == 1] in LDS (four words each per lane); the 3k-bit
		// strings are then put together word by word with wave-uniform bit offsets.
		uint32_t (*fld)[4][HIBAG_WAVE] = pack_s[threadIdx.x >> 6];          // [field][word][lane]
#pragma unroll
		for (int w = 0; w < 4; w++) {
			uint32_t X = 0, Mv = 0, E = 0;
			const int j0 = 32 * w, j1 = min(k, j0 + 32);
#pragma unroll 8
			for (int j = j0; j < j1; j++) {
				const int snp = idx[j];
				const uint32_t g = codes[(size_t)snp * B.n_pad + s];
				const int wt = M.snp_weight[snp];
				den += wt;
				if (g != 3) num += wt;
				X |= (uint32_t)(g == 2) << (j - j0);
				Mv |= (uint32_t)(g == 0 || g == 2) << (j - j0);
				E |= (uint32_t)(g == 1) << (j - j0);
			}
			fld[0][w][lane] = X; fld[1][w][lane] = Mv; fld[2][w][lane] = E;
		}
		// 32 bits of field f starting at bit `off` (bits outside [0, 128) are zero); off is wave-uniform
		auto bits_at = [&](int f, int off) -> uint32_t {
			const int w0 = off >> 5, sh = off & 31;
			const uint32_t v0 = (w0 >= 0 && w0 < 4) ? fld[f][w0][lane] : 0u;
			const uint32_t v1 = (w0 + 1 >= 0 && w0 + 1 < 4) ? fld[f][w0 + 1][lane] : 0u;
			return sh ? (v0 >> sh) | (v1 << (32 - sh)) : v0;
		};
		for (int m = 0; m < nwp; m++) {
			const uint32_t xw = bits_at(0, 32 * m) | bits_at(0, 32 * m - k);
			const uint32_t mw = bits_at(1, 32 * m) | bits_at(1, 32 * m - k) | bits_at(2, 32 * m - 2 * k);
			B.masks[(size_t)(row0 + m) * B.n_pad + s] = xw;
			B.masks[(size_t)(row0 + nwp + m) * B.n_pad + s] = mw;
		}
	}
	const double cw = (s < B.n_samp && den > 0) ? ((double)num / den) : 0.0;
	B.cw[(size_t)c * B.n_pad + s] = cw;
	B.winv[2 * ((size_t)c * B.n_pad + s)] = cw;       // (and beside it, once pass 1 has it, 1/total: what pass 2 reads per block in one load)
}

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
// multiple of 8 apart), so the hand-over goes through that XCD's L2: the parked sums are plain stores, complete in L2
// once the storing wavefront's vmcnt is 0 (the vector L1 writes through); the flag follows behind a workgroup barrier
// as an L1-bypassing (sc1) store; the reader polls it with sc1 loads and fetches the sums with sc1 loads, which
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

// a parked sum: read past the CU's L1
__device__ __forceinline__ double load_parked(const double *p)
{
	return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
template <bool STORE, int OCC, bool FP4ONLY, bool VOTE = false>
__global__ __launch_bounds__(BLOCK_THREADS, OCC) void k_total(HibagModelView M, HibagBatchView B, int gx, int n_whole, int rest, int stride, int K)
{
	static_assert(!(STORE && VOTE), "the majority vote has no second pass to store cell sums for");
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
			auto fin = [&](double v, bool stored) {
#ifdef HIBAG_STORE_PLAIN      // (variant: write-back stores instead of streaming ones)
				if (STORE && stored) { rows[(size_t)row * HIBAG_WAVE + lane] = v; row++; }
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
#define CALLX(E, PRE) { LaneOperand T; load_operand_row<E>(B, M.bt_row[c], c, group, lane, T);                                 \
			ListCursor cur;                                                                                                \
			walk_blocks<E, TOTAL_G, PRE>(M, M.blk_off[c] + (uint64_t)b0 * HIBAG_PLIST_DWORDS, b1 - b0, lane, cur,            \
				hap_rsrc(M, M.hap_off[c]), M.n_snp_c[c], T, WideSrc(), tab_s, cell, fin); }
#define CALL(E) CALLX(E, false)
			// one-step FP4 classifiers of a model small enough for prebuilt A-operand rows walk those (HibagModelView::parow)
			if (FP4ONLY || nkb == HIBAG_ENGINE_FP4) { if (M.p1_prebuilt) CALLX(HIBAG_ENGINE_FP4, true) else CALLX(HIBAG_ENGINE_FP4, false) }
			else if (nkb == HIBAG_ENGINE_I8) CALL(HIBAG_ENGINE_I8)
			else CALL(HIBAG_ENGINE_I8S)
#undef CALL
#undef CALLX
			if (!last) { B.tot[at] = total; B.inv[at] = cell; }
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
			B.winv[2 * at + 1] = inv;                     // (beside the weight k_pack left there: pass 2 reads both in one load)
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
template <bool WHOLE>
__global__ __launch_bounds__(BLOCK_THREADS, 4) void k_total_wide(HibagModelView M, HibagBatchView B)
{
	__shared__ double tab_s[HIBAG_TAB_N];
	stage_table(M, tab_s);
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
	auto fin = [&](double v, bool) {
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
		B.winv[2 * at + 1] = inv;
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
	B.winv[2 * ((size_t)c * B.n_pad + s) + 1] = 1 / total;
	note_infinite_reciprocal(B, c, s, B.cw[(size_t)c * B.n_pad + s], 1 / total);
}

// The per-sample ensemble scalars, classifiers in order (k_scalars, or the tile-0 workgroups of k_accum):
//   part[P]   = sum of weights   (_Sum_Weight, src/LibHLA.cpp:1505; for the
//               majority vote the number of classifiers that produced a call)
//   part[P+1] = sum_matching = sum_c total_c * w_c        (:2458)
//   part[P+2] = num_matching = sum_c w_c                  (:2459)
__device__ __forceinline__ void ensemble_scalars(const HibagModelView &M, const HibagBatchView &B, int s, const int *__restrict__ best_cell)
{
	double sum_w = 0, sum_m = 0, num_m = 0;
	constexpr int NB = 16;
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

// ---------------------------------------------------------------------------
// k_accum (pass 2): for one tile of allele-pair cells and 64 samples, go through the classifiers in order and do
//     S[p] += (cell * (1/total)) * w          (src/LibHLA.cpp:1828 then :1497-1507)
// with the tile's S in LDS (one row of 64 doubles per cell, conflict-free).  What a classifier contributes to a tile is
// either evaluated again from its haplotype pairs (the cells with few pairs) or read back from the sums pass 1 stored
// (HibagModelView::store_cells); both arrive here as ONE STREAM OF BLOCKS per tile (hibag_device.h, "E-stream"): the
// blocks of classifier 0, 1, 2 ... that have anything for the tile, each block 32 pair slots plus a 32-byte header that
// names the block's classifier (-> weight and 1/total rows), its operand row and haplotype table, the tile rows its
// cells close into and up to eight stored sums to add.  Round 2 walked (classifier, tile) "visits" -- mostly one short
// block each -- with a scalar prologue per visit (record, descriptors, engine dispatch) and nothing of the next visit in
// flight while the current one ran: 0.73 us of SIMD time per visit against 0.35 us of instructions.  As a stream the loop
// body is one block, and at its top EVERYTHING of block b + 1 is requested -- haplotype entries, B operand, weight,
// 1/total, stored sums -- plus the slot words and header of block b + 2, so a whole block's evaluation covers each
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

// What a block needs that is requested a block ahead and is still in use while the NEXT block's is in flight: its header,
// the end-of-cell masks, its first factors (scalar registers) and the lane's weight and 1/total.  The loop body exists twice
// (A -> B, B -> A): the two sets take turns, nothing is moved from a "next" register to a "current" one.
struct AccumAhead {
	u32x8 hv;           // the E-stream header (hibag_device.h)
	u32x4 ph;           // {end-of-cell mask, -, slots worth evaluating, -}
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
		const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(M.parow + blk0 * 64), 0,
			bytes32((size_t)(M.parow_blocks - blk0) * 1024u), 0x00020000);
		ConstPtr<double> fac = as_const(M.pfac) + blk0 * HIBAG_PLIST_DWORDS;                 // the slots' frequency factors
		ConstPtr<u32x4> phdr = (ConstPtr<u32x4>)(as_const(M.phdr) + blk0 * 4);               // the blocks' {end mask, -, slots worth evaluating, -}
		ConstPtr<u32x8> eh = (ConstPtr<u32x8>)(as_const(M.ehdr) + blk0 * 8);                 // the blocks' 8-dword headers (scalar loads)
		typedef FactorGroup<ACCUM_G>::type AFG;
		// the batch's operand / {weight, 1/total} rows and this group's stored sums as raw buffers too: a row is then a scalar
		// offset (classifier or row number times the row size, SALU) added to one constant per-lane offset -- no 64-bit address
		// arithmetic on the vector ALU.  (hibag_predict.hip batch_limit keeps every one of these arrays below 4 GB.)
		// (every descriptor ends where its array ends: a request past it -- a look-ahead through a header that names more than
		// exists -- reads zeros instead of faulting)
		const __amdgpu_buffer_rsrc_t r_bt = __builtin_amdgcn_make_buffer_rsrc((void *)B.bt, 0, bytes32((size_t)B.bt_rows * B.n_pad * 16u), 0x00020000);
		const __amdgpu_buffer_rsrc_t r_wi = __builtin_amdgcn_make_buffer_rsrc((void *)B.winv, 0, bytes32((size_t)C * B.n_pad * 16u), 0x00020000);
		const __amdgpu_buffer_rsrc_t r_sv = __builtin_amdgcn_make_buffer_rsrc(
			(void *)(B.cells + (size_t)group * (size_t)as_const(M.cell_row)[C] * HIBAG_WAVE), 0,
			bytes32((size_t)as_const(M.cell_row)[C] * HIBAG_WAVE * 8u), 0x00020000);
		const int vo_a = lane * 16, vo_row = (group * HIBAG_WAVE + lane) * 16, vo_sv = lane * 8;
		const uint32_t row_stride = (uint32_t)B.n_pad * 16u;          // bytes per operand row and per classifier's {weight, 1/total} row
		constexpr int NS = HIBAG_STORED_PER_VISIT;

		// per-lane data of the block in hand, requested a block ahead into the registers the block before has just finished with
		v4i arow, t0, t1;                             // the A-operand row, the B operand (two sample halves)
		double sv[NS];                                // the stored sums
		double cell = 0;
		bool fresh = false;                           // block_accumulate: the record before closed a cell
		// the stored sums of a block (word 1 of its header: first row | count << 25).  Their number differs from block to block,
		// so a wait that leaves them in flight would have to be a counted one the compiler cannot get right; ALWAYS requesting
		// HIBAG_STORED_PER_VISIT of them (the ones a block lacks out of the buffer's range: no memory access) so that every wait
		// is exact, and adding them at the end of the block, was measured: pass 2 +15 % -- the loads that fetch nothing still cost
		// their issue (profiles/r05_pass2_notes.txt).
		auto request_sv = [&](uint32_t w1) {
			const int ns = abl2_stored(w1);
			if (ns > 0) {
				const int sr = (int)(abl2_stored_row(w1) * (uint32_t)(HIBAG_WAVE * 8));
#pragma unroll
				for (int i = 0; i < NS; i++) {
					if (i >= ns) break;
					sv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_sv, vo_sv, sr + i * HIBAG_WAVE * 8, 2));   // (read once: nt)
				}
			}
		};
		// words 0, 1 of a header (its own, or -- words 2, 3 -- the next block's): classifier | operand row << 16, stored row | stored sums << 25
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
			const double w_c = cur.winv[0];
			const bool active = w_c > 0;
			// (as integers in scalar registers: a bool that lives across the requests below ends up in a vector register and back)
			const int any = __builtin_amdgcn_readfirstlane(__ballot(active) != 0 ? 1 : 0);   // 0: nobody in the group uses the classifier (src/LibHLA.cpp:2451): nothing to add
			// inactive lanes (weight 0) must keep their sums: with 1/total replaced by 0 their term is
			// (cell * 0) * 0 = +0 and a + 0 == a, which spares a select per closed cell
			const double inv_e = active ? cur.winv[1] : 0.0;
			const uint32_t endmask = cur.ph[0];
			const int n_valid = (int)cur.ph[2];
			asm volatile("" :: "s"(n_valid));             // (this block's scalar data is waited for before the next block's is requested)
			__builtin_amdgcn_sched_barrier(0);
			nxt.hv = eh[rel + 1];
			nxt.ph = phdr[rel + 1];
			nxt.F = *(ConstPtr<AFG>)(fac + (size_t)(rel + 1) * HIBAG_PLIST_DWORDS);
			// The other three 64-byte lines of block b + 1's factors are touched a block ahead, so that the scalar loads of its
			// later groups hit the scalar cache (-5 % on the kernel): one dword each, volatile so that the loads stay HERE, "used"
			// at the end of this block (the compiler waits for them there, where they are long done).
			typedef const volatile __attribute__((address_space(4))) uint32_t *TouchPtr;
			const TouchPtr touch = (TouchPtr)(uintptr_t)(fac + (size_t)(rel + 1) * HIBAG_PLIST_DWORDS);
			const uint32_t tch0 = touch[16], tch1 = touch[32], tch2 = touch[48];
			ACCUM_STAMP(0);
			const int eval = ABL2_NOEVAL ? 0 : __builtin_amdgcn_readfirstlane(any & (n_valid > 0 ? 1 : 0));
			// ---- the sums pass 1 stored for this block's classifier:   S[p] += (cell * (1/total)) * w
			{
				const int ns = abl2_stored(cur.hv[1]);
				if (any && ns > 0) {
					uint32_t jps = cur.hv[6];
#pragma unroll
					for (int i = 0; i < NS; i++) {
						if (i >= ns) break;
						__hip_atomic_fetch_add(&acc[(int)(jps & 15)][lane], (sv[i] * inv_e) * w_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						jps >>= 4;
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
			// ---- every lane its own sample's distances, then cell += prod * TAB[d] in order
			if (eval) {
				block_own_sample(D0, D1, n_valid);
#ifdef HIBAG_ACCUM_STAMPS
				asm volatile("" :: "v"(D0[0]), "v"(D1[0]));
				ACCUM_STAMP(4);
#endif
				uint64_t jpack = ((uint64_t)cur.hv[5] << 32) | cur.hv[4];
				// S[p] += v as one LDS floating-point add (ds_add_f64: the same IEEE addition, no register for the old
				// sum, nothing to wait for)
				auto fin = [&](double c, bool) {
					const double v = (c * inv_e) * w_c;
					__hip_atomic_fetch_add(&acc[(int)(jpack & 15)][lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					jpack >>= 4;
				};
				block_accumulate<ACCUM_G>(fac + (size_t)rel * HIBAG_PLIST_DWORDS, cur.F, endmask, 0u, n_valid, D0, D1, cell, fresh, tab_s, fin);
			}
			asm volatile("" :: "s"(tch0), "s"(tch1), "s"(tch2));
			ACCUM_STAMP(5);
		};

		AccumAhead A, Bn;
		A.hv = eh[0];
		A.ph = phdr[0];
		A.F = *(ConstPtr<AFG>)fac;
		request_lane(A.hv[0], A.hv[1], 0, A.winv);
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
	for (int j = 0; j < ncell; j++) B.part[(size_t)(p0 + j) * B.n_pad + s] = acc[j][lane];
	// The workgroup that ends tile 0 of its sample groups also forms their ensemble scalars (k_scalars' loop, classifiers in
	// order): one kernel and its launch gap less on the step.
	if (tile == 0 && ce == C) ensemble_scalars(M, B, s, nullptr);
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
		tile[r][tx] = (p < P) ? (sum_w != sum_w ? sum_w : normalised(part[(size_t)p * B.n_pad + s0 + tx], scale, ff)) : 0.0;   // (NaN weight sum: poisoned batch)
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

void hibag_launch_pack(const HibagModelView &M, const HibagBatchView &B, const int32_t *d_geno, int row_len,
	const int32_t *d_col, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st)
{
	if (M.n_classifier == 0 || M.n_snp == 0) return;
	hipLaunchKernelGGL(k_codes, dim3(B.n_pad / 64, (M.n_snp + 63) / 64), dim3(256), 0, st, M, B, d_geno,
		d_col ? row_len : M.n_snp, d_col, d_flip, d_codes);
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
static int resident_blocks(F kernel, int threads)
{
	int per_cu = 0, cus = 0, dev = 0;
	(void)hipGetDevice(&dev);
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess) return 0;
	return per_cu > 0 && cus > 0 ? per_cu * cus : 0;
}

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
	total[0] = resident_blocks(k_total<false, HIBAG_TOT_OCC, false>, BLOCK_THREADS);
	total[1] = resident_blocks(k_total<false, HIBAG_TOT_OCC_MANY, true>, BLOCK_THREADS);
	total[2] = resident_blocks(k_total<true, HIBAG_TOT_OCC, false>, BLOCK_THREADS);
	total[3] = resident_blocks(k_total<true, HIBAG_TOT_OCC_MANY, true>, BLOCK_THREADS);
	*accum = resident_blocks(k_accum, ACCUM_WAVES * HIBAG_WAVE);
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
	if (vote) {
		// majority vote: no second pass, so no cell sums are stored for one; the walk logs its records instead
		if (many) hipLaunchKernelGGL((k_total<false, HIBAG_TOT_OCC_MANY, true, true>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K);
		else hipLaunchKernelGGL((k_total<false, HIBAG_TOT_OCC, false, true>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K);
	} else if (M.store_cells) {
		if (many) hipLaunchKernelGGL((k_total<true, HIBAG_TOT_OCC_MANY, true>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K);
		else hipLaunchKernelGGL((k_total<true, HIBAG_TOT_OCC, false>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K);
	} else {
		if (many) hipLaunchKernelGGL((k_total<false, HIBAG_TOT_OCC_MANY, true>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K);
		else hipLaunchKernelGGL((k_total<false, HIBAG_TOT_OCC, false>), grid, dim3(BLOCK_THREADS), 0, st, V, B, (int)gx, (int)n_whole, (int)rest, (int)stride, (int)K);
	}
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
	hipLaunchKernelGGL(k_accum, dim3(8 * (n_whole + K * (nx - n_whole))), dim3(ACCUM_WAVES * HIBAG_WAVE), 0, st, M, B, (int)n_whole, (int)K);
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
