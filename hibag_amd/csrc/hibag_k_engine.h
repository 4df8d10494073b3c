// hibag_k_engine.h -- part of hibag_kernels.hip (included there, one translation unit: the walks are templates that inline into
// their kernels): the two engines of the pair walk: the lane's genotype masks and the popcount form of the distance (VALU engine), the matrix instructions, lane swaps, in-order accumulation and the block walk (matrix engine).
#ifndef HIBAG_K_ENGINE_H_
#define HIBAG_K_ENGINE_H_

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

// One-step FP4, the walk without lane swaps (walk_blocks, OWN): b[0][0] = the "sum" side (K half 0 of the packed operand) and
// b[1][0] = the "pair" side (K half 1) of the LANE'S OWN sample -- lanes 0..31 take theirs from the operand row of samples
// 0..31, lanes 32..63 from the row of samples 32..63 (k_pack's layout: row = sample half, position = K half * 32 + sample).
__device__ __forceinline__ void load_operand_own_sample(const HibagBatchView &B, int bt_row, int group, int lane, LaneOperand &T)
{
	const size_t n_group = (size_t)(B.n_pad / HIBAG_WAVE);
	const uint4 *row = B.bt + ((size_t)(bt_row + (lane >> 5)) * n_group + group) * HIBAG_WAVE + (lane & 31);
	const uint4 va = row[0], vb = row[32];
	T.b[0][0] = v4i{(int)va.x, (int)va.y, (int)va.z, (int)va.w};
	T.b[1][0] = v4i{(int)vb.x, (int)vb.y, (int)vb.z, (int)vb.w};
	T.b[0][1] = T.b[1][1] = v4i{0, 0, 0, 0};
	T.bias[0] = T.bias[1] = 0;
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
template <class Live8>
__device__ __forceinline__ void block_own_sample(v16i &D0, v16i &D1, Live8 &&live8)    // in place: (acc0, acc1) -> (D0, D1); live8(g): records 8 g .. 8 g + 7 have valid ones
{
#pragma unroll
	for (int g = 0; g < 4; g++) {
		if (!live8(g)) break;
		if (ABL_NOSWAP) continue;
#pragma unroll
		for (int r = 4 * g; r < 4 * g + 4; r++) {
			const auto sw = __builtin_amdgcn_permlane32_swap(D0[r], D1[r], false, false);
			D0[r] = sw[0]; D1[r] = sw[1];
		}
	}
}
__device__ __forceinline__ void block_own_sample(v16i &D0, v16i &D1, int n_valid)
{
	block_own_sample(D0, D1, [&](int g) { return 8 * g < n_valid; });
}

// A bank-interleaved table for classifiers of several K steps (33 .. 112 SNPs) -- BUILT, MEASURED SLOWER, NOT SHIPPED
// (-DHIBAG_WIDE_TAB=true builds it).  Their distances spread over 0 .. 2 k, so the 64 lanes of a look-up hit 64 different
// 8-byte entries of the plain table: k_total_wide counts more bank-conflict cycles than LDS-busy cycles
// (profiles/r05_sq_counters.txt: SQ_LDS_BANK_CONFLICT 976,690 against SQ_ACTIVE_INST_LDS 837,640).  TAB[d] is exactly zero
// from d = 65 on (exp(d log 1e-5) underflows: src/LibHLA.cpp:176-183; checked where a model is created), so the offset can be
// clamped at 65 and the 66 entries laid out 32-way interleaved -- entry d of replica r at (32 d + r) * 8 bytes, lane l reads
// replica l % 32: whatever d, the 32 lanes of a ds_read_b64 group are on 32 different bank pairs; 17 KB per workgroup.
// It removes the conflicts and costs two vector instructions per pair (the clamp, the shift-and-add) where the plain table
// costs none -- the matrix result IS its byte offset -- on a loop whose vector ALU work is two FP64 operations per pair:
// pass 1 of the wide-classifier model 0.60 ms with the plain table, 0.92 ms with this one, same box, three runs each
// (profiles/r06_notes.txt).  The conflicts are the cheaper evil.
#ifndef HIBAG_WIDE_TAB
#define HIBAG_WIDE_TAB false
#endif
#define HIBAG_WIDE_TAB_CLAMP 65
#define HIBAG_WIDE_TAB_N ((HIBAG_WIDE_TAB_CLAMP + 1) * 32)
__device__ __forceinline__ double table_value_wide(const double *tabw_s, int off)     // off = 8 d
{
	const uint32_t o = min((uint32_t)off, (uint32_t)(8 * HIBAG_WIDE_TAB_CLAMP));
	const uint32_t lane8 = (__builtin_amdgcn_mbcnt_lo(~0u, 0u) & 31u) * 8u;        // (lane & 31) * 8: lanes l and l + 32 are in different groups
	return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tabw_s) + (o << 5) + lane8);
}

// cell += prod_i * TAB[d_i] for the first n_valid records of a block, in order;
// `fin(cell, stored, slot)` at every record that closes a cell (end mask, store mask; cells are padded to
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
// `live(g)`: group g of the block has records worth evaluating (the groups before it then have too).
// LINEAR: record i's distance is D0[i] (i < 16) / D1[i - 16] -- the layout of the walk without lane swaps (walk_blocks, OWN).
// WIDE_TAB: `tab_s` is the bank-interleaved table of the classifiers with several K steps (table_value_wide below).
template <int G, bool AHEAD = false, bool LINEAR = false, bool WIDE_TAB = false, class Live, class Fin>
__device__ __forceinline__ void block_accumulate(ConstPtr<double> fac, typename FactorGroup<G>::type F, uint32_t endmask, uint32_t storemask, Live &&live,
	const v16i &D0, const v16i &D1, double &cell, uint32_t fresh, const double *tab_s, Fin &&fin)
{
	typedef typename FactorGroup<G>::type FG;
	// (a copy of the end mask for the "starts" tests below: tested on the same register as the "end" test of the group before,
	// the two become ONE test kept as a 64-bit mask -- a select and two mask operations where two bit tests do)
	uint32_t startmask = endmask;
	asm volatile("" : "+s"(startmask));
	auto look_up = [&](int g, double (&t)[G]) {
#pragma unroll
		for (int q = 0; q < G; q++) {         // D = 8*d: already the byte offset into the table
			const int i = G * g + q;          // record i = 8 m + r  ->  r < 4 ? D0[4 m + r] : D1[4 m + r - 4]
			const int off = LINEAR ? (i < 16 ? D0[i] : D1[i - 16]) : (i & 7) < 4 ? D0[4 * (i >> 3) + (i & 3)] : D1[4 * (i >> 3) + (i & 3)];
			t[q] = WIDE_TAB ? table_value_wide(tab_s, off) : table_value(tab_s, off);
		}
	};
	double tt[AHEAD ? 2 : 1][G];
	if (AHEAD) look_up(0, tt[0]);
#pragma unroll
	for (int g = 0; g < 32 / G; g++) {
		if (!live(g)) break;
		double (&t)[G] = tt[AHEAD ? (g & 1) : 0];
		if (!AHEAD) look_up(g, t);
		// the look-ups are waited for HERE (a use of the first one; LDS returns in order, and a scalar load in flight makes it a
		// wait for everything), and only then are the next group's factors requested: they have this group's arithmetic to arrive.
		// (AHEAD: the NEXT group's look-ups are requested here as well, before this group is added up.  Pass 2 ships with it
		// -- measured three times on round 5's loops: between +1 % and -0.6 %, the last two runs -0.5 % each, for six registers it
		// has to spare; pass 1 has none to spare: +7 %.)
		asm volatile("" : "+v"(t[0]));
		__builtin_amdgcn_sched_barrier(0);
		FG Fn = F;
		if (!ABL_NOFAC && g + 1 < 32 / G) Fn = *(ConstPtr<FG>)(fac + G * (g + 1));
		if (AHEAD && g + 1 < 32 / G && live(g + 1)) look_up(g + 1, tt[(g + 1) & 1]);
		__builtin_amdgcn_sched_barrier(0);
		// the record before this group closed a cell: the caller's word for the block's first group, the end mask's own bit for
		// the others (a bit test and a branch where a carried flag cost a select and a compare per group)
		const bool starts = g == 0 ? fresh != 0 : (startmask & (1u << (G * g - 1))) != 0;
		if (starts) { cell = F[0] * t[0]; asm volatile("" : "+v"(cell)); }     // (the asm keeps this a scalar branch, not a select)
		else cell += F[0] * t[0];
#pragma unroll
		for (int q = 1; q < G; q += 2) {      // cells are padded to an even number of records: only odd positions close one
			cell += F[q] * t[q];
			const bool end = (endmask & (1u << (G * g + q))) != 0;
			const bool stored = (storemask & (1u << (G * g + q))) != 0;
			if (q + 1 < G) {
				if (end) { fin(cell, stored, G * g + q); cell = F[q + 1] * t[q + 1]; }
				else cell += F[q + 1] * t[q + 1];
			} else if (end) fin(cell, stored, G * g + q);
		}
		F = Fn;
	}
}

// `fresh` for the block behind one whose first n_valid records were gone through in groups of G: the end bit of the last of
// them (nothing gone through: unchanged)
template <int G>
__device__ __forceinline__ uint32_t fresh_behind(uint32_t fresh, uint32_t endmask, int n_valid)
{
	const int groups = n_valid >= 32 ? 32 / G : (n_valid + G - 1) / G;
	return groups > 0 ? (endmask >> (G * groups - 1)) & 1u : fresh;
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
// OWN (with PRE): NO LANE SWAPS.  A 32 x 32 matrix result leaves rows 8m .. 8m+3 of sample column j in lane j and rows
// 8m+4 .. 8m+7 in lane j + 32, and sixteen v_permlane32_swap per block (1.7 FP64 operations each) gave every lane its own
// sample's 32 rows.  Instead the A operand holds each of SIXTEEN slots in two rows -- row 8m+q with the slot's image in K half 0
// and zeros in K half 1, row 8m+4+q the other way round -- and the B operand's column j has sample j in K half 0 and sample
// j + 32 in K half 1 (load_operand_own_sample): lane j then receives slot . sample j, lane j + 32 slot . sample (j + 32), both
// in register r = 4m + q.  The two images of a slot (the "sum" side and the "pair" side of the dot product) take two matrix
// instructions that accumulate, the block's 32 slots two such chains: four matrix instructions instead of two, no swap, and
// record i simply sits in register i.  The four A operands are the block's prebuilt row itself, read with one per-lane offset
// (a lane whose row-half is zeros does not load and keeps its zeros).
// MEASURED (round 5, profiles/r05_pass2_notes.txt item 20): every output bit-identical, pass 1 0.760-0.786 ms against 0.743-0.747
// with the swaps -- two more matrix instructions (which FP64 work does not overlap with) and three more loads per block cost
// more than sixteen swaps, whose removal alone is worth 8 %.  Kept as a variant (-DTOTAL_OWN=true), not shipped.
template <int ENG, int G, bool PRE, bool OWN = false, class Fin>
__device__ __forceinline__ void walk_blocks(const HibagModelView &M, uint64_t at, int nblk, int lane, ListCursor &cur,
	__amdgpu_buffer_rsrc_t hp, int k, const LaneOperand &T, const WideSrc &wide, const double *tab_s, double &cell, Fin &&fin)
{
	static_assert(!PRE || ENG == HIBAG_ENGINE_FP4, "prebuilt rows exist for one-step FP4 classifiers only");
	if (nblk <= 0) return;
	typedef typename FactorGroup<G>::type FG;
	uint32_t fresh = 0;                              // block_accumulate: the record before closed a cell
	ConstPtr<double> fac = as_const(M.pfac) + at;                            // this segment's factors and headers
	ConstPtr<u32x4> hdr = (ConstPtr<u32x4>)(as_const(M.phdr) + at / HIBAG_PLIST_DWORDS * 4);
	if (PRE && OWN) {
		const uint64_t blk = at / HIBAG_PLIST_DWORDS;
		const uint64_t left = (M.parow_blocks - blk) * 1024u;
		const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(M.parow + blk * 64), 0,
			left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)left, 0x00020000);
		const int row = lane & 31;
		const bool supplies = ((row >> 2) & 1) == (lane >> 5);        // this lane's (row, K half) of the A operand is not zeros
		const int vo = (4 * (row >> 3) + (row & 3)) * 16;              // its slot among the chain's sixteen (an entry is 16 bytes)
		// A[2 c + image]: chain c = slots 16 c .. 16 c + 15; the prebuilt row has the "sum" images of the 32 slots, then the "pair" images
		v4i A[4];
#pragma unroll
		for (int k = 0; k < 4; k++) A[k] = v4i{0, 0, 0, 0};
		auto request_rows = [&](uint32_t soff) {
			if (supplies) {
				int v = vo;
				asm volatile("" : "+v"(v));                   // (kept out of the loop-invariant code: the four offsets as the instructions' immediates, not as registers)
#pragma unroll
				for (int k = 0; k < 4; k++)
					A[k] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(pr, v + (k & 1) * 512 + (k >> 1) * 256, (int)soff, 0));
			}
		};
		request_rows(0);
		u32x4 H_n = hdr[0];
		FG F_n = *(ConstPtr<FG>)fac;
		uint32_t soff = 1024;
		const v8i b_sum = {T.b[0][0][0], T.b[0][0][1], T.b[0][0][2], T.b[0][0][3], 0, 0, 0, 0};
		const v8i b_pair = {T.b[1][0][0], T.b[1][0][1], T.b[1][0][2], T.b[1][0][3], 0, 0, 0, 0};
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
			if (n_valid > 0) {
				v16f x0, x1;
#pragma unroll
				for (int r = 0; r < 16; r++) { x0[r] = 0.0f; x1[r] = 0.0f; }
				{
					// (the two chains interleaved: a matrix instruction that accumulates onto the one before it waits for its result)
					const v8i a0 = {A[0][0], A[0][1], A[0][2], A[0][3], 0, 0, 0, 0}, a1 = {A[1][0], A[1][1], A[1][2], A[1][3], 0, 0, 0, 0};
					const v8i a2 = {A[2][0], A[2][1], A[2][2], A[2][3], 0, 0, 0, 0}, a3 = {A[3][0], A[3][1], A[3][2], A[3][3], 0, 0, 0, 0};
					// (always all four: pass 1's blocks are 96 % full, and a branch here lets the compiler put the next block's loads in
					// front of the second chain -- whose operands' wait then becomes a wait for those loads)
					x0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b_sum, x0, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, HIBAG_FP4_SCALE_B_LO);
					x1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, b_sum, x1, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, HIBAG_FP4_SCALE_B_LO);
					__builtin_amdgcn_sched_barrier(0);
					x0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b_pair, x0, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, HIBAG_FP4_SCALE_B_HI);
					x1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a3, b_pair, x1, 4, 4, 0, HIBAG_FP4_SCALE_A, 0, HIBAG_FP4_SCALE_B_HI);
					// (both results count as used HERE: the second chain is only read by the block's later groups, and the compiler
					// would sink its matrix instructions down to them -- behind the next block's loads, in the middle of the additions)
					asm volatile("" : "+v"(x0), "+v"(x1));
				}
				__builtin_amdgcn_sched_barrier(0);
				request_rows(soff);                           // the next block's, behind the instructions that read this one's
				__builtin_amdgcn_sched_barrier(0);
				const v16i D0 = __builtin_bit_cast(v16i, x0), D1 = __builtin_bit_cast(v16i, x1);
				block_accumulate<G, false, true>(fac + (size_t)b * HIBAG_PLIST_DWORDS, F, endmask, storemask, [&](int g) { return G * g < n_valid; }, D0, D1, cell, fresh, tab_s, fin);
				fresh = fresh_behind<G>(fresh, endmask, n_valid);
			} else {
				request_rows(soff);
			}
			soff += 1024;
		}
		if (fresh) cell = 0;
		return;
	}
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
#ifdef HIBAG_TOTAL_PRIO            // (measured variant: the head of a block at raised priority)
			__builtin_amdgcn_s_setprio(1);
#endif
			v4i a = arow;
			asm volatile("" : "+v"(a));                   // (this block's rows have arrived: requested a block ago)
			if (n_valid > 0) {
				v16i D0, D1;
				block_mfma<ENG>(a, v4i{0, 0, 0, 0}, lane, v4i{0, 0, 0, 0}, T, D0, D1);
				__builtin_amdgcn_sched_barrier(0);
				arow = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(pr, vo, (int)soff, 0));    // the next block's, behind the instructions that read this one's
				__builtin_amdgcn_sched_barrier(0);
#ifdef HIBAG_TOTAL_PRIO
				__builtin_amdgcn_s_setprio(0);
#endif
				block_own_sample(D0, D1, n_valid);
				block_accumulate<G>(fac + (size_t)b * HIBAG_PLIST_DWORDS, F, endmask, storemask, [&](int g) { return G * g < n_valid; }, D0, D1, cell, fresh, tab_s, fin);
				fresh = fresh_behind<G>(fresh, endmask, n_valid);
			} else {
				arow = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(pr, vo, (int)soff, 0));
			}
			soff += 1024;
		}
		if (fresh) cell = 0;
		return;
	}
#ifndef HIBAG_WIDE_OLDWALK                   // (variant for A/B timing: round 5's walk, which fetches the further steps' operands where they are used)
	if (ENG == HIBAG_ENGINE_FP4W) {
		// An FP4 classifier of `wide.nstep` (2 .. 4) K steps; `k` = SNPs of its LAST step, the others have HIBAG_FP4_STEP_SNPS.
		// Round 6: until now the further steps' images and B operands were fetched where they were used -- two dependent gathers
		// and two operand loads per step, each waited for on the spot, three times per block -- and the kernel ran at a quarter
		// of its issue floor.  Now the B operands of the further steps, which are the same for every block, live in registers,
		// and a block's images of ALL steps are requested a block ahead, right behind the matrix instructions that read the
		// current ones (as the prebuilt rows of the one-step walk are); `tab_s` is the bank-interleaved table.
		const int ns = wide.nstep;
		const uint32_t ES = 4u * (uint32_t)HIBAG_FP4_ENTRY_DWORDS(ns);
		const int vo_i = (lane & 31) * 4;
		const uint32_t BB = 4 * HIBAG_PLIST_DWORDS;
		const uint64_t left = (M.plist_dwords - at) * 4;
		const __amdgpu_buffer_rsrc_t pl = __builtin_amdgcn_make_buffer_rsrc((void *)(M.plist + at), 0,
			left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)left, 0x00020000);
		v4i wb[HIBAG_FP4_MAX_STEPS - 1][2];
#pragma unroll
		for (int j = 1; j < HIBAG_FP4_MAX_STEPS; j++) {
			wb[j - 1][0] = wb[j - 1][1] = v4i{0, 0, 0, 0};
			if (j < ns) {
				const uint4 *row = wide.bt + ((size_t)(wide.bt_row + 2 * j) * wide.n_group + wide.group) * HIBAG_WAVE;
				const uint4 u0 = row[lane], u1 = row[wide.n_group * HIBAG_WAVE + lane];
				wb[j - 1][0] = v4i{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w};
				wb[j - 1][1] = v4i{(int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
			}
		}
		// the steps' constant nibbles: every step but the last has HIBAG_FP4_STEP_SNPS SNPs
		const v4i c_mid = fp4_offset_term(HIBAG_FP4_STEP_SNPS, lane), c_last = fp4_offset_term(k, lane);
		uint32_t soff = 0;
		uint32_t idx_c = __builtin_amdgcn_raw_buffer_load_b32(pl, vo_i, soff, 0), idx_n = __builtin_amdgcn_raw_buffer_load_b32(pl, vo_i, soff + BB, 0);
		v4i im[HIBAG_FP4_MAX_STEPS][2];
		auto request_images = [&](uint32_t word) {
			const uint32_t o1 = (word & 0xFFFFu) * ES, o2 = ((word >> 16) & 0x3FFFu) * ES;
			im[0][0] = load_hap_image(hp, o1); im[0][1] = load_hap_image(hp, o2);
#pragma unroll
			for (int j = 1; j < HIBAG_FP4_MAX_STEPS; j++)
				if (j < ns) { im[j][0] = load_hap_image(hp, o1 + 16u + 16u * (uint32_t)j); im[j][1] = load_hap_image(hp, o2 + 16u + 16u * (uint32_t)j); }
		};
#pragma unroll
		for (int j = 0; j < HIBAG_FP4_MAX_STEPS; j++) im[j][0] = im[j][1] = v4i{0, 0, 0, 0};
		request_images(idx_c);
		u32x4 H_n = hdr[0];
		FG F_n = *(ConstPtr<FG>)fac;
		for (int b = 0; b < nblk; b++) {
			const u32x4 H = H_n;
			const FG F = F_n;
			const uint32_t endmask = abl_endmask(H[0]), storemask = abl_storemask(H[1]);
			const int n_valid = (int)H[2];
			asm volatile("" :: "s"(n_valid));             // (the header is waited for HERE, before the next scalar loads are issued)
			__builtin_amdgcn_sched_barrier(0);
			H_n = hdr[b + 1];
			F_n = *(ConstPtr<FG>)(fac + (size_t)(b + 1) * HIBAG_PLIST_DWORDS);
			idx_c = idx_n;
			idx_n = __builtin_amdgcn_raw_buffer_load_b32(pl, vo_i, soff + 2 * BB, 0);
			v16i D0, D1;
			if (n_valid > 0) {
				v16f d0, d1;
#pragma unroll
				for (int r = 0; r < 16; r++) { d0[r] = 0.0f; d1[r] = 0.0f; }
				fp4_step(im[0][0], im[0][1], lane, c_mid, T.b[0][0], T.b[1][0], d0, d1);
#pragma unroll
				for (int j = 1; j < HIBAG_FP4_MAX_STEPS; j++)
					if (j < ns) fp4_step(im[j][0], im[j][1], lane, j == ns - 1 ? c_last : c_mid, wb[j - 1][0], wb[j - 1][1], d0, d1);
				asm volatile("" : "+v"(d0), "+v"(d1));    // (the matrix instructions stay HERE, in front of the next block's loads)
				D0 = __builtin_bit_cast(v16i, d0);
				D1 = __builtin_bit_cast(v16i, d1);
			}
			__builtin_amdgcn_sched_barrier(0);
			request_images(idx_c);                        // block b + 1's, into the registers the matrix instructions have just read
			__builtin_amdgcn_sched_barrier(0);
			if (n_valid > 0) {
				block_own_sample(D0, D1, n_valid);
				block_accumulate<G, false, false, HIBAG_WIDE_TAB>(fac + (size_t)b * HIBAG_PLIST_DWORDS, F, endmask, storemask, [&](int g) { return G * g < n_valid; }, D0, D1, cell, fresh, tab_s, fin);
				fresh = fresh_behind<G>(fresh, endmask, n_valid);
			}
			soff += BB;
		}
		if (fresh) cell = 0;
		return;
	}
#endif
	// (FP4W is handled above; the flag stays for the shared declarations below)
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
			block_accumulate<G>(fac + (size_t)b * HIBAG_PLIST_DWORDS, F, endmask, storemask, [&](int g) { return G * g < n_valid; }, D0, D1, cell, fresh, tab_s, fin);
			fresh = fresh_behind<G>(fresh, endmask, n_valid);
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

__device__ __forceinline__ void stage_table_wide(const HibagModelView &M, double *tabw_s)
{
	for (int i = threadIdx.x; i < HIBAG_WIDE_TAB_N; i += blockDim.x) tabw_s[i] = M.tab[i >> 5];
	__syncthreads();
}

__device__ __forceinline__ void stage_table(const HibagModelView &M, double *tab_s, int n = HIBAG_TAB_N)
{
	for (int i = threadIdx.x; i < n; i += blockDim.x) tab_s[i] = M.tab[i];
	__syncthreads();
}

#endif
