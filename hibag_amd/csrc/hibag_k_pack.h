// hibag_k_pack.h -- part of hibag_kernels.hip (included there, one translation unit: the walks are templates that inline into
// their kernels): k_codes / k_bed_codes / k_bed_geno / k_pack: the raw genotypes to byte codes, B operands and classifier weights.
#ifndef HIBAG_K_PACK_H_
#define HIBAG_K_PACK_H_

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

// k_codes_rows: the same for a SNP-MAJOR matrix, int32 [rows][ld] with one row of genotypes per SNP (a C / numpy
// [snp][sample] array, the transpose of R's memory; `ld` >= n_samp elements between rows): no transpose to make -- block =
// 256 samples of one model SNP, the read of the row and the write of the codes both contiguous.  col[k] = the row that
// holds model SNP k (-1 = absent -> missing; nullptr = row k), flip[k] != 0 reverses its allele count.
__global__ __launch_bounds__(256) void k_codes_rows(HibagModelView M, HibagBatchView B,
	const int32_t *__restrict__ geno, size_t ld, const int32_t *__restrict__ col, const int32_t *__restrict__ flip,
	uint8_t *__restrict__ codes)
{
	const int k = blockIdx.y;
	const int s = blockIdx.x * 256 + threadIdx.x;
	if (s >= B.n_pad) return;
	const int c = col ? col[k] : k;
	uint8_t v = 3;
	if (s < B.n_samp && c >= 0) {
		const int g = geno[(size_t)c * ld + s];
		if (g >= 0 && g <= 2) v = (flip && flip[k]) ? (uint8_t)(2 - g) : (uint8_t)g;
	}
	codes[(size_t)k * B.n_pad + s] = v;
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
	// (the model's tables through the constant address space: the kernel stores, so a plain wave-uniform load would be a VECTOR
	// load plus a readfirstlane -- three dependent memory round trips per SNP instead of scalar-cache hits)
	const int k = as_const(M.n_snp_c)[c];
	const int nwp = as_const(M.nwp)[c];
	const ConstPtr<int> idx = as_const(M.snp_index) + as_const(M.snp_off)[c];
	const ConstPtr<int> snp_weight = as_const(M.snp_weight);
	const int row0 = as_const(M.mask_row)[c];
	const int nkb = as_const(M.engine)[c];            // matrix-engine variant, 0 = VALU engine
	const int bt_row_c = as_const(M.bt_row)[c];
	int num = 0, den = 0;
	if (nkb > 0) {
	// one pass over the SNPs of each K step (independent byte loads, several in flight): all k <= 32 of them, or 28 per
	// step of a multi-step FP4 classifier
	const int steps = as_const(M.n_step)[c];
	for (int st = 0; st < steps; st++) {
		const int j0 = steps > 1 ? HIBAG_FP4_STEP_SNPS * st : 0, kj = steps > 1 ? min(HIBAG_FP4_STEP_SNPS, k - j0) : k;
		// The codes (0, 1, 2, 3 = missing) are gathered two bits per SNP -- one shift-or each -- and the three bit fields the
		// operands are made of (bit j: g == 2, g == 0, g == 1 at SNP j0 + j; a missing SNP is in none) come out of the two packed
		// words at the end: 70 vector instructions for a K step instead of nine per SNP.
		uint32_t packed[2] = {0u, 0u};         // SNPs 0..15, 16..31 of the step
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const int jb = 16 * h, je = min(kj, jb + 16);
			uint32_t w = 0;
#pragma unroll 8
			for (int j = jb; j < je; j++) {
				const int snp = idx[j0 + j];
				const uint32_t g = codes[(size_t)snp * B.n_pad + s];
				const int wt = snp_weight[snp];
				den += wt;
				if (g != 3) num += wt;
				w |= g << (2 * (j - jb));
			}
			packed[h] = w;
		}
		auto even_bits = [](uint32_t x) {      // bits 0, 2, 4 ... 30 -> bits 0 ... 15
			x &= 0x55555555u;
			x = (x | (x >> 1)) & 0x33333333u;
			x = (x | (x >> 2)) & 0x0F0F0F0Fu;
			x = (x | (x >> 4)) & 0x00FF00FFu;
			return (x | (x >> 8)) & 0xFFFFu;
		};
		uint32_t X = 0, Z = 0, E = 0;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const uint32_t lo = packed[h], hi = packed[h] >> 1;
			X |= even_bits(hi & ~lo) << (16 * h);
			E |= even_bits(lo & ~hi) << (16 * h);
			Z |= even_bits(~(hi | lo)) << (16 * h);
		}
		Z &= kj >= 32 ? ~0u : (1u << kj) - 1u;     // (positions behind the step's last SNP hold code 0)
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
				B.bt[((size_t)(bt_row_c + 2 * st + n) * gridDim.x + blockIdx.x) * HIBAG_WAVE + h * 32 + (lane & 31)] =
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
					B.bt[((size_t)(bt_row_c + n * 2 + m) * gridDim.x + blockIdx.x) * HIBAG_WAVE + h * 32 + (lane & 31)] =
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
				const int wt = snp_weight[snp];
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

#endif
