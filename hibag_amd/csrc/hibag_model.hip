// hibag_model.hip -- the model of libhibag_hip.so: classifiers come in through hibag_hip_model_new / _add_classifier
// (HIBAG_New, HIBAG_NewClassifierHaplo, src/HIBAG.cpp:486-503, :817-841), hibag_hip_model_finalize lays them out for the
// kernels (hibag_device.h: haplotype tables, pair lists with their factors and headers, prebuilt A-operand rows, the
// second pass's block stream, tiles, work items) and decides which cell sums pass 1 stores; replicas and classifier shards
// are models built from the same classifiers on another device.  No compute here.

#include "hibag_internal.h"

namespace hibag_detail {



// The mutation/error weights exp(d*log(1e-5)), TAB[0]=1, non-finite -> 0:
// the same expression, evaluated by the host libm like the reference does in
// its static initialiser (src/LibHLA.cpp:166-183).
void build_table(double *tab)
{
	const double min_rare_freq = 1e-5;   // inst/include/LibHLA_ext.h:230
	for (int i = 0; i < HIBAG_TAB_N; i++) tab[i] = std::exp(i * std::log(min_rare_freq));
	tab[0] = 1;
	for (int i = 0; i < HIBAG_TAB_N; i++)
		if (!std::isfinite(tab[i])) tab[i] = 0;
}

int check_classifier_args(hibag_hip_model *m, int n_snp_c, const int32_t *snpidx, int n_haplo,
	const double *freq, const int32_t *hla)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	if (m->finalized) return hibag_fail(HIBAG_HIP_ESTATE, "model already finalized");
	if (n_snp_c < 0 || n_snp_c > HIBAG_HIP_MAX_SNP_IN_CLASSIFIER)
		return hibag_fail(HIBAG_HIP_EINVAL, "there are too many SNP markers in a classifier (%d > %d).",
			n_snp_c, HIBAG_HIP_MAX_SNP_IN_CLASSIFIER);
	if (n_haplo < 0 || (n_haplo > 0 && (!freq || !hla)))
		return hibag_fail(HIBAG_HIP_EINVAL, "invalid haplotype list");
	if (snpidx)
		for (int i = 0; i < n_snp_c; i++)
			if (snpidx[i] < 0 || snpidx[i] >= m->n_snp)
				return hibag_fail(HIBAG_HIP_EINVAL, "SNP index %d out of range [0,%d)", snpidx[i], m->n_snp);
	for (int i = 0; i < n_haplo; i++) {
		if (hla[i] < 0 || hla[i] >= m->n_hla)
			return hibag_fail(HIBAG_HIP_EINVAL, "HLA allele index %d out of range [0,%d)", hla[i], m->n_hla);
		if (i > 0 && hla[i] < hla[i - 1])
			return hibag_fail(HIBAG_HIP_EINVAL, "haplotypes must be grouped by ascending HLA allele index");
	}
	return 0;
}

void push_classifier(hibag_hip_model *m, int n_snp_c, const int32_t *snpidx, int n_haplo,
	const double *freq, const int32_t *hla, std::vector<uint64_t> &&bits)
{
	HostClassifier c;
	c.n_snp = n_snp_c;
	if (snpidx) c.snpidx.assign(snpidx, snpidx + n_snp_c);
	else m->have_snpidx = false;
	c.freq.assign(freq, freq + n_haplo);
	c.hla.assign(hla, hla + n_haplo);
	c.bits = std::move(bits);
	m->cls.push_back(std::move(c));
}

// Words per pair record: ceil(3k/32) rounded up to a width the kernels are
// specialised for (HIBAG_DISPATCH_NWP in hibag_kernels.hip).
int round_nwp(int n)
{
	for (int v : {1, 2, 3, 4, 6, 8, 10, 12})
		if (n <= v) return v;
	return HIBAG_MAX_NWP;
}

// OR the low `nbits` bits of the 128-bit value src into the multiword string dst at bit `pos`.
void or_bits(uint32_t *dst, const uint64_t src[2], int nbits, int pos)
{
	for (int i = 0; i < nbits; i++)
		if ((src[i >> 6] >> (i & 63)) & 1) dst[(pos + i) >> 5] |= 1u << ((pos + i) & 31);
}

// Flatten one classifier's _PostProb2 loop nest (src/LibHLA.cpp:1776-1821) into
// pair records in the reference's visiting order.  For every allele-pair cell
// (posterior order) appends whole chunks to `stream` and returns the chunk count
// per cell in `cell_chunks[P]`.  The frequency factor is rounded exactly as the
// reference does: f1*f1 for the leading diagonal term (:1786), (2*f1)*f2 else
// (:1789-1793, :1808-1812); this file is compiled with -ffp-contract=off.
void build_pair_stream(const HostClassifier &k, int n_hla, int nwp, const int *st,
	std::vector<uint32_t> &stream, std::vector<uint32_t> &cell_chunks)
{
	const int ks = k.n_snp;
	const uint64_t lowmask[2] = {
		ks >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << ks) - 1),
		ks >= 128 ? ~(uint64_t)0 : (ks <= 64 ? 0 : (((uint64_t)1 << (ks - 64)) - 1)) };
	std::vector<uint32_t> recw;       // records of the current cell: nwp words each
	std::vector<double> recp;
	auto emit = [&](int a, int b, double prod) {
		const uint64_t *A = &k.bits[2 * (size_t)a], *Bb = &k.bits[2 * (size_t)b];
		const uint64_t same[2] = { ~(A[0] ^ Bb[0]) & lowmask[0], ~(A[1] ^ Bb[1]) & lowmask[1] };
		const size_t at = recw.size();
		recw.resize(at + nwp, 0);
		or_bits(&recw[at], A, ks, 0);
		or_bits(&recw[at], Bb, ks, ks);
		or_bits(&recw[at], same, ks, 2 * ks);
		recp.push_back(prod);
	};
	auto flush = [&]() -> uint32_t {
		const size_t n = recp.size();
		const size_t nchunk = (n + HIBAG_CHUNK - 1) / HIBAG_CHUNK;
		for (size_t ch = 0; ch < nchunk; ch++) {
			const size_t base = stream.size();
			stream.resize(base + HIBAG_CHUNK_DWORDS(nwp), 0);
			for (int r = 0; r < HIBAG_CHUNK; r++) {
				const size_t i = ch * HIBAG_CHUNK + r;
				double prod = 0.0;                    // padding record: + (+0.0 * TAB[d]) is exact
				if (i < n) {
					for (int w = 0; w < nwp; w++) stream[base + (size_t)w * HIBAG_CHUNK + r] = recw[i * nwp + w];
					prod = recp[i];
				}
				memcpy(&stream[base + (size_t)nwp * HIBAG_CHUNK + 2 * (size_t)r], &prod, sizeof(double));
			}
		}
		recw.clear(); recp.clear();
		return (uint32_t)nchunk;
	};
	size_t p = 0;
	for (int h1 = 0; h1 < n_hla; h1++) {
		const int a0 = st[h1], a1 = st[h1 + 1];
		for (int a = a0; a < a1; a++) {
			emit(a, a, k.freq[a] * k.freq[a]);
			const double ff = 2 * k.freq[a];
			for (int b = a + 1; b < a1; b++) emit(a, b, ff * k.freq[b]);
		}
		cell_chunks[p++] = flush();
		for (int h2 = h1 + 1; h2 < n_hla; h2++) {
			const int b0 = st[h2], b1 = st[h2 + 1];
			for (int a = a0; a < a1; a++) {
				const double ff = 2 * k.freq[a];
				for (int b = b0; b < b1; b++) emit(a, b, ff * k.freq[b]);
			}
			cell_chunks[p++] = flush();
		}
	}
}

// Matrix-core engine: the pair list of a run of cells [p0, p0 + n) of one classifier, appended to `out`
// as blocks of 32 slots (i1 | i2 << 16 | end << 31).  The visiting order inside a
// cell is the reference's (src/LibHLA.cpp:1776-1821: i1 ascending, then i2; the leading diagonal pair
// (i, i) first on the diagonal cells).  Cells are padded to an even slot count with the classifier's
// all-zero haplotype `pad` (frequency 0: the slot adds +0.0); the end flag marks the slot that closes
// a cell; the unused slots behind the last cell point at `pad` too.  (h1, h2) of the first cell p0 are given; returns the number of blocks.
int append_pair_blocks(const int *st, int n_hla, int h1, int h2, int p0, int n_cells, uint32_t pad, std::vector<uint32_t> &out,
	const uint8_t *mark, const uint8_t *skip)
{
	// mark[p]: the closing slot of cell p carries the STORE flag; skip[p]: cell p is left out (pass-2 lists: its sum
	// comes from memory); both indexed by posterior cell, either may be null
	const uint32_t pad_idx = pad | (pad << 16);
	size_t base = 0;
	int fill = 32, n_blocks = 0;                 // slots used in the open block (32 = none open)
	uint32_t end_flags = HIBAG_PLIST_END;
	auto slot = [&](uint32_t idx, bool end) {
		if (fill == 32) {
			base = out.size();
			out.resize(base + HIBAG_PLIST_DWORDS, pad_idx);
			fill = 0; n_blocks++;
		}
		out[base + fill++] = idx | (end ? end_flags : 0u);
	};
	for (int c = 0; c < n_cells; c++) {
		const int a0 = st[h1], a1 = st[h1 + 1], b0 = st[h2], b1 = st[h2 + 1];
		const uint64_t n = h1 == h2 ? (uint64_t)(a1 - a0) * (a1 - a0 + 1) / 2 : (uint64_t)(a1 - a0) * (b1 - b0);
		if (n && !(skip && skip[p0 + c])) {
			end_flags = HIBAG_PLIST_END | (mark && mark[p0 + c] ? HIBAG_PLIST_STORE : 0u);
			uint64_t i = 0;
			const uint64_t total = n + (n & 1);
			if (h1 == h2) {
				for (int a = a0; a < a1; a++) {
					i++; slot((pad + 1 + (uint32_t)a) | ((uint32_t)a << 16), i == total);      // (a, a): factor f * f
					for (int b = a + 1; b < a1; b++) { i++; slot((uint32_t)a | ((uint32_t)b << 16), i == total); }
				}
			} else {
				for (int a = a0; a < a1; a++)
					for (int b = b0; b < b1; b++) { i++; slot((uint32_t)a | ((uint32_t)b << 16), i == total); }
			}
			if (n & 1) slot(pad_idx, true);
		}
		if (++h2 == n_hla) { h1++; h2 = h1; }
	}
	return n_blocks;
}

// Tiles for pass 2: consecutive posterior cells, at most HIBAG_TILE each, cut so
// that the chunk counts (summed over classifiers, plus a per-cell constant) are
// balanced.
// `cap`: cells per tile at most (<= HIBAG_TILE, what a wavefront of pass 2 has LDS rows for).  Pass 2's parallelism is
// (groups of 64 samples) x (tiles): a model of few alleles cut into tiles of fifteen leaves most of the device idle -- the
// reference's bundled HLA-A model, 14 alleles = 105 cells = 7 tiles, made 1,100 wavefronts of a 10,000-sample batch for
// 1,024 SIMDs that hold five each, and pass 2 took 1.7 x pass 1 -- so such a model gets smaller tiles (tile_cap below).
void build_tiles(int P, const std::vector<uint64_t> &cell_work, std::vector<int> &tile_p0, std::vector<int> &tile_n, int cap)
{
	uint64_t total = 0;
	for (int p = 0; p < P; p++) total += cell_work[p] + 1;
	const int TILE = std::max(1, std::min(cap, HIBAG_TILE));
	const int min_tiles = (P + TILE - 1) / TILE;
	const uint64_t target = std::max<uint64_t>(1, total / (uint64_t)std::max(min_tiles, 1));
	// A tile is a wavefront of pass 2, and four tiles make a workgroup: every tile beyond the minimum
	// is another wavefront per sample group (and possibly another, mostly empty, workgroup).  A tile is
	// therefore closed early for balance only while the cells it leaves unused still fit into the
	// minimum number of tiles.
	int slack = min_tiles * TILE - P;
	tile_p0.clear(); tile_n.clear();
	int p = 0;
	while (p < P) {
		int n = 0;
		uint64_t w = 0;
		while (p + n < P && n < TILE) {
			const uint64_t cw = cell_work[p + n] + 1;
			if (n > 0 && w + cw > target + target / 4 && TILE - n <= slack) break;
			w += cw; n++;
		}
		if (p + n < P) slack -= TILE - n;
		tile_p0.push_back(p); tile_n.push_back(n);
		p += n;
	}
}

int finalize_model(hibag_hip_model *m)
{
	if (m->finalized) return hibag_fail(HIBAG_HIP_ESTATE, "model already finalized");
	HIP_TRY(hipSetDevice(m->device));
	const int C = (int)m->cls.size(), nh = m->n_hla, S = m->n_snp;
	const int P = nh * (nh + 1) / 2;

	std::vector<int> n_snp_c(C), nwp(C), snp_off(C), mask_row(C), c_order(C), snp_index,
		snp_weight(std::max(S, 1), 0);
	std::vector<uint64_t> stream_off(std::max(C, 1), 0), cell_work(P, 0);
	std::vector<uint32_t> stream;
	// per classifier: records (haplotype pairs) of every cell, and 4-record chunks of every cell
	std::vector<std::vector<uint32_t>> cell_chunks(C), cell_pairs(C);
	std::vector<std::vector<int>> starts(C);
	std::vector<int> engine(std::max(C, 1), 0), bt_row(std::max(C, 1), 0), cls_nblk(std::max(C, 1), 0), n_step(std::max(C, 1), 1);
	std::vector<int> &mfma_nkb = engine;                 // (non-zero = a matrix engine)
	std::vector<uint32_t> hap, hap_off(std::max(C, 1), 0);
	std::vector<int64_t> pairs(C);
	int bt_rows = 0;
	int rows = 0;
	m->pair_evals = 0;
	int64_t valu_pairs = 0;
	const bool allow_wide = !(getenv("HIBAG_PASS2") && !strcmp(getenv("HIBAG_PASS2"), "recompute"));
	for (int c = 0; c < C; c++) {
		const HostClassifier &k = m->cls[c];
		const int H = (int)k.freq.size();
		n_snp_c[c] = k.n_snp;
		nwp[c] = round_nwp((3 * k.n_snp + 31) / 32);
		snp_off[c] = (int)snp_index.size();
		for (int v : k.snpidx) { snp_index.push_back(v); snp_weight[v]++; }
		if (k.snpidx.empty()) snp_index.insert(snp_index.end(), (size_t)k.n_snp, 0);
		mask_row[c] = rows;
		rows += 2 * nwp[c];
		std::vector<int> &st = starts[c];
		st.assign(nh + 1, 0);
		for (int i = 0; i < H; i++) st[k.hla[i] + 1]++;
		for (int h = 0; h < nh; h++) st[h + 1] += st[h];
		// matrix-core engines: at most 112 SNPs; table indices: first haplotype < 2H + 1 in 16 bits, second < H + 1 in 14
		engine[c] = (m->use_mfma && H < 16384) ? HIBAG_ENGINE_OF(k.n_snp, m->use_fp4) : HIBAG_ENGINE_VALU;
		// (several K steps need their cells stored: not with pass 2 forced to evaluate every pair)
		if (engine[c] == HIBAG_ENGINE_FP4 && k.n_snp > HIBAG_FP4_MAX_SNPS && !allow_wide) engine[c] = HIBAG_ENGINE_VALU;
		n_step[c] = HIBAG_ENGINE_STEPS(engine[c], k.n_snp);
		bt_row[c] = bt_rows;
		bt_rows += HIBAG_ENGINE_ROWS(engine[c], k.n_snp);
		cell_chunks[c].assign(P, 0);
		cell_pairs[c].assign(P, 0);
		if (mfma_nkb[c]) {
			// no record stream: the kernels generate the records from the haplotype table
			hap_off[c] = (uint32_t)hap.size();
			const bool fp4 = engine[c] == HIBAG_ENGINE_FP4;
			const int steps = n_step[c];
			// bits of a haplotype: SNPs [lo, lo + 32) of its 128-bit string
			auto window = [&](int i, int lo) -> uint32_t {
				if (i < 0) return 0u;
				const unsigned __int128 v = ((unsigned __int128)k.bits[2 * (size_t)i + 1] << 64) | k.bits[2 * (size_t)i];
				return (uint32_t)(v >> lo);
			};
			auto entry = [&](double ff, int i, double f) {
				uint32_t w[12 + 4 * (HIBAG_FP4_MAX_STEPS - 1)] = {0};
				int n = 0;
				if (fp4 && steps == 1) {       // two nibble images, both ADDED by the kernel (K layout in hibag_device.h):
					// the "sum" image has nibble s = 2 (the e2m1 code of 1.0) where bit s is set, the "pair" image the code 3 (1.5) --
					// two of them make the code 6 (4.0), so the sum of two pair images is w = 0 / 1.5 / 4 for 0 / 1 / 2 set bits
					const uint32_t bits = window(i, 0);
					for (int sb = 0; sb < 32; sb++) {
						w[sb >> 3] |= ((bits >> sb) & 1u) << (4 * (sb & 7) + 1);
						w[4 + (sb >> 3)] |= (((bits >> sb) & 1u) * 3u) << (4 * (sb & 7));
					}
					if (i >= 0) {
						// ... plus the A-row constants of the offset digits at nibbles k, k + 1: each image carries half of each (sum
						// image: codes 1 and 3, 0.5 + 0.5 = 1 and 3 + 3 = code 6 = 4; pair image: 3 and 3 -> 4, 4).  (Not the padding
						// entry: its rows must stay zero.)
						const int ks = k.n_snp;
						for (int q = 0; q < 2; q++) {
							const int nib = ks + q;
							w[nib >> 3] |= (q == 0 ? 1u : 3u) << (4 * (nib & 7));
							w[4 + (nib >> 3)] |= 3u << (4 * (nib & 7));
						}
					}
					n = 8;
				} else if (fp4) {              // nibble s = 2 (the e2m1 code of 1.0) where bit s is set
					const uint32_t bits = window(i, 0) & ((1u << HIBAG_FP4_STEP_SNPS) - 1);
					for (int sb = 0; sb < 32; sb++) w[sb >> 3] |= ((bits >> sb) & 1u) << (4 * (sb & 7) + 1);
					n = 4;
				} else {                       // byte s = 1 where bit s is set
					const uint32_t bits = window(i, 0);
					for (int sb = 0; sb < 32; sb++) w[sb >> 2] |= ((bits >> sb) & 1u) << (8 * (sb & 3));
					n = 8;
				}
				memcpy(&w[n], &ff, sizeof(double)); memcpy(&w[n + 2], &f, sizeof(double));
				n += 4;
				for (int j = 1; j < steps; j++, n += 4) {      // further K steps: the next 28 SNPs each
					const uint32_t bits = window(i, HIBAG_FP4_STEP_SNPS * j) & ((1u << HIBAG_FP4_STEP_SNPS) - 1);
					for (int sb = 0; sb < 32; sb++) w[n + (sb >> 3)] |= ((bits >> sb) & 1u) << (4 * (sb & 7) + 1);
				}
				hap.insert(hap.end(), w, w + n);
			};
			for (int i = 0; i < H; i++) entry(2 * k.freq[i], i, k.freq[i]);
			entry(0.0, -1, 0.0);                                   // H: the padding entry (frequency +0.0)
			for (int i = 0; i < H; i++) entry(k.freq[i], i, k.freq[i]);   // H+1+i: first of a diagonal pair
			size_t p = 0;
			for (int h1 = 0; h1 < nh; h1++)
				for (int h2 = h1; h2 < nh; h2++) {
					const uint64_t n1 = (uint64_t)(st[h1 + 1] - st[h1]), n2 = (uint64_t)(st[h2 + 1] - st[h2]);
					const uint64_t n = h1 == h2 ? n1 * (n1 + 1) / 2 : n1 * n2;
					if (n > 0xFFFFFFull * HIBAG_CHUNK) return hibag_fail(HIBAG_HIP_EINVAL, "an allele pair of classifier %d has too many haplotype pairs", c);
					cell_pairs[c][p] = (uint32_t)n;
					cell_chunks[c][p++] = (uint32_t)((n + HIBAG_CHUNK - 1) / HIBAG_CHUNK);
				}
		} else {
			if (stream.size() & 1) stream.push_back(0);          // 8-byte alignment of the doubles inside
			stream_off[c] = stream.size();
			build_pair_stream(k, nh, nwp[c], st.data(), stream, cell_chunks[c]);
		}
		for (int p = 0; p < P; p++) cell_work[p] += (uint64_t)cell_chunks[c][p] * (nwp[c] + 2);
		pairs[c] = (int64_t)H * (H + 1) / 2;
		m->pair_evals += pairs[c];
		if (!mfma_nkb[c]) valu_pairs += pairs[c];
		c_order[c] = c;
	}
	if (!m->snp_weight_override.empty()) snp_weight = m->snp_weight_override;
	std::stable_sort(c_order.begin(), c_order.end(), [&](int a, int b) { return pairs[a] * nwp[a] > pairs[b] * nwp[b]; });
	// the walker fetches one chunk ahead: keep a widest-record chunk of slack behind the last record
	stream.insert(stream.end(), HIBAG_CHUNK_DWORDS(HIBAG_MAX_NWP), 0);
	if (snp_index.empty()) snp_index.push_back(0);
	if (hap.empty()) hap.insert(hap.end(), 12, 0u);
	if (hap.size() * sizeof(uint32_t) > 0x7FFFFF00ull) return hibag_fail(HIBAG_HIP_EINVAL, "the model's haplotype tables exceed 2 GB");

	std::vector<int> tile_p0, tile_n;
	// cells per tile: fifteen where that still makes ~48 tiles or more (50 alleles: 85), fewer for models of few alleles, down
	// to four (a visit of fewer cells is mostly block overhead).  HIBAG_TILE_CAP overrides (diagnostic).
	int tile_cap = std::max(4, std::min(HIBAG_TILE, (P + 47) / 48));
	if (const char *e = getenv("HIBAG_TILE_CAP")) tile_cap = std::max(1, std::min(HIBAG_TILE, atoi(e)));
	build_tiles(P, cell_work, tile_p0, tile_n, tile_cap);
	const int n_tile = (int)tile_p0.size();
	std::vector<int> tile_h1(n_tile, 0), tile_h2(n_tile, 0);      // (h1, h2) of every tile's first cell
	{
		int t = 0, p = 0;
		for (int h1 = 0; h1 < nh && t < n_tile; h1++)
			for (int h2 = h1; h2 < nh && t < n_tile; h2++, p++)
				if (p == tile_p0[t]) { tile_h1[t] = h1; tile_h2[t] = h2; t++; }
	}
	// Which cell sums pass 1 stores for pass 2 (HibagModelView::store_cells).  Measured on MI355X: evaluating a haplotype
	// pair again costs ~0.25 ps per sample, a stored cell ~2.2 ps (written in pass 1, read in pass 2, both at HBM speed).
	// A model with many pairs per non-empty cell (the DRB1 shape: 73) stores every cell and pass 2 only reads; otherwise
	// (the HLA-B benchmark model: 8.5) the cells with more than `store_above` pairs of the matrix-engine classifiers are
	// stored -- 15 % of its cells hold 62 % of its pairs -- and pass 2 evaluates the rest (thresholds 8 .. 16 measure the same;
	// below, the stores slow pass 1 down more than pass 2 gains).  HIBAG_PASS2 = stream |
	// hybrid | recompute and HIBAG_STORE_PAIRS override.
	uint64_t store_above = 12;
	if (const char *e = getenv("HIBAG_STORE_PAIRS")) store_above = (uint64_t)std::max(0, atoi(e));
	uint32_t fit_min = 5;
	if (const char *e = getenv("HIBAG_STORE_FIT")) fit_min = atoi(e) > 0 ? (uint32_t)atoi(e) : ~0u;
	if (getenv("HIBAG_PASS2") && !strcmp(getenv("HIBAG_PASS2"), "recompute")) { store_above = ~(uint64_t)0; fit_min = ~0u; }   // (no cell of theirs is stored)
	// Pass 2 evaluates the pairs of one-step FP4 classifiers only (k_accum's block stream); a classifier on any other engine
	// -- int8 (29..32 SNPs), FP4 in several K steps, VALU -- has all its cells stored by pass 1 and read back.
	auto pass2_evaluates = [&](int c) { return engine[c] == HIBAG_ENGINE_FP4 && n_step[c] == 1; };
	{
		long long n_cells = 0, n_big = 0;
		double cost = 0;                                   // pairs, a VALU-engine pair counted five times (what it costs)
		for (int c = 0; c < C; c++) {
			cost += (double)pairs[c] * (mfma_nkb[c] ? 1.0 : 5.0);
			for (int p = 0; p < P; p++) {
				n_cells += cell_chunks[c][p] != 0;
				n_big += pass2_evaluates(c) ? cell_pairs[c][p] > store_above : cell_chunks[c][p] != 0;
			}
		}
		m->store_mode = C == 0 ? 0 : cost >= 14.0 * (double)std::max<long long>(n_cells, 1) ? 1 : n_big ? 2 : 0;
		if (const char *e = getenv("HIBAG_PASS2")) {
			if (!strcmp(e, "stream")) m->store_mode = C > 0;
			else if (!strcmp(e, "recompute")) m->store_mode = n_big ? 2 : 0;       // (only what pass 2 cannot evaluate is stored)
			else if (!strcmp(e, "hybrid")) m->store_mode = n_big ? 2 : 0;
		}
	}
	{
		// nothing pass 2 could evaluate (no one-step FP4 classifier, e.g. HIBAG_ENGINE=valu): read everything back
		bool any_eval = false;
		for (int c = 0; c < C; c++) any_eval |= pass2_evaluates(c);
		if (m->store_mode == 2 && !any_eval) m->store_mode = 1;
	}
	const int store_mode = m->store_mode;
	// stored[c][p]: pass 1 stores the sum of cell p of classifier c.  Mode 2: the cells of a matrix-engine classifier with
	// more than `store_above` pairs, at most HIBAG_STORED_PER_VISIT per (classifier, tile) -- the ones with the most pairs --
	// which is what pass 2 keeps in registers for a visit, and every cell of a VALU-engine classifier; mode 1: every non-empty cell.
	std::vector<std::vector<uint8_t>> stored(C);
	for (int c = 0; c < C; c++) {
		stored[c].assign(P, 0);
		if (store_mode == 1) { for (int p = 0; p < P; p++) stored[c][p] = cell_chunks[c][p] != 0; }
		else if (store_mode == 2 && !pass2_evaluates(c)) {
			// pass 2 evaluates one-step FP4 classifiers only: all the cells of the others
			for (int p = 0; p < P; p++) stored[c][p] = cell_chunks[c][p] != 0;
		} else if (store_mode == 2)
			for (int t = 0; t < n_tile; t++) {
				// Largest cells first: a cell with more than `store_above` pairs is stored; so is -- while the visit's
				// remaining pair slots would not fit ONE 32-slot block -- any cell of at least `fit_min` pairs: a second,
				// mostly empty block costs pass 2 more than a stored sum (HIBAG_STORE_FIT=0 switches that off).
				std::vector<std::pair<uint32_t, int>> cells;
				uint32_t slots = 0;                            // pair slots of the visit (cells padded to an even count)
				for (int j = 0; j < tile_n[t]; j++) {
					const uint32_t n = cell_pairs[c][tile_p0[t] + j];
					if (n) { cells.push_back({n, tile_p0[t] + j}); slots += n + (n & 1u); }
				}
				std::stable_sort(cells.begin(), cells.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
				for (size_t i = 0; i < cells.size() && i < HIBAG_STORED_PER_VISIT; i++) {
					const uint32_t n = cells[i].first;
					if (!(n > store_above || (slots > HIBAG_PLIST_DWORDS && n >= fit_min))) break;
					stored[c][cells[i].second] = 1;
					slots -= n + (n & 1u);
				}
			}
	}
	// a cell of a matrix-engine classifier whose sum pass 2 reads instead of evaluating its pairs (mode 2)
	auto stored_big = [&](int c, int p) { return store_mode == 2 && stored[c][p] != 0; };

	// pass 1 lists (non-empty cells per classifier) and pass 2 tile entries
	std::vector<uint32_t> cls_cnt, cls_cell, tile_meta((size_t)std::max(C, 1) * n_tile * HIBAG_TILE_META + 1, 0);
	std::vector<int> cls_off(std::max(C, 1), 0), cls_n(std::max(C, 1), 0);
	std::vector<uint32_t> tile_k0((size_t)std::max(C, 1) * n_tile, 0), tile_nlist((size_t)std::max(C, 1) * n_tile, 0),
		tile_nstored((size_t)std::max(C, 1) * n_tile, 0);
	std::vector<uint64_t> tile_jpack((size_t)std::max(C, 1) * n_tile, 0);
	std::vector<int> n_stored_c(std::max(C, 1), 0);        // cells of the classifier pass 1 stores in mode 2
	m->second_pass_pairs = 0;
	for (int c = 0; c < C; c++) {
		cls_off[c] = (int)cls_cnt.size();
		for (int p = 0; p < P; p++)
			if (cell_chunks[c][p]) { cls_cnt.push_back(cell_chunks[c][p]); cls_cell.push_back((uint32_t)p); }
		cls_n[c] = (int)cls_cnt.size() - cls_off[c];
		cls_cnt.push_back(0); cls_cell.push_back(0);        // the walker reads one count ahead
		uint64_t off = 0;
		int k_first = 0;                                    // non-empty cells of the classifier in earlier tiles
		for (int t = 0; t < n_tile; t++) {
			uint32_t *me = &tile_meta[((size_t)c * n_tile + t) * HIBAG_TILE_META];
			tile_k0[(size_t)c * n_tile + t] = (uint32_t)k_first;
			if (off > 0xFFFFFFFFull) return hibag_fail(HIBAG_HIP_EINVAL, "classifier %d has too many haplotype pairs", c);
			me[1] = (uint32_t)off;
			int k = 0;
			uint64_t jpack = 0;
			for (int j = 0; j < tile_n[t]; j++) {
				const uint32_t n = cell_chunks[c][tile_p0[t] + j];
				if (n > 0xFFFFFFu) return hibag_fail(HIBAG_HIP_EINVAL, "an allele pair of classifier %d has too many haplotype pairs", c);
				if (n) { jpack |= (uint64_t)j << (4 * k); me[4 + k++] = ((uint32_t)j << 24) | n; off += n; }
			}
			me[0] = (uint32_t)k;
			me[2] = (uint32_t)jpack; me[3] = (uint32_t)(jpack >> 32);
			{
				// what pass 2 gets per (classifier, tile): the cells it evaluates (in closing order), then those it reads
				uint64_t jp = 0;
				int nl = 0, ns = 0;
				for (int j = 0; j < tile_n[t]; j++)
					if (cell_chunks[c][tile_p0[t] + j] && !stored_big(c, tile_p0[t] + j)) {
						jp |= (uint64_t)j << (4 * nl++);
						if (store_mode != 1) m->second_pass_pairs += pass2_evaluates(c) ? cell_pairs[c][tile_p0[t] + j] : 0;
					}
				for (int j = 0; j < tile_n[t]; j++)
					if (stored_big(c, tile_p0[t] + j)) jp |= (uint64_t)j << (4 * (nl + ns++));
				tile_jpack[(size_t)c * n_tile + t] = jp;
				tile_nlist[(size_t)c * n_tile + t] = (uint32_t)nl;
				tile_nstored[(size_t)c * n_tile + t] = (uint32_t)ns;
				if (store_mode == 2) tile_k0[(size_t)c * n_tile + t] = (uint32_t)n_stored_c[c];   // first stored row of the tile
				n_stored_c[c] += ns;
			}
			k_first += k;
			for (int j = 0; j < tile_n[t]; j++)
				if (!cell_chunks[c][tile_p0[t] + j]) me[4 + k++] = (uint32_t)j << 24;
		}
	}
	if (cls_cnt.empty()) { cls_cnt.push_back(0); cls_cell.push_back(0); }

	// pass-1 work items.  One per classifier, except VALU-engine classifiers (more than 112 SNPs)
	// whose work dwarfs the typical one: a single wavefront per sample group would walk them for
	// many times the duration of the rest of the pass, so they are cut into items of typical size
	// that store per-cell sums, added in order afterwards (k_total_scan).
	std::vector<int> item, item_whole, split_row(std::max(C, 1), -1), split_cls, wide_cls;
	double split_heavy_ns = 0, split_rest_ns = 0;
	{
		// rough wavefront-time per record: matrix engine 50 ns at full occupancy, VALU engine 18 ns per
		// 32-bit word while other wavefronts share its SIMD (measured), 48 ns at full occupancy
		std::vector<double> work(C, 0.0);
		double typical = 0;
		int n_typ = 0;
		for (int c = 0; c < C; c++) {
			work[c] = (double)pairs[c] * (mfma_nkb[c] ? 50.0 * (0.5 + 0.5 * n_step[c]) : 48.0 * nwp[c]);
			if (mfma_nkb[c]) { typical += work[c]; n_typ++; }
			split_rest_ns += work[c];
		}
		typical = n_typ ? typical / n_typ : 0;
		std::vector<std::pair<double, std::vector<int>>> items, whole;
		for (int c = 0; c < C; c++) {
			if (n_step[c] > 1) { wide_cls.push_back(c); continue; }        // pass 1 in k_total_wide
			whole.push_back({work[c], {c, 0, cls_n[c], 0}});
			int nseg = 1;
			if (!mfma_nkb[c] && typical > 0 && work[c] > 3 * typical)
				nseg = (int)std::min<double>(64, std::max(2.0, std::floor(work[c] / typical)));
			if (nseg == 1 || cls_n[c] < 2) {
				items.push_back({work[c], {c, 0, cls_n[c], 0}});
				continue;
			}
			split_heavy_ns = std::max(split_heavy_ns, (double)pairs[c] * 18.0 * nwp[c]);       // measured: 1.1 ms for 5,050 pairs x 12 words
			split_row[c] = 1;                              // (>= 0: split; its cells have rows in HibagBatchView::cells)
			split_cls.push_back(c);
			uint64_t total = 0, acc = 0, chunk0 = 0;
			for (int i = 0; i < cls_n[c]; i++) total += cls_cnt[cls_off[c] + i] + 1;
			int i0 = 0, k = 1;
			for (int i = 0; i < cls_n[c]; i++) {
				acc += cls_cnt[cls_off[c] + i] + 1;
				const bool last = i + 1 == cls_n[c];
				if (last || acc * nseg >= total * k) {
					uint64_t chunks = 0;
					for (int j = i0; j <= i; j++) chunks += cls_cnt[cls_off[c] + j];
					items.push_back({work[c] * (double)(chunks + 1) / (double)total, {c, i0, i + 1, (int)chunk0}});
					chunk0 += chunks;
					i0 = i + 1;
					while (k < nseg && acc * nseg >= total * k) k++;
				}
			}
		}
		std::stable_sort(whole.begin(), whole.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
		for (const auto &it : whole) item_whole.insert(item_whole.end(), it.second.begin(), it.second.end());
		std::stable_sort(items.begin(), items.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
		for (const auto &it : items) item.insert(item.end(), it.second.begin(), it.second.end());
	}

	// pair lists of the matrix-core engine.  Pass 2 first, tile-major: the segments (tile t, classifier 0),
	// (t, 1), ... follow each other, which is the order a pass-2 wavefront reads them in; then, per
	// classifier, all cells back to back for pass 1 (no block left half empty at a tile boundary).
	std::vector<uint32_t> plist;
	struct SlotRange { size_t first, n; int c; };
	std::vector<SlotRange> slot_ranges;          // which classifier's haplotype table the slots of plist[first, first + n) index
	std::vector<uint64_t> blk_off(std::max(C, 1), 0), seg_off((size_t)std::max(C, 1) * n_tile, 0);
	std::vector<uint32_t> seg_nblk((size_t)std::max(C, 1) * n_tile, 0);
	long long dbg_b1 = 0, dbg_b2 = 0, dbg_seg = 0;
	std::vector<int> cell_row((size_t)C + 1, 0);
	for (int c = 0; c < C; c++)
		cell_row[c + 1] = cell_row[c] + (store_mode == 1 || split_row[c] >= 0 ? cls_n[c] : store_mode == 2 ? n_stored_c[c] : 0);
	// The E-stream of pass 2: per tile the blocks of classifier 0, 1, 2 ... (hibag_device.h).  A (classifier, tile) visit is
	// the blocks of its evaluated cells' pair slots -- one-step FP4 classifiers only -- with the visit's stored sums attached
	// four per block; a visit with more stored sums than its slot blocks carry (any classifier of another engine) gets
	// blocks of padding slots for the rest.
	std::vector<uint32_t> ehdr, etile_cstart((size_t)n_tile * (C + 1), 0);
	std::vector<uint64_t> etile_blk0(std::max(n_tile, 1), 0);
	// (an all-zero FP4 entry behind the tables: reads of the haplotype table through a slot of a padding block land here)
	hap.insert(hap.end(), HIBAG_ENGINE_HAP_DWORDS(HIBAG_ENGINE_FP4), 0u);
	while (hap.size() % 4) hap.push_back(0u);
	for (int t = 0; t < n_tile && store_mode != 1; t++) {
		etile_blk0[t] = plist.size() / HIBAG_PLIST_DWORDS;
		for (int c = 0; c < C; c++) {
			const size_t ct = (size_t)c * n_tile + t;
			etile_cstart[(size_t)t * (C + 1) + c] = (uint32_t)(plist.size() / HIBAG_PLIST_DWORDS - etile_blk0[t]);
			const size_t first = plist.size();
			int nb = 0;
			if (pass2_evaluates(c) && tile_nlist[ct] > 0)
				nb = append_pair_blocks(starts[c].data(), nh, tile_h1[t], tile_h2[t], tile_p0[t], tile_n[t],
					(uint32_t)m->cls[c].freq.size(), plist, nullptr, store_mode == 2 ? stored[c].data() : nullptr);
			if (nb > 0) slot_ranges.push_back({first, plist.size() - first, c});
			const int ns = (int)tile_nstored[ct];
			const int nvb = std::max(nb, (ns + HIBAG_STORED_PER_VISIT - 1) / HIBAG_STORED_PER_VISIT);
			for (int b = nb; b < nvb; b++) plist.insert(plist.end(), HIBAG_PLIST_DWORDS, 0u);     // padding slots: entry 0 of the zero entry's "table"
			dbg_b2 += nvb; dbg_seg += nvb > 0;
			// the visit's cells in closing order, then its stored ones (tile_jpack)
			uint64_t jp = tile_jpack[ct];
			uint64_t jps = jp >> (4 * tile_nlist[ct]);
			uint32_t srow = (uint32_t)cell_row[c] + tile_k0[ct];
			for (int b = 0; b < nvb; b++) {
				// the tile rows of the cells that close in this block, each at the place of its closing slot: slot i (odd: cells
				// are padded to an even number of slots) -> field i / 2
				uint64_t jq = 0;
				if (b < nb)
					for (int i = 0; i < HIBAG_PLIST_DWORDS; i++)
						if (plist[first + (size_t)b * HIBAG_PLIST_DWORDS + i] >> 31) {
							if (!(i & 1)) return hibag_fail(HIBAG_HIP_ESTATE, "internal: a cell of classifier %d closes at an even slot", c);
							jq |= (jp & 15u) << (4 * (i >> 1));
							jp >>= 4;
						}
				const int nsb = std::max(0, std::min(HIBAG_STORED_PER_VISIT, ns - HIBAG_STORED_PER_VISIT * b));
				if (c > 0xFFFF) return hibag_fail(HIBAG_HIP_EINVAL, "too many classifiers (%d) for the second pass's block headers", C);
				// (pass 2 requests the operand rows of every block it passes, also of blocks that only carry stored sums: a
				// classifier of the vector engine has no rows -- bt_row[c] is then the NEXT classifier's first row, or one past
				// the last row of the batch's array for the model's last classifiers: rows 0 and 1 instead)
				const uint32_t bt = (uint32_t)(HIBAG_ENGINE_ROWS(engine[c], n_snp_c[c]) > 0 ? bt_row[c] : 0);
				if (bt > 0xFFFFu) return hibag_fail(HIBAG_HIP_EINVAL, "too many classifiers for the matrix engine's operand rows");
				const uint32_t h[8] = {
					(uint32_t)c | (bt << 16), srow | ((uint32_t)nsb << 25),
					0u, 0u,                                   // (the next block's first two words: filled in below)
					(uint32_t)jq, (uint32_t)(jq >> 32),
					(uint32_t)jps, 0u};
				ehdr.insert(ehdr.end(), h, h + 8);
				jps >>= 4 * nsb;
				srow += (uint32_t)nsb;
			}
		}
		etile_cstart[(size_t)t * (C + 1) + C] = (uint32_t)(plist.size() / HIBAG_PLIST_DWORDS - etile_blk0[t]);
	}
	// look-ahead slack: the loop requests block b + 1 whole and the slots / header of block b + 2
	const uint64_t estream_blocks = plist.size() / HIBAG_PLIST_DWORDS + 4;
	plist.insert(plist.end(), 4 * HIBAG_PLIST_DWORDS, 0u);
	ehdr.resize(estream_blocks * 8, 0u);
	for (uint64_t b = 0; b + 1 < estream_blocks; b++) { ehdr[b * 8 + 2] = ehdr[(b + 1) * 8]; ehdr[b * 8 + 3] = ehdr[(b + 1) * 8 + 1]; }
	const uint64_t p1_base = plist.size();
	std::vector<uint32_t> blk_close;
	// segments of the classifiers with several K steps (k_total_wide): {classifier, first stored row, blocks} + list offset
	std::vector<int> wseg, wide_scan;                   // wide_scan: the classifiers of several K steps whose total k_total_scan forms
	std::vector<uint64_t> wseg_off;
	for (int c = 0; c < C; c++) {
		if (!mfma_nkb[c]) continue;
		blk_off[c] = plist.size();
		if (n_step[c] > 1) {
			// A classifier of several K steps: its list in segments of whole cells, each starting a block of its own, so
			// that different workgroups can walk them (their cell sums are stored, k_total_scan adds them in order);
			// walked as one list (majority vote) the padding between the segments adds nothing.
			// (pairs per segment: about what a typical one-step classifier of 5,000 pairs costs)
			// A model with many such classifiers has parallelism enough: then a classifier is ONE segment, its walk forms the
			// in-order total itself (wide_seg[3] = 1) and k_total_scan -- a second pass over every stored sum, HBM-bound --
			// is not needed for it.
			const bool whole = (int)wide_cls.size() >= 8;
			const long long seg_pairs = whole ? (1ll << 62) : std::max<long long>(512, 6000 / n_step[c]);
			bool any_seg = false;
			int p_lo = 0, h1_lo = 0, h2_lo = 0, row = 0, h1 = 0, h2 = 0;
			long long acc_pairs = 0;
			int rows_in_seg = 0;
			for (int p = 0; p < P; p++) {
				acc_pairs += cell_pairs[c][p];
				rows_in_seg += cell_pairs[c][p] != 0;
				int nh1 = h1, nh2 = h2 + 1;
				if (nh2 == nh) { nh1++; nh2 = nh1; }
				if (acc_pairs >= seg_pairs || p + 1 == P) {
					const size_t off = plist.size();
					const int nb = append_pair_blocks(starts[c].data(), nh, h1_lo, h2_lo, p_lo, p + 1 - p_lo, (uint32_t)m->cls[c].freq.size(),
						plist, stored[c].data(), nullptr);
					if (nb > 0) { wseg.insert(wseg.end(), {c, row, nb, whole ? 1 : 0}); wseg_off.push_back(off); any_seg = true; }
					row += rows_in_seg; rows_in_seg = 0; acc_pairs = 0;
					p_lo = p + 1; h1_lo = nh1; h2_lo = nh2;
				}
				h1 = nh1; h2 = nh2;
			}
			if (!whole || !any_seg) wide_scan.push_back(c);          // (a classifier without haplotypes has no segment: the scan writes its zero total)
			cls_nblk[c] = (int)((plist.size() - blk_off[c]) / HIBAG_PLIST_DWORDS);
			slot_ranges.push_back({(size_t)blk_off[c], plist.size() - (size_t)blk_off[c], c});
			dbg_b1 += cls_nblk[c];
			for (int b = 0; b < cls_nblk[c] && store_mode; b++) blk_close.push_back(0);     // (keeps the block numbering; not used for these)
			continue;
		}
		cls_nblk[c] = append_pair_blocks(starts[c].data(), nh, 0, 0, 0, P, (uint32_t)m->cls[c].freq.size(), plist,
			store_mode ? stored[c].data() : nullptr, nullptr);
		slot_ranges.push_back({(size_t)blk_off[c], plist.size() - (size_t)blk_off[c], c});
		dbg_b1 += cls_nblk[c];
		uint32_t closed = 0;
		for (int b = 0; b < cls_nblk[c] && store_mode; b++) {      // stored cells closed before block b
			blk_close.push_back(closed);
			for (int i = 0; i < HIBAG_PLIST_DWORDS; i++) closed += (plist[blk_off[c] + (size_t)b * HIBAG_PLIST_DWORDS + i] >> 30) & 1u;
		}
	}
	if (blk_close.empty()) blk_close.push_back(0);
	(void)valu_pairs;
	if (getenv("HIBAG_DEBUG_MODEL"))
		fprintf(stderr, "[hibag model] %d classifiers, %d tiles, pairs %lld; blocks of 32: pass 1 %lld, pass 2 %lld in %lld (classifier, tile) segments; "
			"pair lists %.1f MB + factors %.1f MB + block headers %.1f MB, haplotype table %.1f KB, VALU-engine stream %.1f MB\n",
			C, n_tile, (long long)m->pair_evals, dbg_b1, dbg_b2, dbg_seg, plist.size() * 4e-6, plist.size() * 8e-6, plist.size() / 32 * 16e-6,
			hap.size() * 4e-3, stream.size() * 4e-6);
	plist.insert(plist.end(), 4 * HIBAG_PLIST_DWORDS, 0u);   // look-ahead slack of the block walker
	// What the kernels take from a block through the SCALAR cache (hibag_device.h): the frequency factor of every slot --
	// ff[i1] * f[i2], the one rounded multiplication of src/LibHLA.cpp:1786-1813, made here once instead of by every wavefront
	// that walks the list -- and a header {cell ends, stored cell ends, slots worth evaluating}.
	std::vector<double> pfac(plist.size(), 0.0);
	for (const SlotRange &r : slot_ranges) {
		const std::vector<double> &freq = m->cls[r.c].freq;
		const uint32_t H = (uint32_t)freq.size();
		// table entries (above): [0, H) = {2 f, f}, H = the padding entry {0, 0}, H + 1 + i = {f, f} (first of a diagonal pair)
		auto ff_of = [&](uint32_t e) { return e < H ? 2 * freq[e] : e == H ? 0.0 : freq[e - H - 1]; };
		auto f_of = [&](uint32_t e) { return e < H ? freq[e] : e == H ? 0.0 : freq[e - H - 1]; };
		for (size_t i = r.first; i < r.first + r.n; i++) pfac[i] = ff_of(plist[i] & 0xFFFFu) * f_of((plist[i] >> 16) & 0x3FFFu);
	}
	std::vector<uint32_t> phdr(plist.size() / HIBAG_PLIST_DWORDS * 4, 0u);
	for (size_t b = 0; b < plist.size() / HIBAG_PLIST_DWORDS; b++) {
		uint32_t ends = 0, stores = 0, live = 0;
		for (int i = 0; i < HIBAG_PLIST_DWORDS; i++) {
			const uint32_t w = plist[b * HIBAG_PLIST_DWORDS + i];
			if (w & HIBAG_PLIST_END) ends |= 1u << i;
			if (w >= (HIBAG_PLIST_END | HIBAG_PLIST_STORE)) stores |= 1u << i;
			if (pfac[b * HIBAG_PLIST_DWORDS + i] != 0.0) live |= 1u << i;      // (a zero factor adds +0.0: skipping it is exact)
		}
		live |= ends;
		int n_valid = 0;
		while (n_valid < 32 && (live >> n_valid)) n_valid++;
		phdr[4 * b] = ends; phdr[4 * b + 1] = stores; phdr[4 * b + 2] = (uint32_t)n_valid;
	}
	// Pass 2 takes everything of a block from its E-stream header alone (hibag_device.h): the end mask goes where the block's
	// own request words were (word 0; they move to word 7 -- a walk needs them for its first block only, every other block
	// is requested through the words 2, 3 of the header before it), the groups of four records worth evaluating above the stored cells' rows.
	// Bit 29 of word 1: the record before the block's first one -- the last record gone through of the nearest block before it
	// that has any -- closed a cell, so the block's first product STARTS a sum (block_accumulate); a block that is passed over
	// (nobody uses its classifier) resets the sum instead, and a walk that begins at the block begins at zero: the bit is
	// right whichever blocks came before.
	uint32_t closed_before = 0;
	for (uint64_t b = 0; b < estream_blocks; b++) {
		const uint32_t groups = (phdr[4 * b + 2] + 3) / 4;
		ehdr[b * 8 + 7] = ehdr[b * 8];
		ehdr[b * 8] = phdr[4 * b];
		ehdr[b * 8 + 1] |= closed_before << 29;
		ehdr[b * 8 + 6] = (ehdr[b * 8 + 6] & 0x0FFFFFFFu) | (groups << 28);
		if (groups > 0) closed_before = (phdr[4 * b] >> (4 * groups - 1)) & 1u;
	}
	for (uint64_t b = 0; b + 1 < estream_blocks; b++) ehdr[b * 8 + 3] = ehdr[(b + 1) * 8 + 1];
	// Prebuilt A-operand rows (HibagModelView::parow): for every slot of a one-step FP4 classifier the element-wise sum of its
	// two haplotypes' images -- the "sum" images for the lower K half (lanes 0..31), the "pair" images for the upper one
	// (lanes 32..63); nibble sums never carry (codes 0..3 + 0..3).  Blocks outside a slot range (padding blocks) stay zero.
	const size_t n_blocks_all = plist.size() / HIBAG_PLIST_DWORDS;
	double pre_mb = 128;
	if (const char *e = getenv("HIBAG_PREBUILT_MB")) pre_mb = atof(e);
	bool p1_prebuilt = false;
	{
		size_t fp4_p1_blocks = 0;
		for (int c = 0; c < C; c++) if (pass2_evaluates(c)) fp4_p1_blocks += (size_t)cls_nblk[c];
		p1_prebuilt = fp4_p1_blocks > 0 && (double)(n_blocks_all) * 1024.0 <= pre_mb * 1e6;
	}
	const size_t parow_blocks = p1_prebuilt ? n_blocks_all : (size_t)estream_blocks;
	std::vector<uint32_t> parow(parow_blocks * 256, 0u);
	for (const SlotRange &r : slot_ranges) {
		if (!pass2_evaluates(r.c)) continue;
		const uint32_t *tab_c = hap.data() + hap_off[r.c];
		for (size_t i = r.first; i < r.first + r.n; i++) {
			const size_t b = i / HIBAG_PLIST_DWORDS, sl = i % HIBAG_PLIST_DWORDS;
			if (b >= parow_blocks) break;
			const uint32_t *e1 = tab_c + (size_t)(plist[i] & 0xFFFFu) * 12, *e2 = tab_c + (size_t)((plist[i] >> 16) & 0x3FFFu) * 12;
			for (int h = 0; h < 2; h++)
				for (int d = 0; d < 4; d++) parow[(b * 64 + (size_t)h * 32 + sl) * 4 + d] = e1[4 * h + d] + e2[4 * h + d];
		}
	}
	// per (classifier, tile) record of pass 2 (one s_load_dwordx8)
	std::vector<uint32_t> ctile((size_t)std::max(C, 1) * n_tile * 8 + 8, 0);
	for (int c = 0; c < C; c++)
		for (int t = 0; t < n_tile; t++) {
			uint32_t *r = &ctile[((size_t)c * n_tile + t) * 8];
			const uint32_t *me = &tile_meta[((size_t)c * n_tile + t) * HIBAG_TILE_META];
			const uint64_t off = seg_off[(size_t)c * n_tile + t];
			if (bt_row[c] > 0xFFFF) return hibag_fail(HIBAG_HIP_EINVAL, "too many classifiers for the matrix engine's operand rows");
			(void)me;
			const int k_last = n_snp_c[c] - HIBAG_FP4_STEP_SNPS * (n_step[c] - 1);        // SNPs of the last K step (all of them for one step)
			r[0] = (uint32_t)mfma_nkb[c] | ((uint32_t)k_last << 2 & 0xFCu) | (tile_nlist[(size_t)c * n_tile + t] << 8) |
			       ((uint32_t)(n_step[c] - 1) << 13) | ((uint32_t)bt_row[c] << 16);
			r[1] = hap_off[c];
			r[2] = (uint32_t)off; r[3] = (uint32_t)(off >> 32);
			r[4] = seg_nblk[(size_t)c * n_tile + t];
			// first stored row of the (classifier, tile) among all stored cells of the model
			const uint64_t row = (uint64_t)cell_row[c] + tile_k0[(size_t)c * n_tile + t];
			if (row >> 27) return hibag_fail(HIBAG_HIP_EINVAL, "the model has too many allele pairs to store their sums");
			r[5] = (uint32_t)row | (tile_nstored[(size_t)c * n_tile + t] << 27);
			r[6] = (uint32_t)tile_jpack[(size_t)c * n_tile + t]; r[7] = (uint32_t)(tile_jpack[(size_t)c * n_tile + t] >> 32);
		}

	// one int arena
	std::vector<int> arena;
	auto put = [&](const std::vector<int> &v) {
		size_t off = arena.size();
		arena.insert(arena.end(), v.begin(), v.end());
		if (v.empty()) arena.push_back(0);
		return off;
	};
	std::vector<int> hap_off_i(hap_off.begin(), hap_off.end());
	const size_t o_nsnp = put(n_snp_c), o_nwp = put(nwp), o_snpoff = put(snp_off), o_snpidx = put(snp_index),
		o_snpw = put(snp_weight), o_mrow = put(mask_row), o_order = put(c_order), o_tp0 = put(tile_p0), o_tn = put(tile_n),
		o_coff = put(cls_off), o_cn = put(cls_n), o_nkb = put(mfma_nkb), o_nstep = put(n_step), o_btrow = put(bt_row), o_nblk = put(cls_nblk), o_hapoff = put(hap_off_i),
		o_item = put(item), o_srow = put(split_row), o_scls = put(split_cls), o_itemw = put(item_whole), o_crow = put(cell_row),
		o_wide = put(wide_cls), o_wseg = put(wseg), o_wscan = put(wide_scan);

	if (int rc = m->d_int.reserve(arena.size() * sizeof(int))) return rc;
	if (int rc = m->d_stream.reserve(stream.size() * sizeof(uint32_t))) return rc;
	const size_t tb_off = 0, tb_meta = stream_off.size() * sizeof(uint64_t), tb_cnt = tb_meta + tile_meta.size() * sizeof(uint32_t),
		tb_cell = tb_cnt + cls_cnt.size() * sizeof(uint32_t),
		tb_boff = (tb_cell + cls_cell.size() * sizeof(uint32_t) + 7) & ~(size_t)7,
		tb_ctile = (tb_boff + blk_off.size() * sizeof(uint64_t) + 31) & ~(size_t)31,
		tb_hap = (tb_ctile + ctile.size() * sizeof(uint32_t) + 15) & ~(size_t)15,
		tb_ehdr = (tb_hap + hap.size() * sizeof(uint32_t) + 31) & ~(size_t)31,
		tb_ecst = tb_ehdr + ehdr.size() * sizeof(uint32_t),
		tb_eblk = (tb_ecst + std::max<size_t>(etile_cstart.size(), 1) * sizeof(uint32_t) + 7) & ~(size_t)7,
		tb_close = tb_eblk + etile_blk0.size() * sizeof(uint64_t),
		tb_wsoff = (tb_close + blk_close.size() * sizeof(uint32_t) + 7) & ~(size_t)7,
		tb_end = tb_wsoff + std::max<size_t>(wseg_off.size(), 1) * sizeof(uint64_t);
	if (int rc = m->d_tile.reserve(tb_end)) return rc;
	if (int rc = m->d_tab.reserve(sizeof(m->tab))) return rc;
	HIP_TRY(hipMemcpy(m->d_int.p, arena.data(), arena.size() * sizeof(int), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(m->d_stream.p, stream.data(), stream.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	char *tbase = m->d_tile.as<char>();
	HIP_TRY(hipMemcpy(tbase + tb_off, stream_off.data(), stream_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_meta, tile_meta.data(), tile_meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_cnt, cls_cnt.data(), cls_cnt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_cell, cls_cell.data(), cls_cell.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_boff, blk_off.data(), blk_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_ctile, ctile.data(), ctile.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_hap, hap.data(), hap.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_ehdr, ehdr.data(), ehdr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	if (!etile_cstart.empty())
		HIP_TRY(hipMemcpy(tbase + tb_ecst, etile_cstart.data(), etile_cstart.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_eblk, etile_blk0.data(), etile_blk0.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(tbase + tb_close, blk_close.data(), blk_close.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	if (!wseg_off.empty())
		HIP_TRY(hipMemcpy(tbase + tb_wsoff, wseg_off.data(), wseg_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	if (int rc = m->d_blk.reserve(plist.size() * sizeof(uint32_t))) return rc;
	HIP_TRY(hipMemcpy(m->d_blk.p, plist.data(), plist.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	if (int rc = m->d_pfac.reserve(pfac.size() * sizeof(double))) return rc;
	HIP_TRY(hipMemcpy(m->d_pfac.p, pfac.data(), pfac.size() * sizeof(double), hipMemcpyHostToDevice));
	if (int rc = m->d_phdr.reserve(phdr.size() * sizeof(uint32_t))) return rc;
	HIP_TRY(hipMemcpy(m->d_phdr.p, phdr.data(), phdr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	if (int rc = m->d_parow.reserve(std::max<size_t>(parow.size(), 256) * sizeof(uint32_t))) return rc;
	if (!parow.empty()) HIP_TRY(hipMemcpy(m->d_parow.p, parow.data(), parow.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(m->d_tab.p, m->tab, sizeof(m->tab), hipMemcpyHostToDevice));

	HibagModelView &V = m->view;
	const int *base = m->d_int.as<int>();
	V.n_hla = nh; V.n_classifier = C; V.n_snp = S; V.n_cell = P; V.mask_rows = rows; V.n_tile = n_tile;
	V.n_snp_c = base + o_nsnp; V.nwp = base + o_nwp; V.snp_off = base + o_snpoff;
	V.snp_index = base + o_snpidx; V.snp_weight = base + o_snpw; V.mask_row = base + o_mrow;
	V.c_order = base + o_order; V.tile_p0 = base + o_tp0; V.tile_n = base + o_tn;
	V.stream_off = (const uint64_t *)(tbase + tb_off);
	V.tile_meta = (const uint32_t *)(tbase + tb_meta);
	V.cls_cnt = (const uint32_t *)(tbase + tb_cnt);
	V.cls_cell = (const uint32_t *)(tbase + tb_cell);
	V.cls_off = base + o_coff; V.cls_n = base + o_cn;
	V.engine = base + o_nkb; V.n_step = base + o_nstep; V.bt_row = base + o_btrow; V.cls_nblk = base + o_nblk;
	V.hap_off = (const uint32_t *)(base + o_hapoff);
	V.n_item_split = (int)item.size() / 4; V.n_item_whole = (int)item_whole.size() / 4; V.n_split = (int)split_cls.size();
	V.item_split = base + o_item; V.item_whole = base + o_itemw; V.item = V.item_whole; V.n_item = V.n_item_whole;
	V.split_row = base + o_srow; V.split_cls = base + o_scls;
	V.all_fp4 = 1;
	for (int c = 0; c < C; c++)
		if (n_step[c] == 1 && !(engine[c] == HIBAG_ENGINE_FP4)) V.all_fp4 = 0;      // (classifiers of several K steps are not work items of k_total)
	V.n_wide = (int)wide_cls.size(); V.wide_cls = base + o_wide;
	V.n_valu = 0;
	for (int c = 0; c < C; c++) V.n_valu += engine[c] == HIBAG_ENGINE_VALU;
	V.n_wide_scan = (int)wide_scan.size(); V.wide_scan = base + o_wscan;
	V.n_wide_seg = (int)wseg.size() / 4; V.wide_seg = base + o_wseg; V.wide_seg_off = (const uint64_t *)(tbase + tb_wsoff);
	if (V.n_wide > 0 && !m->side.stream) {
		HIP_TRY(hipStreamCreateWithFlags(&m->side.stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&m->side.fork, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&m->side.join, hipEventDisableTiming));
	}
	V.split_heavy_ns = split_heavy_ns; V.split_rest_ns = split_rest_ns;
	V.blk_off = (const uint64_t *)(tbase + tb_boff);
	V.ctile = (const uint32_t *)(tbase + tb_ctile);
	V.hap = (const uint32_t *)(tbase + tb_hap);
	V.hap_dwords = (uint32_t)hap.size();
	V.ehdr = (const uint32_t *)(tbase + tb_ehdr);
	V.estream_blocks = estream_blocks;
	V.etile_cstart = (const uint32_t *)(tbase + tb_ecst);
	V.etile_blk0 = (const uint64_t *)(tbase + tb_eblk);
	V.blk_close = (const uint32_t *)(tbase + tb_close);
	V.p1_base = p1_base;
	V.p1_blocks = dbg_b1;
	V.cell_row = base + o_crow;
	V.store_cells = store_mode;
	hibag_query_slots(V.slots_total, &V.slots_accum);
	{
		// A chunk waits for the chunk before it, which was dispatched a whole round earlier; in the worst case the chunks of
		// an item run one after the other, so the wait is bounded by the item's own length.  One poll lasts ~1 us (s_sleep +
		// an L2 round trip), a 32-slot block ~1.5 us of elapsed time at full occupancy: 16 polls per block of the longest
		// item is an order of magnitude of slack on top of the fixed 2^19 (~0.5 s).
		long long longest = 0;
		for (int c = 0; c < C; c++) longest = std::max<long long>(longest, mfma_nkb[c] ? cls_nblk[c] : pairs[c] / 8);
		m->spin_limit = (uint32_t)std::min<long long>(0xFFFFFFF0ll, (1ll << 19) + 16 * longest);
	}
	m->cell_rows = cell_row[C];
	if (store_mode != 1 && (uint64_t)cell_row[C] >= (1ull << 23))      // (k_accum: a stored row's byte offset within a sample group in 32 bits)
		return hibag_fail(HIBAG_HIP_EINVAL, "the model stores too many cell sums per sample (%d) for the second pass", cell_row[C]);
	V.plist = m->d_blk.as<uint32_t>();
	V.pfac = m->d_pfac.as<double>();
	V.phdr = m->d_phdr.as<uint32_t>();
	V.plist_dwords = plist.size();
	V.parow = m->d_parow.as<uint4>();
	V.parow_blocks = parow_blocks;
	V.p1_prebuilt = p1_prebuilt ? 1 : 0;
	m->bt_rows = bt_rows;
	V.stream = m->d_stream.as<uint32_t>();
	V.tab = m->d_tab.as<double>();
	m->mask_rows = rows;
	m->stream_bytes = stream.size() * sizeof(uint32_t);
	m->engine_of.assign(engine.begin(), engine.begin() + C);
	m->steps_of.assign(n_step.begin(), n_step.begin() + C);
	m->finalized = true;
	return 0;
}

} // namespace hibag_detail

// ===========================================================================
// C ABI: the model

extern "C" {

hibag_hip_model *hibag_hip_model_new(int n_hla, int n_snp)
{
	if (n_hla <= 0 || n_hla > 46340 || n_snp < 0) {
		hibag_fail(HIBAG_HIP_EINVAL, "invalid model dimensions (n_hla=%d, n_snp=%d)", n_hla, n_snp);
		return nullptr;
	}
	hibag_hip_model *m = new (std::nothrow) hibag_hip_model;
	if (!m) { hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	m->device = hibag_selected_device();
	const char *engine = getenv("HIBAG_ENGINE");         // "valu": bit logic + popcount on the vector ALU for every classifier
	m->use_mfma = !(engine && strcmp(engine, "valu") == 0);
	m->use_fp4 = !(engine && strcmp(engine, "i8") == 0);
	m->n_hla = n_hla;
	m->n_snp = n_snp;
	build_table(m->tab);
	// k_total_wide clamps a distance at 65 (hibag_k_engine.h, table_value_wide): every entry from there on must be the exact
	// zero IEEE arithmetic makes of exp(65 log 1e-5) = 1e-325 -- a libm that returned a denormal there would be a different
	// table from the reference's, and the clamp would change results
	for (int i = 65; i < HIBAG_TAB_N; i++)
		if (m->tab[i] != 0.0) {
			delete m;
			hibag_fail(HIBAG_HIP_ESTATE, "this libm's exp() does not underflow to zero at exp(%d * log(1e-5)): the mutation table differs from the reference's", i);
			return nullptr;
		}
	return m;
}

int hibag_hip_model_add_classifier(hibag_hip_model *m, int n_snp_c, const int32_t *snpidx,
	int n_haplo, const double *freq, const int32_t *hla, const char *const *haplo)
{
	if (int rc = check_classifier_args(m, n_snp_c, snpidx, n_haplo, freq, hla)) return rc;
	if (n_snp_c > 0 && !snpidx) return hibag_fail(HIBAG_HIP_EINVAL, "snpidx is NULL");
	if (n_haplo > 0 && !haplo) return hibag_fail(HIBAG_HIP_EINVAL, "haplo is NULL");
	std::vector<uint64_t> bits((size_t)n_haplo * 2, 0);
	for (int i = 0; i < n_haplo; i++) {
		const char *s = haplo[i];
		const size_t len = s ? strlen(s) : 0;
		if (len > HIBAG_HIP_MAX_SNP_IN_CLASSIFIER)   // src/LibHLA.cpp:328-329
			return hibag_fail(HIBAG_HIP_EINVAL, "THaplotype::StrToHaplo, the input string is too long.");
		if ((int)len != n_snp_c)
			return hibag_fail(HIBAG_HIP_EINVAL, "haplotype %d has %zu alleles, expected %d", i, len, n_snp_c);
		for (size_t j = 0; j < len; j++) {
			if (s[j] == '1') bits[2 * (size_t)i + (j >> 6)] |= (uint64_t)1 << (j & 63);
			else if (s[j] != '0')                    // src/LibHLA.cpp:333-334
				return hibag_fail(HIBAG_HIP_EINVAL, "THaplotype::StrToHaplo, the input string should be '0' or '1'");
		}
	}
	push_classifier(m, n_snp_c, snpidx, n_haplo, freq, hla, std::move(bits));
	return 0;
}

int hibag_hip_model_add_classifier_packed(hibag_hip_model *m, int n_snp_c, const int32_t *snpidx,
	int n_haplo, const double *freq, const int32_t *hla, const uint64_t *bits_in)
{
	if (int rc = check_classifier_args(m, n_snp_c, snpidx, n_haplo, freq, hla)) return rc;
	if (n_haplo > 0 && !bits_in) return hibag_fail(HIBAG_HIP_EINVAL, "bits is NULL");
	// clear bits >= n_snp_c: the reference leaves them uninitialised (src/LibHLA.cpp:287-292)
	uint64_t mask[2];
	for (int w = 0; w < 2; w++) {
		const int lo = 64 * w;
		mask[w] = n_snp_c >= lo + 64 ? ~(uint64_t)0 : (n_snp_c <= lo ? 0 : (((uint64_t)1 << (n_snp_c - lo)) - 1));
	}
	std::vector<uint64_t> bits((size_t)n_haplo * 2);
	for (int i = 0; i < n_haplo; i++)
		for (int w = 0; w < 2; w++) bits[2 * (size_t)i + w] = bits_in[2 * (size_t)i + w] & mask[w];
	push_classifier(m, n_snp_c, snpidx, n_haplo, freq, hla, std::move(bits));
	return 0;
}

int hibag_hip_model_set_snp_weights(hibag_hip_model *m, const int32_t *snp_weight)
{
	if (!m || !snp_weight) return hibag_fail(HIBAG_HIP_EINVAL, "NULL argument");
	if (m->finalized) return hibag_fail(HIBAG_HIP_ESTATE, "model already finalized");
	m->snp_weight_override.assign(snp_weight, snp_weight + std::max(m->n_snp, 1));
	return 0;
}

int hibag_hip_model_finalize(hibag_hip_model *m)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	return finalize_model(m);
}

void hibag_hip_model_free(hibag_hip_model *m) { delete m; }

int hibag_hip_model_device(const hibag_hip_model *m) { return m ? m->device : -1; }
int hibag_hip_model_n_hla(const hibag_hip_model *m) { return m ? m->n_hla : 0; }
int hibag_hip_model_n_snp(const hibag_hip_model *m) { return m ? m->n_snp : 0; }
int hibag_hip_model_n_classifier(const hibag_hip_model *m) { return m ? (int)m->cls.size() : 0; }

int64_t hibag_hip_model_pair_evals(const hibag_hip_model *m)
{
	if (!m) return 0;
	int64_t n = 0;
	for (const auto &c : m->cls) n += (int64_t)c.freq.size() * ((int64_t)c.freq.size() + 1) / 2;
	return n;
}

int64_t hibag_hip_model_stored_cells(const hibag_hip_model *m)
{
	return m && m->finalized && m->store_mode ? (int64_t)m->cell_rows : 0;
}

int64_t hibag_hip_model_second_pass_pairs(const hibag_hip_model *m)
{
	return m && m->finalized ? m->second_pass_pairs : 0;
}

int hibag_hip_model_mutation_table(const hibag_hip_model *m, double *out)
{
	if (!m || !out) return hibag_fail(HIBAG_HIP_EINVAL, "NULL argument");
	memcpy(out, m->tab, sizeof(m->tab));
	return 0;
}

// ---- several devices ------------------------------------------------------------------------------------

hibag_hip_model *hibag_hip_model_replicate(const hibag_hip_model *src, int device)
{
	if (!src) { hibag_fail(HIBAG_HIP_EINVAL, "model is NULL"); return nullptr; }
	const int n = hibag_hip_device_count();
	if (device < 0 || device >= n) { hibag_fail(HIBAG_HIP_ENODEV, "HIP device %d not available (%d visible)", device, n); return nullptr; }
	hibag_hip_model *m = new (std::nothrow) hibag_hip_model;
	if (!m) { hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	m->device = device;
	m->n_hla = src->n_hla; m->n_snp = src->n_snp;
	m->have_snpidx = src->have_snpidx; m->use_mfma = src->use_mfma; m->use_fp4 = src->use_fp4;
	m->cls = src->cls;
	m->snp_weight_override = src->snp_weight_override;
	memcpy(m->tab, src->tab, sizeof(m->tab));
	if (src->finalized && hibag_hip_model_finalize(m)) { delete m; return nullptr; }
	return m;
}

// A shard of a model for classifier-sharded prediction (hibag_shard.hip): classifiers [first, first + count) of `src`, order
// kept, with the FULL model's per-SNP classifier counts (_GetSNPWeights, src/LibHLA.cpp:2484-2496), on `device`.
hibag_hip_model *hibag_hip_model_shard(const hibag_hip_model *src, int shard, int n_shards, int device)
{
	if (!src) { hibag_fail(HIBAG_HIP_EINVAL, "model is NULL"); return nullptr; }
	int first = 0, count = 0;
	if (hibag_hip_shard_bounds((int)src->cls.size(), n_shards, shard, &first, &count)) return nullptr;
	const int n = hibag_hip_device_count();
	if (device < 0 || device >= n) { hibag_fail(HIBAG_HIP_ENODEV, "HIP device %d not available (%d visible)", device, n); return nullptr; }
	hibag_hip_model *m = new (std::nothrow) hibag_hip_model;
	if (!m) { hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	try {
		m->device = device;
		m->n_hla = src->n_hla; m->n_snp = src->n_snp;
		m->have_snpidx = src->have_snpidx; m->use_mfma = src->use_mfma; m->use_fp4 = src->use_fp4;
		m->cls.assign(src->cls.begin() + first, src->cls.begin() + first + count);
		if (!src->snp_weight_override.empty()) m->snp_weight_override = src->snp_weight_override;     // (a shard of a shard keeps the full model's counts)
		else {
			m->snp_weight_override.assign(std::max(src->n_snp, 1), 0);
			for (const HostClassifier &k : src->cls)
				for (int v : k.snpidx) m->snp_weight_override[v]++;
		}
		memcpy(m->tab, src->tab, sizeof(m->tab));
	} catch (...) { delete m; hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	if (src->finalized && hibag_hip_model_finalize(m)) { delete m; return nullptr; }
	return m;
}

int hibag_hip_model_engine(const hibag_hip_model *m, int classifier, int *engine, int *k_steps)
{
	if (!m || !m->finalized) return hibag_fail(HIBAG_HIP_ESTATE, "model not finalized");
	if (classifier < 0 || classifier >= (int)m->cls.size()) return hibag_fail(HIBAG_HIP_EINVAL, "classifier %d out of range", classifier);
	if (engine) *engine = m->engine_of[classifier];
	if (k_steps) *k_steps = m->steps_of[classifier];
	return 0;
}

} // extern "C"
