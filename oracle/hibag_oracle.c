/*
 * hibag_oracle.c -- CPU restatement of HIBAG's attribute-bagging prediction
 * hot path.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is linked into,
 * imported by, or executed from the product (hibag_amd/, libhibag_hip.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * PARITY PIN.  The reference sources cannot be compiled in this image (every
 * translation unit of /root/reference/src includes <R.h>; R, RcppParallel and
 * TBB are absent, and stand-in headers are not allowed), so there is no
 * oracle/_ref build.  The restatement is instead pinned against outputs the
 * reference itself produced and stored in its own fixtures:
 *   - `outofbag.acc` of each of the 2 x 100 classifiers in
 *     inst/extdata/ModelList.RData and inst/extdata/OutOfBag.RData
 *     (= 0.5 * #correct alleles / #OOB from _BestGuess over the out-of-bag
 *     samples, src/LibHLA.cpp:1934-1955, 2121), and
 *   - the 34-value `matching` vector of inst/extdata/OutOfBag.RData
 *     (hlaPredict() output on the training samples, R/HIBAG.R:253-258).
 * tests/test_oracle_pin.py recomputes both from data/HapMap_CEU_Geno.rdata and
 * data/HLA_Type_Table.rdata (committed as fixtures under tests/golden/).
 *   - Beyond single values: hibag_oracle_train.c re-runs the training calls that
 *     produced those two files (set.seed(100); hlaAttrBagging(...)) and gets
 *     all 200 stored classifiers back bit for bit -- thousands of _BestGuess /
 *     _PostProb / _PrepHaploMatch evaluations per model feed every decision
 *     (tests/test_oracle_train.py).
 *   - oracle_conv_bed (below) decodes inst/extdata/HapMap_CEU.bed to the
 *     genotypes of data/HapMap_CEU_Geno.rdata (tests/test_bed_host.py).
 *
 * Every function cites the reference lines it follows.  Data layout is this
 * repo's own (flat SoA arrays), not the reference's classes.
 *
 * Bit conventions (inst/include/LibHLA_ext.h:240-255):
 *   haplotype: bit s of the 128-bit word = allele of SNP s (1 = A allele)
 *   genotype : two bit planes (S1,S2): g=0 -> (0,0), 1 -> (1,0), 2 -> (1,1),
 *              missing -> (0,1)
 */

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_SNP 128           /* LibHLA_ext.h:223 */
#define ORACLE_NA_INTEGER (-2147483647 - 1) /* R's NA_integer_ */
#define ORACLE_MIN_RARE_FREQ 1e-5    /* LibHLA_ext.h:230 */

static double TAB[2 * ORACLE_MAX_SNP + 1];
static int tab_ready = 0;

/* src/LibHLA.cpp:166-183: TAB[i] = exp(i*log(1e-5)), TAB[0] = 1, non-finite -> 0 */
static void build_tab(void)
{
	if (tab_ready) return;
	const int n = 2 * ORACLE_MAX_SNP;
	for (int i = 0; i <= n; i++)
		TAB[i] = exp(i * log(ORACLE_MIN_RARE_FREQ));
	TAB[0] = 1;
	for (int i = 0; i <= n; i++)
		if (!isfinite(TAB[i])) TAB[i] = 0;
	tab_ready = 1;
}

void oracle_mutation_table(double out[2 * ORACLE_MAX_SNP + 1])
{
	build_tab();
	memcpy(out, TAB, sizeof(TAB));
}

/* src/LibHLA.cpp:326-345 (StrToHaplo/_SetAllele).  Unlike the reference the
 * unused high bits are zeroed; they are masked by the genotype's missing flag
 * anyway (src/LibHLA.cpp:672-673). Returns 0, or -1 on a bad character. */
int oracle_haplo_from_string(const char *str, uint64_t bits[2])
{
	bits[0] = bits[1] = 0;
	size_t n = strlen(str);
	if (n > ORACLE_MAX_SNP) return -1;
	for (size_t i = 0; i < n; i++) {
		if (str[i] == '1') bits[i >> 6] |= (uint64_t)1 << (i & 63);
		else if (str[i] != '0') return -1;
	}
	return 0;
}

/* src/LibHLA.cpp:662-706 (TGenotype::IntToSNP): gather `length` genotypes of
 * one sample through `index`, anything outside 0..2 is missing; positions
 * >= length are missing too (S1=0,S2=1), so they never count. */
void oracle_int_to_snp(int length, const int *geno_base, const int *index,
	uint64_t s1[2], uint64_t s2[2])
{
	s1[0] = s1[1] = 0;
	s2[0] = s2[1] = ~(uint64_t)0;
	for (int i = 0; i < length; i++) {
		const int g = geno_base[index[i]];
		const uint64_t bit = (uint64_t)1 << (i & 63);
		const int w = i >> 6;
		if (g == 0)      { s2[w] &= ~bit; }
		else if (g == 1) { s1[w] |= bit; s2[w] &= ~bit; }
		else if (g == 2) { s1[w] |= bit; }
		/* else missing: (0,1) already */
	}
}

/* src/LibHLA.cpp:747-819 (hamm_d): words used = 1 if n_snp <= 64 else 2 */
static inline int hamm(int n_snp, const uint64_t s1[2], const uint64_t s2[2],
	const uint64_t h1[2], const uint64_t h2[2])
{
	const int nw = (n_snp <= 64) ? 1 : 2;
	int d = 0;
	for (int w = 0; w < nw; w++) {
		const uint64_t miss = s2[w] & ~s1[w];
		const uint64_t mask = ((h1[w] ^ s2[w]) | (h2[w] ^ s1[w])) & ~miss;
		d += __builtin_popcountll((h1[w] ^ s1[w]) & mask) +
		     __builtin_popcountll((h2[w] ^ s2[w]) & mask);
	}
	return d;
}

int oracle_hamm_d(int n_snp, const uint64_t s1[2], const uint64_t s2[2],
	const uint64_t h1[2], const uint64_t h2[2])
{
	return hamm(n_snp, s1, s2, h1, h2);
}

/* One classifier = haplotypes grouped by ascending allele:
 *   bits[2*i], bits[2*i+1] : packed haplotype i
 *   freq[i]                : its frequency
 *   len_per_hla[h]         : number of haplotypes of allele h
 * (CHaplotypeList, src/LibHLA.h:85-140)                                     */

/* Visit every allele pair h1<=h2 in posterior-vector order and hand the
 * strictly ordered cell sum to `emit`.  Loop nest and the rounding sequence
 * `sum += (2*f1*f2) * TAB[d]` (diagonal first term `(f1*f1) * TAB[d]`) follow
 * src/LibHLA.cpp:1776-1821 / 1648-1701 / 1717-1764 exactly.               */
#define FOR_EACH_CELL(BODY)                                                    \
	build_tab();                                                               \
	const uint64_t *b1 = bits;                                                 \
	const double *f1p = freq;                                                  \
	for (int h1 = 0; h1 < n_hla; h1++) {                                       \
		const int n1 = len_per_hla[h1];                                        \
		double cell = 0;                                                       \
		for (int a = 0; a < n1; a++) {                                         \
			const uint64_t *ha = b1 + 2 * a;                                   \
			cell += (f1p[a] * f1p[a]) * TAB[hamm(n_snp, s1, s2, ha, ha)];      \
			const double ff = 2 * f1p[a];                                      \
			for (int b = a + 1; b < n1; b++)                                   \
				cell += (ff * f1p[b]) * TAB[hamm(n_snp, s1, s2, ha, b1 + 2 * b)]; \
		}                                                                      \
		{ const int h2 = h1; BODY }                                            \
		const uint64_t *b2 = b1 + 2 * n1;                                      \
		const double *f2p = f1p + n1;                                          \
		for (int h2 = h1 + 1; h2 < n_hla; h2++) {                              \
			const int n2 = len_per_hla[h2];                                    \
			cell = 0;                                                          \
			for (int a = 0; a < n1; a++) {                                     \
				const uint64_t *ha = b1 + 2 * a;                               \
				const double ff = 2 * f1p[a];                                  \
				for (int b = 0; b < n2; b++)                                   \
					cell += (ff * f2p[b]) * TAB[hamm(n_snp, s1, s2, ha, b2 + 2 * b)]; \
			}                                                                  \
			{ BODY }                                                           \
			b2 += 2 * n2; f2p += n2;                                           \
		}                                                                      \
		b1 += 2 * n1; f1p += n1;                                               \
	}

/* src/LibHLA.cpp:1769-1830 (_PostProb2_def): fills prob[P] (P = n_hla(n_hla+1)/2),
 * normalises it by the in-order total and returns that total. */
double oracle_post_prob2(int n_hla, int n_snp, const int *len_per_hla,
	const uint64_t *bits, const double *freq,
	const uint64_t s1[2], const uint64_t s2[2], double *prob)
{
	double *p = prob;
	FOR_EACH_CELL({ (void)h2; *p++ = cell; })
	const size_t n = (size_t)n_hla * (n_hla + 1) / 2;
	double sum = 0;
	for (size_t i = 0; i < n; i++) sum += prob[i];
	const double ff = 1 / sum;
	for (size_t i = 0; i < n; i++) prob[i] *= ff;
	return sum;
}

/* src/LibHLA.cpp:1639-1704 (_BestGuess_def): running strict arg-max over cells */
void oracle_best_guess(int n_hla, int n_snp, const int *len_per_hla,
	const uint64_t *bits, const double *freq,
	const uint64_t s1[2], const uint64_t s2[2], int out_hla[2])
{
	double max = 0;
	out_hla[0] = out_hla[1] = ORACLE_NA_INTEGER;
	FOR_EACH_CELL({
		if (max < cell) { max = cell; out_hla[0] = h1; out_hla[1] = h2; }
	})
}

/* src/LibHLA.cpp:1706-1767 (_PostProb_def): cell(true pair) / in-order total */
double oracle_post_prob(int n_hla, int n_snp, const int *len_per_hla,
	const uint64_t *bits, const double *freq,
	const uint64_t s1[2], const uint64_t s2[2], int a1, int a2)
{
	if (a1 > a2) { int t = a1; a1 = a2; a2 = t; }
	const int want = a2 + a1 * (2 * n_hla - a1 - 1) / 2;
	int idx = 0;
	double sum = 0, hit = 0;
	FOR_EACH_CELL({
		(void)h2;
		if (idx == want) hit = cell;
		idx++; sum += cell;
	})
	return hit / sum;
}

/* src/LibHLA.cpp:1569-1637 (_PrepHaploMatch_def): for the haplotype ranges of
 * a sample's two true alleles, list the pairs with distance 0, or -- if there
 * is none -- all pairs at the minimum distance.  Ranges are [st1,st1+n1) and
 * [st2,st2+n2); when st1 == st2 only pairs a<=b are visited.  Writes pair
 * indices (absolute haplotype indices) to out_pairs[2*k], returns k.  When
 * out_pairs is NULL only counts. */
int oracle_prep_haplo_match(int n_snp, const uint64_t *bits,
	int st1, int n1, int st2, int n2,
	const uint64_t s1[2], const uint64_t s2[2], int *out_pairs)
{
	int min_d = n_snp * 4, k = 0;
	const int same = (st1 == st2);
	for (int pass = 0; pass < 2; pass++) {
		for (int a = 0; a < n1; a++) {
			for (int b = same ? a : 0; b < (same ? n1 : n2); b++) {
				const int d = hamm(n_snp, s1, s2, bits + 2 * (st1 + a), bits + 2 * (st2 + b));
				if (pass == 0) {
					if (d < min_d) min_d = d;
					if (d != 0) continue;
				} else if (d != min_d) continue;
				if (out_pairs) { out_pairs[2 * k] = st1 + a; out_pairs[2 * k + 1] = st2 + b; }
				k++;
			}
		}
		if (min_d == 0) break;  /* exact matches already listed in pass 0 */
	}
	return k;
}

/* CHLATypeList::Compare, src/LibHLA.cpp:912-924: number of correct alleles */
int oracle_compare_hla(int p1, int p2, int t1, int t2)
{
	int cnt = 0;
	if (p1 == t1 || p1 == t2) {
		cnt = 1;
		if (p1 == t1) t1 = -1; else t2 = -1;
	}
	if (p2 == t1 || p2 == t2) cnt++;
	return cnt;
}

/* ------------------------------------------------------------------------ */
/* Whole-model prediction.  The model is passed flat:
 *   n_snp_c[c], snp_off[c] -> snp_index[snp_off[c] .. +n_snp_c[c])  (0-based)
 *   hap_off[c]  -> bits[2*hap_off[c]..], freq[hap_off[c]..]
 *   len_per_hla[c*n_hla + h]
 */

/* _GetSNPWeights, src/LibHLA.cpp:2484-2496 */
void oracle_snp_weights(int n_classifier, int n_snp_total, const int *n_snp_c,
	const int *snp_off, const int *snp_index, int *out_weight)
{
	memset(out_weight, 0, sizeof(int) * (size_t)n_snp_total);
	for (int c = 0; c < n_classifier; c++)
		for (int i = 0; i < n_snp_c[c]; i++)
			out_weight[snp_index[snp_off[c] + i]]++;
}

/* Strict first-max scan, src/LibHLA.cpp:1549-1566 */
static void arg_max_pair(int n_hla, const double *v, int out[2])
{
	out[0] = out[1] = ORACLE_NA_INTEGER;
	double max = 0;
	for (int h1 = 0; h1 < n_hla; h1++)
		for (int h2 = h1; h2 < n_hla; h2++, v++)
			if (max < *v) { max = *v; out[0] = h1; out[1] = h2; }
}

/* One sample: CAttrBag_Model::_PredictHLA CPU branch, src/LibHLA.cpp:2414-2482.
 * `post` is scratch [P], `sum_post` receives the ensemble posterior [P]. */
static double predict_one(int n_hla, int n_classifier, const int *n_snp_c,
	const int *snp_off, const int *snp_index, const int *hap_off,
	const int *len_per_hla, const uint64_t *bits, const double *freq,
	const int *snp_weight, const int *geno, int vote_method,
	double *post, double *sum_post)
{
	const size_t P = (size_t)n_hla * (n_hla + 1) / 2;
	memset(sum_post, 0, sizeof(double) * P);
	double sum_weight = 0, sum_matching = 0, num_matching = 0;

	for (int c = 0; c < n_classifier; c++) {
		const int *idx = snp_index + snp_off[c];
		/* classifier weight from missingness, :2418-2431 */
		int nw = 0, tot = 0;
		for (int i = 0; i < n_snp_c[c]; i++) {
			const int k = idx[i];
			tot += snp_weight[k];
			if (0 <= geno[k] && geno[k] <= 2) nw += snp_weight[k];
		}
		const double w = (tot > 0) ? ((double)nw / tot) : 0;
		if (w <= 0) continue;                                   /* :2451 */

		uint64_t s1[2], s2[2];
		oracle_int_to_snp(n_snp_c[c], geno, idx, s1, s2);       /* :2453 */
		const double pm = oracle_post_prob2(n_hla, n_snp_c[c],
			len_per_hla + (size_t)c * n_hla, bits + 2 * (size_t)hap_off[c],
			freq + hap_off[c], s1, s2, post);                   /* :2456 */
		sum_matching += pm * w;                                 /* :2458-2459 */
		num_matching += w;

		if (vote_method == 1) {                                 /* :1497-1507 */
			for (size_t i = 0; i < P; i++) sum_post[i] += post[i] * w;
			sum_weight += w;
		} else {                                                /* :2465-2475 */
			int pd[2];
			arg_max_pair(n_hla, post, pd);
			if (pd[0] != ORACLE_NA_INTEGER && pd[1] != ORACLE_NA_INTEGER) {
				/* one-hot posterior added with weight 1.0: every other cell gets +0.0 */
				sum_post[pd[1] + pd[0] * (2 * n_hla - pd[0] - 1) / 2] += 1.0;
				sum_weight += 1.0;
			}
		}
	}
	if (sum_weight > 0) {                                       /* :1509-1518 */
		const double ff = 1.0 / sum_weight;
		for (size_t i = 0; i < P; i++) sum_post[i] *= ff;
	}
	return sum_matching / num_matching;                         /* :2480 */
}

/* CAttrBag_Model::PredictHLA, src/LibHLA.cpp:2317-2412.  genomat is
 * [n_samp][n_snp_total] sample-major.  Any output pointer may be NULL.
 * Returns 0, -1 for a bad vote_method. */
int oracle_predict(int n_hla, int n_classifier, int n_snp_total,
	const int *n_snp_c, const int *snp_off, const int *snp_index,
	const int *hap_off, const int *len_per_hla,
	const uint64_t *bits, const double *freq,
	const int *genomat, int n_samp, int vote_method,
	int *out_h1, int *out_h2, double *out_max_prob, double *out_matching,
	double *out_dosage, double *out_prob)
{
	if (vote_method < 1 || vote_method > 2) return -1;      /* :2321-2322 */
	const size_t P = (size_t)n_hla * (n_hla + 1) / 2;
	int *snp_weight = (int *)malloc(sizeof(int) * (size_t)(n_snp_total > 0 ? n_snp_total : 1));
	double *post = (double *)malloc(sizeof(double) * 2 * P);
	double *sum_post = post + P;
	oracle_snp_weights(n_classifier, n_snp_total, n_snp_c, snp_off, snp_index, snp_weight);

	for (int i = 0; i < n_samp; i++) {
		const double match = predict_one(n_hla, n_classifier, n_snp_c, snp_off,
			snp_index, hap_off, len_per_hla, bits, freq, snp_weight,
			genomat + (size_t)i * n_snp_total, vote_method, post, sum_post);
		int hla[2];
		arg_max_pair(n_hla, sum_post, hla);                  /* :2370 */
		if (out_h1 && out_h2) { out_h1[i] = hla[0]; out_h2[i] = hla[1]; }
		if (out_max_prob) {                                  /* :2376-2382 */
			if (hla[0] != ORACLE_NA_INTEGER && hla[1] != ORACLE_NA_INTEGER)
				out_max_prob[i] = sum_post[hla[1] + hla[0] * (2 * n_hla - hla[0] - 1) / 2];
			else
				out_max_prob[i] = 0;
		}
		if (out_matching) out_matching[i] = match;
		if (out_dosage) {                                    /* :2387-2402 */
			double *d = out_dosage + (size_t)i * n_hla;
			memset(d, 0, sizeof(double) * (size_t)n_hla);
			const double *s = sum_post;
			for (int h1 = 0; h1 < n_hla; h1++) {
				d[h1] += 2 * (*s++);
				for (int h2 = h1 + 1; h2 < n_hla; h2++) {
					const double v = *s++;
					d[h1] += v; d[h2] += v;
				}
			}
		}
		if (out_prob) memcpy(out_prob + (size_t)i * P, sum_post, sizeof(double) * P);
	}
	free(post);
	free(snp_weight);
	return 0;
}

/* ---------------------------------------------------------------------------
 * PLINK BED decoding: HIBAG_ConvBED (src/HIBAG.cpp:1094-1191) on a memory
 * image of the file (3-byte prefix included).  snp_flag[n_snp] selects the
 * n_save SNPs to keep; out is int[n_samp][n_save] (R's n_save x n_samp matrix).
 * Returns 0, -1 on a bad prefix, -2 when the image is too short.
 * Pinned by the reference's own pair of fixtures: inst/extdata/HapMap_CEU.bed
 * decoded with this function reproduces data/HapMap_CEU_Geno.rdata. */
int oracle_conv_bed(const unsigned char *image, size_t n_bytes, int n_samp, int n_snp,
	int n_save, const int *snp_flag, int *out)
{
	static const int cvt[4] = { 2, ORACLE_NA_INTEGER, 1, 0 };      /* :1135 */
	if (n_bytes < 3 || image[0] != 0x6C || image[1] != 0x1B) return -1;   /* :1112-1113 */
	const int mode = image[2];
	const int n_row = (mode == 0) ? n_samp : n_snp;               /* :1118-1131 */
	const int n_col = (mode == 0) ? n_snp : n_samp;
	const size_t stride = ((size_t)n_col + 3) / 4;
	if (n_bytes < 3 + stride * (size_t)n_row) return -2;
	int i_snp = 0;
	for (int i = 0; i < n_row; i++) {
		const unsigned char *src = image + 3 + stride * (size_t)i;
		if (mode == 0) {                                           /* row = individual, :1166-1174 */
			int *dst = out + (size_t)i * n_save;
			for (int j = 0; j < n_snp; j++)
				if (snp_flag[j]) *dst++ = cvt[(src[j >> 2] >> (2 * (j & 3))) & 3];
		} else if (snp_flag[i]) {                                  /* row = SNP, :1175-1186 */
			for (int j = 0; j < n_samp; j++)
				out[(size_t)j * n_save + i_snp] = cvt[(src[j >> 2] >> (2 * (j & 3))) & 3];
			i_snp++;
		}
	}
	return 0;
}
