/*
 * hibag_oracle_avx2.c -- the timed CPU baseline ("port" of the reference's
 * AVX2 kernel target): 4-wide inner loop over an SoA haplotype table with the
 * terms added one lane at a time so that the rounding sequence equals the
 * scalar oracle's (the reference does the same and notes why at
 * src/LibHLA_ext_avx2.cpp:226-227), plus a sample-parallel driver that mirrors
 * the reference's PARALLEL_FOR over samples (src/LibHLA.cpp:2362).
 *
 * TEST INFRASTRUCTURE ONLY -- see the header of hibag_oracle.c.  Results are
 * bit-identical to oracle_predict() (tests/test_oracle.py checks that).
 *
 * Follows: src/LibHLA_ext_avx2.cpp:188-278 (4-wide accumulate), :495-561
 * (_PostProb2_avx2), src/LibHLA.cpp:543-563 (SetHaploAux SoA copy),
 * src/LibHLA.cpp:2317-2482 (PredictHLA/_PredictHLA).
 */

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NA_INT (-2147483647 - 1)

void oracle_mutation_table(double out[257]);
void oracle_int_to_snp(int length, const int *geno_base, const int *index,
	uint64_t s1[2], uint64_t s2[2]);
void oracle_snp_weights(int n_classifier, int n_snp_total, const int *n_snp_c,
	const int *snp_off, const int *snp_index, int *out_weight);

typedef struct {
	int n_hla, n_classifier, n_snp_total;
	const int *n_snp_c, *snp_off, *snp_index, *hap_off, *len_per_hla;
	const uint64_t *w0, *w1;   /* SoA: word 0 / word 1 of every haplotype */
	const double *freq;
	const int *snp_weight;
	const double *tab;
	const int *genomat;
	int n_samp, vote_method;
	int *out_h1, *out_h2;
	double *out_max_prob, *out_matching, *out_dosage, *out_prob;
} job_t;

typedef struct { const job_t *job; int lo, hi; } slice_t;

__attribute__((target("avx2,popcnt")))
static inline __m256i popcnt_bytes(__m256i v, __m256i lut, __m256i low4)
{
	/* per-byte popcount by two nibble table look-ups (pshufb) */
	const __m256i lo = _mm256_and_si256(v, low4);
	const __m256i hi = _mm256_and_si256(_mm256_srli_epi32(v, 4), low4);
	return _mm256_add_epi8(_mm256_shuffle_epi8(lut, lo), _mm256_shuffle_epi8(lut, hi));
}

/* cell += (ff*f[b]) * TAB[d(a,b)] for b in [lo,hi) -- 4 at a time, in order */
__attribute__((target("avx2,popcnt")))
static inline double row_accumulate(double cell, const double *tab, int two_words,
	uint64_t a0, uint64_t a1, double ff,
	const uint64_t *w0, const uint64_t *w1, const double *freq, int lo, int hi,
	const uint64_t s1[2], const uint64_t s2[2])
{
	const __m256i lut = _mm256_setr_epi8(0,1,1,2,1,2,2,3,1,2,2,3,2,3,3,4,
	                                     0,1,1,2,1,2,2,3,1,2,2,3,2,3,3,4);
	const __m256i low4 = _mm256_set1_epi8(0x0F);
	const __m256i zero = _mm256_setzero_si256();
	int b = lo;
	if (hi - lo >= 4) {
		const __m256i S1a = _mm256_set1_epi64x((long long)s1[0]);
		const __m256i S2a = _mm256_set1_epi64x((long long)s2[0]);
		const __m256i NMa = _mm256_set1_epi64x((long long)~(s2[0] & ~s1[0]));
		const __m256i A0 = _mm256_set1_epi64x((long long)a0);
		const __m256i A0xS2 = _mm256_xor_si256(A0, S2a), A0xS1 = _mm256_xor_si256(A0, S1a);
		__m256i S1b = zero, S2b = zero, NMb = zero, A1xS2 = zero, A1xS1 = zero;
		if (two_words) {
			S1b = _mm256_set1_epi64x((long long)s1[1]);
			S2b = _mm256_set1_epi64x((long long)s2[1]);
			NMb = _mm256_set1_epi64x((long long)~(s2[1] & ~s1[1]));
			const __m256i A1 = _mm256_set1_epi64x((long long)a1);
			A1xS2 = _mm256_xor_si256(A1, S2b); A1xS1 = _mm256_xor_si256(A1, S1b);
		}
		const __m256d FF = _mm256_set1_pd(ff);
		for (; b + 4 <= hi; b += 4) {
			const __m256i B0 = _mm256_loadu_si256((const __m256i *)(w0 + b));
			__m256i mask = _mm256_and_si256(NMa, _mm256_or_si256(A0xS2, _mm256_xor_si256(B0, S1a)));
			__m256i cnt = _mm256_add_epi8(
				popcnt_bytes(_mm256_and_si256(A0xS1, mask), lut, low4),
				popcnt_bytes(_mm256_and_si256(_mm256_xor_si256(B0, S2a), mask), lut, low4));
			if (two_words) {
				const __m256i B1 = _mm256_loadu_si256((const __m256i *)(w1 + b));
				mask = _mm256_and_si256(NMb, _mm256_or_si256(A1xS2, _mm256_xor_si256(B1, S1b)));
				cnt = _mm256_add_epi8(cnt, _mm256_add_epi8(
					popcnt_bytes(_mm256_and_si256(A1xS1, mask), lut, low4),
					popcnt_bytes(_mm256_and_si256(_mm256_xor_si256(B1, S2b), mask), lut, low4)));
			}
			const __m256i d4 = _mm256_sad_epu8(cnt, zero);       /* 4 x 64-bit distances */
			const __m256d t = _mm256_i64gather_pd(tab, d4, 8);
			const __m256d term = _mm256_mul_pd(_mm256_mul_pd(FF, _mm256_loadu_pd(freq + b)), t);
			double lane[4];
			_mm256_storeu_pd(lane, term);
			cell += lane[0]; cell += lane[1]; cell += lane[2]; cell += lane[3];
		}
	}
	for (; b < hi; b++) {
		uint64_t m = ((a0 ^ s2[0]) | (w0[b] ^ s1[0])) & ~(s2[0] & ~s1[0]);
		int d = __builtin_popcountll((a0 ^ s1[0]) & m) + __builtin_popcountll((w0[b] ^ s2[0]) & m);
		if (two_words) {
			m = ((a1 ^ s2[1]) | (w1[b] ^ s1[1])) & ~(s2[1] & ~s1[1]);
			d += __builtin_popcountll((a1 ^ s1[1]) & m) + __builtin_popcountll((w1[b] ^ s2[1]) & m);
		}
		cell += (ff * freq[b]) * tab[d];
	}
	return cell;
}

__attribute__((target("avx2,popcnt")))
static double post_prob2_soa(const job_t *J, int c, const uint64_t s1[2],
	const uint64_t s2[2], double *prob)
{
	const int n_hla = J->n_hla, two = J->n_snp_c[c] > 64;
	const int *len = J->len_per_hla + (size_t)c * n_hla;
	const uint64_t *w0 = J->w0 + J->hap_off[c], *w1 = J->w1 + J->hap_off[c];
	const double *fq = J->freq + J->hap_off[c], *tab = J->tab;
	double *p = prob;
	int st1 = 0;
	for (int h1 = 0; h1 < n_hla; h1++) {
		const int n1 = len[h1];
		double cell = 0;
		for (int a = st1; a < st1 + n1; a++) {
			uint64_t m = ((w0[a] ^ s2[0]) | (w0[a] ^ s1[0])) & ~(s2[0] & ~s1[0]);
			int d = __builtin_popcountll((w0[a] ^ s1[0]) & m) + __builtin_popcountll((w0[a] ^ s2[0]) & m);
			if (two) {
				m = ((w1[a] ^ s2[1]) | (w1[a] ^ s1[1])) & ~(s2[1] & ~s1[1]);
				d += __builtin_popcountll((w1[a] ^ s1[1]) & m) + __builtin_popcountll((w1[a] ^ s2[1]) & m);
			}
			cell += (fq[a] * fq[a]) * tab[d];
			cell = row_accumulate(cell, tab, two, w0[a], w1[a], 2 * fq[a],
				w0, w1, fq, a + 1, st1 + n1, s1, s2);
		}
		*p++ = cell;
		int st2 = st1 + n1;
		for (int h2 = h1 + 1; h2 < n_hla; h2++) {
			const int n2 = len[h2];
			cell = 0;
			for (int a = st1; a < st1 + n1; a++)
				cell = row_accumulate(cell, tab, two, w0[a], w1[a], 2 * fq[a],
					w0, w1, fq, st2, st2 + n2, s1, s2);
			*p++ = cell;
			st2 += n2;
		}
		st1 += n1;
	}
	const size_t P = (size_t)n_hla * (n_hla + 1) / 2;
	double sum = 0;
	for (size_t i = 0; i < P; i++) sum += prob[i];
	const double ff = 1 / sum;
	for (size_t i = 0; i < P; i++) prob[i] *= ff;
	return sum;
}

static void first_max(int n_hla, const double *v, int out[2])
{
	out[0] = out[1] = NA_INT;
	double max = 0;
	for (int h1 = 0; h1 < n_hla; h1++)
		for (int h2 = h1; h2 < n_hla; h2++, v++)
			if (max < *v) { max = *v; out[0] = h1; out[1] = h2; }
}

static void *run_slice(void *arg)
{
	const slice_t *S = (const slice_t *)arg;
	const job_t *J = S->job;
	const int n_hla = J->n_hla;
	const size_t P = (size_t)n_hla * (n_hla + 1) / 2;
	double *post = (double *)malloc(sizeof(double) * 2 * P);
	double *acc = post + P;

	for (int i = S->lo; i < S->hi; i++) {
		const int *geno = J->genomat + (size_t)i * J->n_snp_total;
		memset(acc, 0, sizeof(double) * P);
		double sum_w = 0, sum_match = 0, num_match = 0;
		for (int c = 0; c < J->n_classifier; c++) {
			const int *idx = J->snp_index + J->snp_off[c];
			int nw = 0, tot = 0;
			for (int k = 0; k < J->n_snp_c[c]; k++) {
				tot += J->snp_weight[idx[k]];
				if (0 <= geno[idx[k]] && geno[idx[k]] <= 2) nw += J->snp_weight[idx[k]];
			}
			const double w = (tot > 0) ? ((double)nw / tot) : 0;
			if (w <= 0) continue;
			uint64_t s1[2], s2[2];
			oracle_int_to_snp(J->n_snp_c[c], geno, idx, s1, s2);
			const double pm = post_prob2_soa(J, c, s1, s2, post);
			sum_match += pm * w; num_match += w;
			if (J->vote_method == 1) {
				for (size_t q = 0; q < P; q++) acc[q] += post[q] * w;
				sum_w += w;
			} else {
				int pd[2];
				first_max(n_hla, post, pd);
				if (pd[0] != NA_INT && pd[1] != NA_INT) {
					acc[pd[1] + pd[0] * (2 * n_hla - pd[0] - 1) / 2] += 1.0;
					sum_w += 1.0;
				}
			}
		}
		if (sum_w > 0) {
			const double ff = 1.0 / sum_w;
			for (size_t q = 0; q < P; q++) acc[q] *= ff;
		}
		int hla[2];
		first_max(n_hla, acc, hla);
		if (J->out_h1 && J->out_h2) { J->out_h1[i] = hla[0]; J->out_h2[i] = hla[1]; }
		if (J->out_max_prob)
			J->out_max_prob[i] = (hla[0] != NA_INT && hla[1] != NA_INT)
				? acc[hla[1] + hla[0] * (2 * n_hla - hla[0] - 1) / 2] : 0;
		if (J->out_matching) J->out_matching[i] = sum_match / num_match;
		if (J->out_dosage) {
			double *d = J->out_dosage + (size_t)i * n_hla;
			memset(d, 0, sizeof(double) * (size_t)n_hla);
			const double *s = acc;
			for (int h1 = 0; h1 < n_hla; h1++) {
				d[h1] += 2 * (*s++);
				for (int h2 = h1 + 1; h2 < n_hla; h2++) { const double v = *s++; d[h1] += v; d[h2] += v; }
			}
		}
		if (J->out_prob) memcpy(J->out_prob + (size_t)i * P, acc, sizeof(double) * P);
	}
	free(post);
	return NULL;
}

int oracle_cpu_supports_avx2(void)
{
	__builtin_cpu_init();
	return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
}

/* Same contract as oracle_predict() plus a thread count; `bits` is the AoS
 * [sumH][2] table, converted here to the SoA copy the 4-wide loop reads.
 * Returns 0, -1 bad vote_method, -2 no AVX2 on this CPU. */
int oracle_predict_avx2_mt(int n_hla, int n_classifier, int n_snp_total,
	const int *n_snp_c, const int *snp_off, const int *snp_index,
	const int *hap_off, const int *len_per_hla,
	const uint64_t *bits, const double *freq,
	const int *genomat, int n_samp, int vote_method, int n_threads,
	int *out_h1, int *out_h2, double *out_max_prob, double *out_matching,
	double *out_dosage, double *out_prob)
{
	if (vote_method < 1 || vote_method > 2) return -1;
	if (!oracle_cpu_supports_avx2()) return -2;
	if (n_threads < 1) n_threads = 1;
	if (n_threads > n_samp) n_threads = n_samp > 0 ? n_samp : 1;

	size_t total_h = 0;
	for (int c = 0; c < n_classifier; c++) {
		size_t h = 0;
		for (int a = 0; a < n_hla; a++) h += (size_t)len_per_hla[(size_t)c * n_hla + a];
		if ((size_t)hap_off[c] + h > total_h) total_h = (size_t)hap_off[c] + h;
	}
	uint64_t *w0 = (uint64_t *)malloc(sizeof(uint64_t) * (2 * total_h + 8));
	uint64_t *w1 = w0 + total_h + 4;
	for (size_t i = 0; i < total_h; i++) { w0[i] = bits[2 * i]; w1[i] = bits[2 * i + 1]; }
	int *snp_weight = (int *)malloc(sizeof(int) * (size_t)(n_snp_total > 0 ? n_snp_total : 1));
	oracle_snp_weights(n_classifier, n_snp_total, n_snp_c, snp_off, snp_index, snp_weight);
	double tab[257];
	oracle_mutation_table(tab);

	job_t J = { n_hla, n_classifier, n_snp_total, n_snp_c, snp_off, snp_index, hap_off,
		len_per_hla, w0, w1, freq, snp_weight, tab, genomat, n_samp, vote_method,
		out_h1, out_h2, out_max_prob, out_matching, out_dosage, out_prob };

	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
	slice_t *sl = (slice_t *)malloc(sizeof(slice_t) * (size_t)n_threads);
	for (int t = 0; t < n_threads; t++) {
		sl[t].job = &J;
		sl[t].lo = (int)((long long)n_samp * t / n_threads);
		sl[t].hi = (int)((long long)n_samp * (t + 1) / n_threads);
		if (t > 0) pthread_create(&th[t], NULL, run_slice, &sl[t]);
	}
	run_slice(&sl[0]);
	for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
	free(sl); free(th); free(snp_weight); free(w0);
	return 0;
}
