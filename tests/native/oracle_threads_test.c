/* Stand-alone harness for the oracle's threaded AVX2 port (oracle/hibag_oracle_avx2.c), built together with the oracle's
 * sources under -fsanitize=thread (and address,undefined) by tools/run_sanitizers.sh -- the Python test-suite cannot run
 * under ThreadSanitizer (the interpreter is not instrumented and hangs).  A seeded random model (haplotypes grouped by
 * allele, classifiers of 5..40 SNPs over 64 SNPs) and cohort with missing genotypes: the threaded port at 1, 2, 3, 8 and 16
 * threads, both vote methods, must equal the scalar oracle bit for bit. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_predict(int n_hla, int n_classifier, int n_snp_total, const int *n_snp_c, const int *snp_off, const int *snp_index,
	const int *hap_off, const int *len_per_hla, const uint64_t *bits, const double *freq, const int *genomat, int n_samp, int vote_method,
	int *out_h1, int *out_h2, double *out_max_prob, double *out_matching, double *out_dosage, double *out_prob);
int oracle_predict_avx2_mt(int n_hla, int n_classifier, int n_snp_total, const int *n_snp_c, const int *snp_off, const int *snp_index,
	const int *hap_off, const int *len_per_hla, const uint64_t *bits, const double *freq, const int *genomat, int n_samp, int vote_method,
	int n_threads, int *out_h1, int *out_h2, double *out_max_prob, double *out_matching, double *out_dosage, double *out_prob);

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static double unif(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }

enum { NH = 6, NC = 12, NS = 64, NSAMP = 300, HAPS = 14 };

int main(void)
{
	static int n_snp_c[NC], snp_off[NC], snp_index[NC * 40], hap_off[NC + 1], len_per_hla[NC * NH];
	static uint64_t bits[NC * HAPS * 2];
	static double freq[NC * HAPS];
	static int geno[NSAMP * NS];
	int n_idx = 0, n_hap = 0;
	for (int c = 0; c < NC; c++) {
		n_snp_c[c] = 5 + (int)(rnd() % 36);
		snp_off[c] = n_idx;
		for (int s = 0; s < n_snp_c[c]; s++) snp_index[n_idx++] = (int)(rnd() % NS);
		hap_off[c] = n_hap;
		int left = HAPS;
		for (int h = 0; h < NH; h++) {                 /* haplotypes grouped by allele; some alleles have none */
			int k = h == NH - 1 ? left : (int)(rnd() % 4);
			if (k > left) k = left;
			len_per_hla[c * NH + h] = k;
			left -= k;
		}
		double tot = 0;
		for (int i = 0; i < HAPS; i++) {
			const uint64_t mask = n_snp_c[c] >= 64 ? ~0ull : ((1ull << n_snp_c[c]) - 1);
			bits[2 * (n_hap + i)] = rnd() & mask; bits[2 * (n_hap + i) + 1] = 0;
			freq[n_hap + i] = unif() + 1e-3; tot += freq[n_hap + i];
		}
		for (int i = 0; i < HAPS; i++) freq[n_hap + i] /= tot;
		n_hap += HAPS;
	}
	hap_off[NC] = n_hap;
	for (int i = 0; i < NSAMP * NS; i++) { const uint64_t r = rnd() % 100; geno[i] = r < 4 ? (int)0x80000000 : (int)(r % 3); }
	const size_t P = (size_t)NH * (NH + 1) / 2;
	int *h1 = malloc(sizeof(int) * NSAMP * 2), *g1 = malloc(sizeof(int) * NSAMP * 2);
	double *o = malloc(sizeof(double) * NSAMP * (2 + NH + P)), *q = malloc(sizeof(double) * NSAMP * (2 + NH + P));
	int bad = 0;
	for (int vote = 1; vote <= 2; vote++) {
		if (oracle_predict(NH, NC, NS, n_snp_c, snp_off, snp_index, hap_off, len_per_hla, bits, freq, geno, NSAMP, vote,
				h1, h1 + NSAMP, o, o + NSAMP, o + 2 * NSAMP, o + (2 + NH) * NSAMP)) { printf("scalar oracle failed\n"); return 1; }
		const int threads[] = {1, 2, 3, 8, 16};
		for (int t = 0; t < 5; t++) {
			memset(g1, 0xFF, sizeof(int) * NSAMP * 2); memset(q, 0xFF, sizeof(double) * NSAMP * (2 + NH + P));
			const int rc = oracle_predict_avx2_mt(NH, NC, NS, n_snp_c, snp_off, snp_index, hap_off, len_per_hla, bits, freq, geno, NSAMP, vote,
				threads[t], g1, g1 + NSAMP, q, q + NSAMP, q + 2 * NSAMP, q + (2 + NH) * NSAMP);
			if (rc == -2) { printf("oracle_threads_test: no AVX2 on this CPU, skipped\n"); return 0; }
			if (rc || memcmp(h1, g1, sizeof(int) * NSAMP * 2) || memcmp(o, q, sizeof(double) * NSAMP * (2 + NH + P))) {
				printf("oracle_threads_test: vote %d, %d threads: differs from the scalar oracle (rc %d)\n", vote, threads[t], rc);
				bad++;
			}
		}
	}
	free(h1); free(g1); free(o); free(q);
	if (!bad) printf("oracle_threads_test OK\n");
	return bad != 0;
}
