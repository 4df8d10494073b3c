/* A C99 client of libhibag_hip.so: builds a two-classifier model through the C ABI, predicts four
 * samples and prints every output as hex doubles / ints.  tests/test_c_client.py compiles it with gcc
 * (-std=c99 -pedantic: the header must be plain C), runs it on the GPU box and compares the printed
 * values with the Python binding's and the oracle's.  This is the shape of the call sequence a HIBAG
 * maintainer's C++ glue would make. */
#include <stdio.h>
#include <stdlib.h>
#include "hibag_hip.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, hibag_hip_last_error()); return 1; } } while (0)

int main(void)
{
	char info[256];
	CHECK(hibag_hip_set_kernel_target("hip", info, sizeof info));
	enum { N_HLA = 3, N_SNP = 6, N_SAMP = 4 };
	hibag_hip_model *m = hibag_hip_model_new(N_HLA, N_SNP);
	if (!m) { fprintf(stderr, "%s\n", hibag_hip_last_error()); return 1; }
	{
		const int32_t snpidx[3] = {0, 2, 5};
		const double freq[4] = {0.4, 0.1, 0.3, 0.2};
		const int32_t hla[4] = {0, 0, 1, 2};
		const char *haplo[4] = {"010", "110", "001", "111"};
		CHECK(hibag_hip_model_add_classifier(m, 3, snpidx, 4, freq, hla, haplo));
	}
	{
		const int32_t snpidx[4] = {1, 2, 3, 4};
		const double freq[3] = {0.5, 0.25, 0.25};
		const int32_t hla[3] = {0, 1, 2};
		const uint64_t bits[6] = {0x5, 0, 0xA, 0, 0xF, 0};     /* 1010, 0101, 1111 */
		CHECK(hibag_hip_model_add_classifier_packed(m, 4, snpidx, 3, freq, hla, bits));
	}
	CHECK(hibag_hip_model_finalize(m));
	const int32_t geno[N_SAMP][N_SNP] = {
		{0, 1, 2, 1, 0, 1}, {2, 2, 2, 2, 2, 2}, {1, HIBAG_HIP_NA_INTEGER, 1, 0, 1, 0}, {0, 0, 0, 0, 0, 0}};
	int32_t h1[N_SAMP], h2[N_SAMP];
	double prob[N_SAMP], matching[N_SAMP], dosage[N_SAMP][N_HLA], post[N_SAMP][N_HLA * (N_HLA + 1) / 2];
	CHECK(hibag_hip_predict(m, &geno[0][0], N_SAMP, 1, h1, h2, prob, matching, &dosage[0][0], &post[0][0]));
	printf("n_hla %d n_snp %d n_classifier %d pair_evals %lld\n", hibag_hip_model_n_hla(m), hibag_hip_model_n_snp(m),
		hibag_hip_model_n_classifier(m), (long long)hibag_hip_model_pair_evals(m));
	for (int i = 0; i < N_SAMP; i++) {
		printf("%d %d %a %a", (int)h1[i], (int)h2[i], prob[i], matching[i]);
		for (int k = 0; k < N_HLA; k++) printf(" %a", dosage[i][k]);
		for (int k = 0; k < N_HLA * (N_HLA + 1) / 2; k++) printf(" %a", post[i][k]);
		printf("\n");
	}
	if (hibag_hip_predict(m, &geno[0][0], N_SAMP, 3, h1, h2, prob, matching, NULL, NULL) == 0) return 2;
	printf("error: %s\n", hibag_hip_last_error());
	hibag_hip_model_free(m);
	return 0;
}
