// hibag_em.h -- the device EM fit of a growth step's candidate SNPs (hibag_em.hip), called by the training driver
// (hibag_train.hip) in place of one host thread per candidate.
#ifndef HIBAG_EM_H_
#define HIBAG_EM_H_

#include <stdint.h>

// The in-bag samples' haplotype pairs after CAlg_EM::PrepareHaplotypes (src/LibHLA.cpp:1002-1125): indices into the DOUBLED
// haplotype list (entry 2 i: allele 0 of the new SNP, 2 i + 1: allele 1), sample i owning pairs [off[i], off[i + 1]).
struct HibagEmPairs {
	int n_ib, n_pair, n_hap, n_samp_total;
	const int *h1, *h2;          // [n_pair]
	const int *off;              // [n_ib + 1]
	const int *boot;             // [n_ib] bootstrap count of each in-bag sample
	const int *hoff;             // [n_hap + 1] transposed: the pairs that contain haplotype h ...
	const int *hent;             // [2 n_pair]  ... in pair order, a homozygous pair (h, h) twice
	const double *cur_freq;      // [n_hap / 2] frequencies of the CURRENT haplotypes (before doubling)
};

// Fits n_cand candidate SNPs on the calling thread's current device (src/LibHLA.cpp:1127-1255 per candidate).
//   geno[c][n_ib]   genotype of the in-bag samples at candidate c: 0, 1, 2, or 3 = missing
//   afreq[c]        its allele frequency in the bag (not 0 or 1: the caller skips monomorphic SNPs like the reference, :1140-1143)
//   out_freq[c][n_hap]   the fitted frequencies of the doubled haplotypes
//   status[c]       1 = fitted, bit-identical to the host's fit; 2 = the stopping test came too close to its tolerance for
//                   the device's log() to decide it the way the host's would: the caller fits this candidate on the host
//   iters[c]        (optional) iterations run
// Throws `const char *` on a HIP error (like the build entries).  The step must satisfy hibag_em_fits() (the pair set and a
// candidate's state live in LDS); larger steps are the host's.
void hibag_em_fit_batch(const HibagEmPairs &P, const int8_t *const geno[], const double afreq[], int n_cand, double *out_freq, int *status, int *iters);
bool hibag_em_fits(int n_ib, int n_pair, int n_hap);
void hibag_em_release();

#endif
