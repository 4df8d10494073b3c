// hibag_device.h -- device-side views shared by the kernels and the host code
// of libhibag_hip.so.  gfx950 only.
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//
//  Model (read-only, a few hundred KB, uniform across a wavefront -> fetched
//  with scalar loads):
//    hbits  : per classifier c, NW_c rows of H_c uint32: row w holds bits
//             32w..32w+31 of every haplotype (SoA; the reference's AVX path
//             keeps the same idea with 64-bit words, src/LibHLA.cpp:543-563)
//    hfreq  : per classifier H_c doubles
//    hla_start[c][0..n_hla] : prefix sums of LenPerHLA (src/LibHLA.h:85-140)
//
//  Per batch of samples ("lane = sample": consecutive samples are consecutive
//  addresses, so every wave access below is one coalesced row segment):
//    planes : uint32 [geno_rows][n_pad]; classifier c owns rows
//             geno_row[c] + 2*w + {0: S1, 1: S2}  (TGenotype bit planes,
//             inst/include/LibHLA_ext.h:245-255)
//    cw, tot, inv : double [C][n_pad]  classifier weight, in-order posterior
//             total, 1/total
//    part   : double [P+3][n_pad]  ensemble sums per allele pair + 3 scalars
#ifndef HIBAG_DEVICE_H_
#define HIBAG_DEVICE_H_

#include <stdint.h>

#define HIBAG_WAVE 64
#define HIBAG_TAB_N 257          // 2*128 + 1 distances (src/LibHLA.cpp:167)
#define HIBAG_MAX_WORDS 4        // 128 SNPs / 32

struct HibagModelView {
	int n_hla;
	int n_classifier;
	int n_snp;          // SNPs in the model (row length of the raw genotype matrix)
	int n_cell;         // n_hla*(n_hla+1)/2
	int geno_rows;      // sum_c 2*NW_c
	int n_tile;         // cell tiles for the accumulate pass
	int tile_cells;     // cells per tile (compile-time T of the kernel)

	const int *n_snp_c;       // [C]
	const int *n_word;        // [C] 32-bit words per bit plane = ceil(n_snp_c/32)
	const int *snp_off;       // [C]
	const int *snp_index;     // concatenated 0-based SNP indices
	const int *snp_weight;    // [n_snp] #classifiers using the SNP (src/LibHLA.cpp:2484-2496)
	const int *n_hap;         // [C]
	const int *hap_off;       // [C] into hfreq
	const int *bits_off;      // [C] into hbits (uint32 units)
	const int *geno_row;      // [C] first plane row of the classifier
	const int *hla_start;     // [C][n_hla+1]
	const int *c_order;       // [C] classifiers sorted by pair count, heaviest first
	const uint32_t *hbits;
	const double *hfreq;
	const double *tab;        // [257] exp(d*log(1e-5))

	const int *tile_cell;     // [n_tile][tile_cells] posterior index p, -1 = padding
	const int *cell_h1;       // [n_cell]
	const int *cell_h2;       // [n_cell]
};

struct HibagBatchView {
	int n_samp;         // samples in this batch
	int n_pad;          // rounded up to a multiple of 64
	uint32_t *planes;   // [geno_rows][n_pad]
	double *cw;         // [C][n_pad]
	double *tot;        // [C][n_pad]
	double *inv;        // [C][n_pad]
	double *part;       // [P+3][n_pad]
};

#endif
