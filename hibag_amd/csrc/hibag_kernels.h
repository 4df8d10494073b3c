// hibag_kernels.h -- launchers of the gfx950 kernels (hibag_kernels.hip).
// All launchers enqueue on `st` and return; they never allocate or synchronise.
#ifndef HIBAG_KERNELS_H_
#define HIBAG_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hibag_device.h"

// d_col / d_flip (optional): the cohort's own matrix with `row_len` SNPs per sample, gathered and flipped on the device
void hibag_launch_pack(const HibagModelView &M, const HibagBatchView &B, const int32_t *d_geno, int row_len,
	const int32_t *d_col, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st);
// the same for a SNP-major matrix: int32 [rows][ld], row d_col[k] (nullptr: k) holds model SNP k
void hibag_launch_pack_rows(const HibagModelView &M, const HibagBatchView &B, const int32_t *d_geno, size_t ld,
	const int32_t *d_col, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st);
// PLINK BED sources: `d_bed` is the payload after the 3-byte prefix, rows of `stride` bytes
void hibag_launch_pack_bed(const HibagModelView &M, const HibagBatchView &B, const uint8_t *d_bed, int mode,
	size_t stride, int samp0, const int32_t *d_snp_row, const int32_t *d_flip, uint8_t *d_codes, hipStream_t st);
void hibag_launch_bed_geno(const uint8_t *d_bed, int mode, size_t stride, int n_samp, int n_save,
	const int32_t *d_sel, int32_t *d_geno, hipStream_t st);
// `side`: a second stream and two events of the caller's, for the kernel that runs beside pass 1 where the model has
// FP4 classifiers of several K steps (fork behind what is already on `st`, join before anything that follows)
struct HibagSideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
// vote: the call is a majority vote (vote_method = 2) -- pass 1 logs the records of the cell sums (HibagBatchView::vrec) for
// hibag_launch_vote, stores no cell sums for a second pass and cuts no work items
void hibag_launch_total(const HibagModelView &M, const HibagBatchView &B, hipStream_t st, const HibagSideStream &side, bool vote = false);
// resident workgroups of k_total<STORE, occupancy> -- [0] <false, 5>, [1] <false, 6>, [2] <true, 5>, [3] <true, 6> -- and of
// k_accum on the current device (0 = unknown)
void hibag_query_slots(int total[4], int *accum);
void hibag_launch_accum(const HibagModelView &M, const HibagBatchView &B, hipStream_t st);
void hibag_launch_vote(const HibagModelView &M, const HibagBatchView &B, int *d_best_cell, hipStream_t st);
void hibag_launch_scalars(const HibagModelView &M, const HibagBatchView &B, const int *d_best_cell, hipStream_t st);
void hibag_launch_finish(const HibagModelView &M, const HibagBatchView &B, double *d_part,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching,
	double *d_dosage, double *d_postprob, hipStream_t st);

#endif
