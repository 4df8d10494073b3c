/*
 * hibag_hip.h -- C ABI of libhibag_hip.so, the MI355X (gfx950) implementation
 * of HIBAG's attribute-bagging prediction hot path.
 *
 * Plain C: pointers, sizes, int/double only.  No exception crosses this
 * boundary: every entry returns 0 on success or a negative HIBAG_HIP_E* code,
 * and hibag_hip_last_error() returns the message of the calling thread's last
 * failure (the reference throws ErrHLA and turns it into Rf_error,
 * src/HIBAG.cpp:41-60; a binding re-raises from the code + message).
 *
 * Each entry names the reference interface it replaces.  Paths are relative to
 * the HIBAG source tree (zhengxwen/HIBAG, package 1.47.3, kernel 1.5).
 */
#ifndef HIBAG_HIP_H_
#define HIBAG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIBAG_HIP_ABI_VERSION 7   /* 2: + PLINK BED entries, training driver; 3: + hibag_hip_predict_mapped[_device]; 4: + hibag_hip_model_stored_cells;
                                     5: + hibag_hip_model_status / _clear_status, hibag_hip_predict_multi, hibag_hip_model_replicate, hibag_hip_model_engine;
                                     7: + hibag_hip_predict_snp_major[_device], hibag_hip_trainer_set_shared, hibag_hip_train_set_thread_budget */

/* error codes */
#define HIBAG_HIP_OK          0
#define HIBAG_HIP_EINVAL     (-1)  /* bad argument (message says which)            */
#define HIBAG_HIP_ENODEV     (-2)  /* no usable HIP device / HIP runtime error     */
#define HIBAG_HIP_ENOMEM     (-3)  /* host or device allocation failed             */
#define HIBAG_HIP_ESTATE     (-4)  /* call order violated (e.g. predict before finalize) */
#define HIBAG_HIP_EHANDOVER  (-5)  /* a launch could not vouch for its sums (see "launch status" below); outputs were poisoned */

/* R's NA_integer_: what H1/H2 hold when no allele pair has positive
 * probability (src/LibHLA.cpp:1552), and the usual missing-genotype code. */
#define HIBAG_HIP_NA_INTEGER (-2147483647 - 1)

/* limits fixed by the reference's packed types */
#define HIBAG_HIP_MAX_SNP_IN_CLASSIFIER 128   /* inst/include/LibHLA_ext.h:223 */

typedef struct hibag_hip_model hibag_hip_model;  /* opaque handle */

/* ---- library / device ---------------------------------------------------- */

int hibag_hip_abi_version(void);

/* Message of this thread's most recent failing call ("" if none). */
const char *hibag_hip_last_error(void);

/* Number of HIP devices visible to the process (0 if none / no driver). */
int hibag_hip_device_count(void);

/* Device used by models created afterwards on this thread (default 0). */
int hibag_hip_set_device(int device);
/* The calling thread's selection (what a host passes on to the worker threads it starts: the selection is per thread). */
int hibag_hip_get_device(void);

/* Kernel-target selection -- the extension of hlaSetKernelTarget()
 * (R/HIBAG.R:1668-1674 -> HIBAG_Kernel_SetTarget, src/HIBAG.cpp:1430-1435 ->
 * CAlg_Prediction::Init_Target_IFunc, src/LibHLA.cpp:1266-1475).  The
 * reference accepts CPU names only; this library accepts exactly "hip" and
 * fails (EINVAL) for anything else, and fails (ENODEV) when no gfx950 device is
 * present -- there is no CPU fallback in this library.  On success writes a
 * description such as "HIP, gfx950, AMD Instinct MI355X, 256 CUs" to `info`. */
int hibag_hip_set_kernel_target(const char *target, char *info, size_t info_len);

/* ---- model construction: replaces HIBAG_New + HIBAG_NewClassifierHaplo --- */

/* HIBAG_New(n.samp, n.snp, n.hla) (src/HIBAG.cpp:486-503).  n_samp is only
 * book-keeping in the reference (bootstrap counts) and is not needed here. */
hibag_hip_model *hibag_hip_model_new(int n_hla, int n_snp);

/* HIBAG_NewClassifierHaplo(model, snpidx-1, samp.num, freq, hla-1, haplo, acc)
 * (src/HIBAG.cpp:817-841 -> CAttrBag_Classifier::Assign, src/LibHLA.cpp:2142-2165).
 *   snpidx[n_snp_c]  0-based indices into the model's SNP list
 *   freq[n_haplo], hla[n_haplo] (0-based, ascending -- haplotypes are grouped
 *   by allele, src/HIBAG.cpp:915-924), haplo[n_haplo] strings of '0'/'1' of
 *   length n_snp_c (char s <-> SNP snpidx[s]).
 * Errors like the reference: more than 128 SNPs, characters other than 0/1. */
int hibag_hip_model_add_classifier(hibag_hip_model *m, int n_snp_c,
	const int32_t *snpidx, int n_haplo, const double *freq,
	const int32_t *hla, const char *const *haplo);

/* Same, from packed haplotypes: bits[2*i], bits[2*i+1] are the two 64-bit words
 * of THaplotype::PackedHaplo (inst/include/LibHLA_ext.h:261-299); bits at
 * positions >= n_snp_c are ignored. */
int hibag_hip_model_add_classifier_packed(hibag_hip_model *m, int n_snp_c,
	const int32_t *snpidx, int n_haplo, const double *freq,
	const int32_t *hla, const uint64_t *bits);

/* Build the device tables (haplotype SoA, cell schedule, mutation table) and
 * upload them.  Must be called once after the last add_classifier. */
int hibag_hip_model_finalize(hibag_hip_model *m);

void hibag_hip_model_free(hibag_hip_model *m);

/* queries */
int hibag_hip_model_device(const hibag_hip_model *m);      /* the HIP device the model lives on (-1 for NULL) */
int hibag_hip_model_n_hla(const hibag_hip_model *m);
int hibag_hip_model_n_snp(const hibag_hip_model *m);
int hibag_hip_model_n_classifier(const hibag_hip_model *m);
/* sum over classifiers of H_c(H_c+1)/2: haplotype-pair evaluations per sample */
int64_t hibag_hip_model_pair_evals(const hibag_hip_model *m);
/* How a finalized model runs its second pass: pass 1 stores `stored_cells` cell sums per sample
 * (8 bytes each: a classifier's sum over the haplotype pairs of one allele pair) that pass 2 reads back, and pass 2
 * evaluates `second_pass_pairs` haplotype pairs per sample again.  All cells / no pairs for models with many pairs per
 * cell; otherwise the cells with many pairs are stored and the pairs of the others evaluated. */
int64_t hibag_hip_model_stored_cells(const hibag_hip_model *m);
int64_t hibag_hip_model_second_pass_pairs(const hibag_hip_model *m);
/* the 257-entry mutation/error table the device uses, exp(d*log(1e-5))
 * (src/LibHLA.cpp:166-183); out[257] */
int hibag_hip_model_mutation_table(const hibag_hip_model *m, double *out);

/* ---- prediction: replaces CAttrBag_Model::PredictHLA -------------------- */

/* CAttrBag_Model::PredictHLA (src/LibHLA.cpp:2317-2412) as reached from
 * HIBAG_Predict_Resp / _Dosage / _Resp_Prob (src/HIBAG.cpp:649-803).
 *   geno        int32 [n_samp][n_snp], sample-major (the memory of the R SNP x
 *               sample matrix); values outside 0..2 (incl. NA_integer_) = missing
 *   vote_method 1 = average posteriors, 2 = majority vote; else EINVAL with the
 *               reference's message "Invalid 'vote_method'."
 * Outputs (any may be NULL; H1 and H2 only together):
 *   H1,H2[n_samp]        0-based allele indices or NA_integer_
 *   max_prob[n_samp]     posterior of the called pair (0 if NA)
 *   matching[n_samp]     weighted mean of the pre-normalisation totals
 *   dosage[n_samp][n_hla]
 *   postprob[n_samp][n_hla(n_hla+1)/2]   pair order h1<=h2, h2 fastest
 * Host-pointer form: copies geno to the device, runs, copies results back. */
int hibag_hip_predict(hibag_hip_model *m, const int32_t *geno, int n_samp,
	int vote_method, int32_t *H1, int32_t *H2, double *max_prob,
	double *matching, double *dosage, double *postprob);

/* Device-pointer form of the same call: every pointer is device memory on the
 * model's device, work is enqueued on `stream` (a hipStream_t, NULL = default
 * stream) and the call returns without synchronising.  A return of 0 means
 * "enqueued": whether the launch could vouch for its sums is the model's status,
 * see "launch status" below (outputs are NA / NaN if it could not). */
int hibag_hip_predict_device(hibag_hip_model *m, const int32_t *d_geno, int n_samp,
	int vote_method, int32_t *d_H1, int32_t *d_H2, double *d_max_prob,
	double *d_matching, double *d_dosage, double *d_postprob, void *stream);

/* hlaPredict()'s SNP selection and strand / allele-order fix-up (R/HIBAG.R:640-676: the rows of the
 * cohort's matrix picked by match(), absent SNPs as NA rows, hlaGenoSwitchStrand's g -> 2 - g) done on the
 * device while the genotypes are packed, instead of building a second matrix on the host:
 *   geno        int32 [n_samp][n_geno_snp]: the COHORT's matrix, its own SNPs in its own order
 *   snp_col[model n_snp]  0-based column of each model SNP in geno, -1 = the cohort lacks it
 *   flip[model n_snp]     NULL, or != 0 where the allele count must be reversed
 * Otherwise as hibag_hip_predict / hibag_hip_predict_device (the _device form takes every pointer,
 * snp_col and flip included, in device memory). */
int hibag_hip_predict_mapped(hibag_hip_model *m, const int32_t *geno, int n_samp, int n_geno_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob);
int hibag_hip_predict_mapped_device(hibag_hip_model *m, const int32_t *d_geno, int n_samp, int n_geno_snp,
	const int32_t *d_snp_col, const int32_t *d_flip, int vote_method,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, void *stream);

/* The same for a cohort matrix stored SNP-MAJOR -- geno[row * ld + sample], one row of n_samp genotypes per SNP, `ld` >= n_samp
 * elements from one row to the next: a C / numpy [snp][sample] array, i.e. the TRANSPOSE of the memory R hands
 * HIBAG_Predict_* (R/HIBAG.R:715-725), which a host that is not R would otherwise have to transpose first.
 *   snp_col[model n_snp]  row of each model SNP in geno, -1 = the cohort lacks it; NULL = row k holds model SNP k
 *   flip[model n_snp]     NULL, or != 0 where the allele count must be reversed
 * The host-pointer form uploads only the rows the model uses (one block of the caller's matrix when they are consecutive,
 * otherwise gathered through pinned staging); no transpose happens anywhere -- the kernels' packed form is SNP-major itself.
 * Results are bit-identical to hibag_hip_predict_mapped on the transposed matrix. */
int hibag_hip_predict_snp_major(hibag_hip_model *m, const int32_t *geno, size_t ld, int n_samp, int n_geno_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob);
int hibag_hip_predict_snp_major_device(hibag_hip_model *m, const int32_t *d_geno, size_t ld, int n_samp, int n_geno_snp,
	const int32_t *d_snp_col, const int32_t *d_flip, int vote_method,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, void *stream);

/* ---- launch status of the device-pointer entries ---------------------------------------------------------
 *
 * The last rounds of a pass are cut into chunks that hand their running sums over through the L2 of one XCD
 * ("Hand-overs").  That rests on observed dispatch behaviour, so every hand-over is checked
 * on the device, and one that never arrives (or crosses XCDs) can NOT come back to the caller as numbers:
 *   - the launch's outputs are poisoned on the device: H1 = H2 = NA_integer_, max_prob / matching / dosage /
 *     postprob = NaN for every sample of the batch (partial sums: the three scalar rows are NaN, which survives an
 *     all-reduce and poisons hibag_hip_finish_device's outputs the same way);
 *   - the model gets a STICKY status HIBAG_HIP_EHANDOVER: hibag_hip_model_status() (which waits for the model's
 *     outstanding launches first), hibag_hip_get_timing() and every further compute entry on the model return it until
 *     hibag_hip_model_clear_status() -- the convention of the reference, whose entries never return garbage
 *     (CORE_TRY / CORE_CATCH, src/HIBAG.cpp:41-60; the RAII try_final_* guards, src/LibHLA.cpp:2307-2315);
 *   - from the first such fault on the model launches WITHOUT hand-overs (every work item undivided), so the
 *     caller's second attempt -- clear the status, call again, same process -- cannot fail the same way.
 * The host-pointer entries (hibag_hip_predict, _mapped, _bed, _multi) do all of that themselves:
 * they see the fault when they synchronise, run the call again without hand-overs and return the repaired result
 * with code 0; hibag_hip_model_handover_faults() counts how often that (or a sticky fault) happened.
 *
 * Protocol for a device-pointer caller:   enqueue ... ; synchronise the stream ;
 *     if (hibag_hip_model_status(m) == HIBAG_HIP_EHANDOVER) { hibag_hip_model_clear_status(m); enqueue again; }
 *
 * Streams: a model owns ONE workspace.  Calls on the same model from different streams (or host threads) are legal:
 * the library chains them on the device with an event, each waits for the one enqueued before it.  For concurrency
 * across streams or devices use one model per stream / device (hibag_hip_model_replicate). */
int hibag_hip_model_status(hibag_hip_model *m);
int hibag_hip_model_clear_status(hibag_hip_model *m);
int64_t hibag_hip_model_handover_faults(const hibag_hip_model *m);
/* Fault injection for the tests of the above: the next batch on the model drops the first hand-over of pass `pass`
 * (1 or 2; 0 = disarm) and uses a short time-out.  Not for production use. */
int hibag_hip_test_inject_handover_fault(hibag_hip_model *m, int pass);
/* Diagnostic kernel builds only (-DHIBAG_ACCUM_STAMPS): the clock sums of pass 2's block phases since the last call
 * (n <= 40 values; all zero with the shipped kernels). */
int hibag_hip_test_read_diag(hibag_hip_model *m, unsigned long long *out, int n);

/* How classifier `classifier` of a finalized model computes its distances: *engine = HIBAG_HIP_ENGINE_VALU (bit logic +
 * popcount on the vector ALU), _FP4 (v_mfma_scale_f32_32x32x64_f8f6f4, *k_steps instructions per sample half and
 * 32-pair block) or _I8 / _I8S (v_mfma_i32_32x32x32_i8, two K blocks); what bench.py prices its issue floor with. */
#define HIBAG_HIP_ENGINE_VALU 0
#define HIBAG_HIP_ENGINE_FP4  1
#define HIBAG_HIP_ENGINE_I8   2
#define HIBAG_HIP_ENGINE_I8S  3
int hibag_hip_model_engine(const hibag_hip_model *m, int classifier, int *engine, int *k_steps);

/* ---- several devices of one node: replaces hlaPredict(cl = <cluster>) ----------------------------------------
 *
 * The reference spreads a cohort over the workers of a `parallel` cluster: contiguous sample slices, every worker
 * holding the whole model, results concatenated (R/HIBAG.R:764-808).  Here a "worker" is a device:
 *   hibag_hip_model_replicate  a finalized copy of `m` on `device` (the model takes about 12 bytes per listed haplotype pair -- 11 MB for the benchmark model; every device holds all of it)
 *   hibag_hip_predict_multi    hibag_hip_predict over `n_models` replicas at once: one host thread per replica, each
 *                              takes the contiguous slice hibag_hip_multi_slice gives it and writes its part of every
 *                              output in place.  Samples are independent (src/LibHLA.cpp:2362-2411): no collective, and
 *                              every output is bit-identical to the single-device call.  The replicas may sit on any
 *                              devices, also several on one.
 *   hibag_hip_multi_slice      the slice of replica i: [*first, *first + *count) -- boundaries on multiples of 64 samples. */
hibag_hip_model *hibag_hip_model_replicate(const hibag_hip_model *m, int device);
int hibag_hip_multi_slice(int n_samp, int n_models, int i, int *first, int *count);
int hibag_hip_predict_multi(hibag_hip_model *const *models, int n_models, const int32_t *geno, int n_samp,
	int vote_method, int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob);

/* Classifier-sharded partial pass for multi-GPU runs: each rank owns a model
 * holding a subset of the classifiers but built with the FULL model's per-SNP
 * classifier counts (snp_weight[n_snp], _GetSNPWeights src/LibHLA.cpp:2484-2496,
 * which the classifier weights depend on).  Writes the un-normalised partial
 * ensemble sums so that one sum all-reduce merges ranks:
 *   d_partial [n_hla(n_hla+1)/2 + 3][n_pad]  (n_pad = n_samp rounded up to 64):
 *   rows 0..P-1 sum_c w_c*prob_c, row P sum_c w_c (_Sum_Weight), row P+1
 *   sum_c w_c*total_c (sum_matching), row P+2 sum_c w_c (num_matching,
 *   src/LibHLA.cpp:2458-2459).  hibag_hip_finish_device() turns merged partials into the
 *   PredictHLA outputs. */
int hibag_hip_model_set_snp_weights(hibag_hip_model *m, const int32_t *snp_weight);
int hibag_hip_predict_partial_device(hibag_hip_model *m, const int32_t *d_geno,
	int n_samp, double *d_partial, void *stream);
int hibag_hip_finish_device(hibag_hip_model *m, const double *d_partial, int n_samp,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching,
	double *d_dosage, double *d_postprob, void *stream);

/* The same split driven from ONE process over the devices of a node, merged by RCCL (hibag_amd/csrc/hibag_shard.hip) --
 * what an R / C++ host uses where the model is too large to replicate, or a batch too small to be worth slicing:
 *   hibag_hip_shard_bounds      classifiers [*first, *first + *count) of shard `shard` of `n_shards` (sizes differ by at most one)
 *   hibag_hip_model_shard       that run of `src`'s classifiers as a model of its own on `device`, built with the FULL
 *                               model's per-SNP classifier counts; finalized if `src` is
 *   hibag_hip_model_batch_limit samples one call of hibag_hip_predict_partial_device takes (0 = not finalized)
 *   hibag_hip_shard_group_new   n shards of ONE model, each on its device: one RCCL rank per distinct device
 *                               (ncclCommInitAll; shards sharing a device are added up on it first, in shard order),
 *                               a stream and the batch buffers per rank.  RCCL (librccl.so.1) is loaded here, not before.
 *   hibag_hip_shard_group_predict   CAttrBag_Model::PredictHLA with vote_method 1 (averaged posteriors) on host pointers, as
 *                               hibag_hip_predict: per batch the genotypes go to every rank, every shard writes its
 *                               un-normalised partial sums (the sum split is src/LibHLA.cpp:2448-2476 with :1497-1518),
 *                               ONE ncclAllReduce(ncclDouble, ncclSum) of [P + 3][n_pad] doubles merges them over xGMI,
 *                               rank 0 finishes.  Calls are identical to the unsharded run's and posteriors within 1e-10
 *                               relative (the order of the classifiers' additions changes with the split).  A failed
 *                               hand-over on any shard reaches every rank as NaN through the sum and the batch is run again.
 *   hibag_hip_predict_multi_sharded   group_new + group_predict + group_free in one call (pays the communicator set-up each time)
 *   hibag_hip_shard_group_ranks / _allreduces   RCCL ranks of the group, all-reduces it has issued;  hibag_hip_rccl_version: NCCL_VERSION_CODE of the loaded library, 0 = none
 * The reference's own multi-worker branch (R/HIBAG.R:764-808) splits samples: that is hibag_hip_predict_multi above. */
typedef struct hibag_hip_shard_group hibag_hip_shard_group;
int hibag_hip_shard_bounds(int n_classifier, int n_shards, int shard, int *first, int *count);
hibag_hip_model *hibag_hip_model_shard(const hibag_hip_model *src, int shard, int n_shards, int device);
int hibag_hip_model_batch_limit(const hibag_hip_model *m);
hibag_hip_shard_group *hibag_hip_shard_group_new(hibag_hip_model *const *shards, int n_shards);
void hibag_hip_shard_group_free(hibag_hip_shard_group *g);
int hibag_hip_shard_group_ranks(const hibag_hip_shard_group *g);
int64_t hibag_hip_shard_group_allreduces(const hibag_hip_shard_group *g);
int hibag_hip_rccl_version(void);
int hibag_hip_shard_group_predict(hibag_hip_shard_group *g, const int32_t *geno, int n_samp,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob);
int hibag_hip_predict_multi_sharded(hibag_hip_model *const *shards, int n_shards, const int32_t *geno, int n_samp,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob);

/* ---- PLINK BED input: replaces HIBAG_BEDFlag + HIBAG_ConvBED --------------- */

/* HIBAG_BEDFlag(bed.fn) (src/HIBAG.cpp:1068-1081): returns the storage-mode byte
 * of the file (0 = individual-major, otherwise SNP-major) or a negative error
 * with the reference's messages ("Cannot open the file %s.", "Invalid prefix in
 * the PLINK BED file.").  Host only. */
int hibag_hip_bed_flag(const char *bed_fn);

/* HIBAG_ConvBED(bed.fn, n.samp, n.snp, n.save.snp, snp.flag) (src/HIBAG.cpp:1094-1191):
 * decodes the 2-bit genotypes of the SNPs with snp_flag[j] != 0 (R logical
 * vector, n_snp entries, n_save_snp of them set) on the device into
 *   geno  int32 [n_samp][n_save_snp]  (the memory of R's n_save_snp x n_samp matrix),
 * values 2 / NA_integer_ / 1 / 0 for the codes 0 / 1 / 2 / 3 (:1135).  Only the
 * selected SNP rows of a SNP-major file are read and uploaded.  Unlike the
 * reference, a file shorter than n_samp x n_snp genotypes is an error (EINVAL)
 * rather than a silent reuse of the previous row. */
int hibag_hip_conv_bed(const char *bed_fn, int n_samp, int n_snp, int n_save_snp,
	const int32_t *snp_flag, int32_t *geno);

/* hlaBED2Geno() + hlaPredict() fused: PredictHLA (as hibag_hip_predict) with the
 * genotypes decoded on the device straight from the BED file into the packed
 * form the kernels use, without materialising the int32 matrix (16x less
 * host->device traffic).
 *   n_samp, n_snp   dimensions of the BED file (.fam / .bim line counts)
 *   snp_col[model n_snp]  0-based BED SNP index of each model SNP, -1 = the
 *                   cohort lacks it (treated as missing, like the NA rows
 *                   hlaPredict builds, R/HIBAG.R:640-660)
 *   flip[model n_snp]     NULL, or != 0 where the allele count must be
 *                   reversed, g -> 2 - g (A/B order or strand differs between
 *                   cohort and model, R/HIBAG.R:661-676, src/HIBAG.cpp:221-342)
 * All samples of the file are predicted, in file order. */
int hibag_hip_predict_bed(hibag_hip_model *m, const char *bed_fn, int n_samp, int n_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob);

/* ---- training: replaces HIBAG_Training + HIBAG_NewClassifiers ---------------- */

typedef struct hibag_hip_trainer hibag_hip_trainer;  /* opaque handle */

/* HIBAG_Training(n.snp, n.samp, snp.geno, n.hla, H1, H2) (src/HIBAG.cpp:516-535 ->
 * CAttrBag_Model::InitTraining, src/LibHLA.cpp:2196-2218): snp_geno int32
 * [n_samp][n_snp] (the memory of R's SNP x sample matrix; values outside 0..2 are
 * missing), H1/H2[n_samp] 0-based allele indices < n_hla.  The arrays are copied.
 * Errors carry the reference's messages ("Invalid number of samples: %d.", ...). */
hibag_hip_trainer *hibag_hip_trainer_new(int n_snp, int n_samp, const int32_t *snp_geno, int n_hla,
	const int32_t *H1, const int32_t *H2);
void hibag_hip_trainer_free(hibag_hip_trainer *t);

/* Host threads of the trainer: they pack, reduce and -- in EM mode 1 -- fit the candidate SNPs of a growth step
 * (CAlg_EM::ExpectationMaximization, src/LibHLA.cpp:1185-1255).  The count also SELECTS THE EM ROUTE while the mode is
 * automatic (hibag_hip_trainer_set_em_mode 0, the default; HIBAG_TRAIN_EM=host|device overrides): a trainer with two
 * threads or fewer -- what a torchrun rank on a small quota gets, or a caller passing 1 -- fits on the device, one
 * with more on its threads.  Both routes give the same classifiers bit for bit (the device decides the stopping test
 * with a margin for its own log() and hands the candidates it cannot decide back to a host thread).
 * Default: the CPUs the process may use (affinity mask, cgroup quota) divided by LOCAL_WORLD_SIZE, so that the ranks of
 * one node -- one process per GPU -- share the host instead of oversubscribing it; HIBAG_TRAIN_THREADS overrides the
 * default, n_threads <= 0 restores it.  The counterpart of HIBAG_NewClassifiers' `nthread` (src/HIBAG.cpp:599-634),
 * whose default in hlaAttrBagging is 1 (R/HIBAG.R:48-52): a deviation, see hibag_amd/train.py. */
int hibag_hip_trainer_set_threads(hibag_hip_trainer *t, int n_threads);
int hibag_hip_trainer_threads(const hibag_hip_trainer *t);
/* Where the EM fits of a growth step's candidate SNPs run (CAlg_EM::ExpectationMaximization, src/LibHLA.cpp:1185-1255):
 * 1 = on the trainer's host threads, 2 = on the device (hibag_amd/csrc/hibag_em.hip: workgroup = candidate, every sum in the
 * host's order; the stopping test decided with a margin for the device's log(), candidates it cannot decide handed back to the
 * host), 0 = automatic: the device where the trainer has two host threads or fewer.  Both give the same classifiers bit for bit. */
int hibag_hip_trainer_set_em_mode(hibag_hip_trainer *t, int mode);

/* Several trainers of ONE process on one device -- the decomposition of hlaParallelAttrBagging's workers (R/HIBAG.R:329-390:
 * independent classifiers, one random stream per worker) without a process per worker.  A trainer flagged `shared` hands
 * the device work of its growth steps -- the candidate pair lists (src/LibHLA.cpp:1569-1637), the EM fits (:1127-1255), the
 * scoring (:1639-1767) -- to the device's combiners: whichever trainer thread finds a combiner idle launches ONE fused
 * kernel per kind of operation for every trainer that has one pending (hibag_amd/csrc/hibag_combine.h), instead of each
 * trainer queueing short kernels behind the others' on the runtime's few hardware queues.  Classifiers are bit-identical to
 * a trainer that runs alone.  Each shared trainer is still driven by a host thread of its own (hibag_hip_trainer_new_classifiers
 * blocks), but those threads mostly wait: hibag_hip_train_set_thread_budget(n) lets at most n of them be runnable at a
 * time (0 = no limit; process-wide), so sixteen trainers fit the two host threads a rank of an eight-GPU node gets.
 * hibag_hip_train_combine_stats: fused launches and the operations in them since the last reset, by kind
 * (0 / 1 pair lists pass 0 / 1 as operations of their own, 2 scoring, 3 EM fits, 4 pair lists, both passes in one); out
 * arrays of 8 (either may be NULL). */
int hibag_hip_trainer_set_shared(hibag_hip_trainer *t, int shared);
int hibag_hip_train_set_thread_budget(int n_threads);
int hibag_hip_train_combine_stats(long long *launches, long long *ops, int reset);
/* seconds the shared trainers' operations took from hand-over to results, summed by kind [0..7]; seconds spent in fused
 * batches and their number, for the short operations' lane and the EM lane [8..9], [10..11] (measurement only) */
int hibag_hip_train_combine_times(double *out12, int reset);

/* Source of the uniform draws the reference takes from R's unif_rand()
 * (src/LibHLA.cpp:120-126; bootstrap :2236, SNP sampling :957).  An R binding
 * passes a trampoline to unif_rand between GetRNGstate()/PutRNGstate()
 * (src/HIBAG.cpp:611, :632); other hosts call set_seed, which reproduces R's
 * set.seed(seed) + default Mersenne-Twister stream exactly. */
int hibag_hip_trainer_set_rng(hibag_hip_trainer *t, double (*unif_rand)(void *ctx), void *ctx);
int hibag_hip_trainer_set_seed(hibag_hip_trainer *t, uint32_t seed);

/* HIBAG_NewClassifiers(model, nclassifier, mtry, prune, nthread, verbose,
 * verbose.detail, proc_ptr) (src/HIBAG.cpp:599-634 -> CAttrBag_Model::BuildClassifiers,
 * src/LibHLA.cpp:2268-2305): grows `nclassifier` more individual classifiers -- bootstrap,
 * greedy SNP selection, EM haplotype fit -- with the haplotype-pair scoring
 * (build_haplomatch / build_acc_oob / build_acc_ib) on the device.  Given the same
 * random stream the result is bit-identical to the reference's CPU training. */
int hibag_hip_trainer_new_classifiers(hibag_hip_trainer *t, int nclassifier, int mtry, int prune,
	int verbose, int verbose_detail);

/* HIBAG_GetNumClassifiers / HIBAG_GetClassifierList / HIBAG_Classifier_GetHaplos
 * (src/HIBAG.cpp:843-953): read the grown classifiers back.
 *   snpidx[n_snp_c] 0-based, samp_num[n_samp] bootstrap counts, freq/hla[n_haplo],
 *   bits[n_haplo][2] packed haplotypes (bit s = allele of SNP snpidx[s]).
 * Feed them to hibag_hip_model_add_classifier_packed to predict with the new model. */
int hibag_hip_trainer_n_classifier(const hibag_hip_trainer *t);
int hibag_hip_trainer_classifier_dims(const hibag_hip_trainer *t, int idx, int *n_snp_c, int *n_haplo);
int hibag_hip_trainer_classifier_get(const hibag_hip_trainer *t, int idx, int32_t *snpidx, int32_t *samp_num,
	double *freq, int32_t *hla, uint64_t *bits, double *outofbag_acc);

/* ---- kernel timing (HIP events on the launch stream) --------------------- */

#define HIBAG_HIP_K_PACK     0   /* genotype packing + classifier weights        */
#define HIBAG_HIP_K_TOTAL    1   /* pass 1: per-classifier in-order totals       */
#define HIBAG_HIP_K_ACCUM    2   /* pass 2: normalised weighted accumulation     */
#define HIBAG_HIP_K_FINISH   3   /* arg-max, dosage, transposed posterior output */
#define HIBAG_HIP_K_COUNT    4

/* enabled = 1: every kernel class is bracketed by hipEvents on its stream (classes that follow each other directly share
 * the event between them: five records per batch).  enabled = 2 * mask, mask = OR of (1 << HIBAG_HIP_K_*): only those
 * classes (an event record is a packet of its own on the queue and costs the step a few microseconds each: a caller that
 * wants one kernel's duration out of a timed region asks for that kernel alone).  0: off. */
int hibag_hip_set_timing(hibag_hip_model *m, int enabled);
/* Resolves pending events (synchronises on them) and returns, for kernel `k`,
 * the summed duration in ms and the number of launches since the last reset. */
int hibag_hip_get_timing(hibag_hip_model *m, int k, double *ms_total, int64_t *launches);
int hibag_hip_reset_timing(hibag_hip_model *m);

/* The instruction costs the kernels' issue floor is priced with, measured on the calling thread's
 * current device in about 50 ms: ns per wave64 instruction per SIMD with 8 wavefronts on every SIMD, for FP64 mul / add
 * (the one multiplication and one addition per haplotype pair of src/LibHLA.cpp:1786-1813 are the irreducible part of a
 * pass), v_mfma_i32_32x32x32_i8 and v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands), and the time of a half / half mix of
 * FP64 and int8-MFMA wavefronts as a fraction of the serial sum (1 = the matrix pipe does not hide behind FP64 work).
 * Any pointer may be NULL.  Measurement only: nothing in the library reads these numbers. */
int hibag_hip_measure_issue_costs(double *fp64_op_ns, double *mfma_i8_ns, double *mfma_fp4_ns, double *mix_frac_of_serial);

/* ---- HIBAG plugin table (per-sample, drop-in for an unmodified HIBAG) ---- */

/* Returns a pointer to a static struct laid out exactly like
 * HLA_LIB::TypeGPUExtProc (inst/include/LibHLA_ext.h:358-388): ten function
 * pointers, all implemented.
 *   predict_init / predict_done / predict_avg_prob  (src/LibHLA.cpp:2498-2531, :2433-2441)
 *   build_init / build_done / build_set_bootstrap    (src/LibHLA.cpp:2256-2266, :2290-2293)
 *   build_haplomatch                                 (src/LibHLA.cpp:1037-1063)
 *   build_set_haplo_geno / build_acc_oob / build_acc_ib (src/LibHLA.cpp:1916-1920, :1938-1941, :1961-1964)
 * An R package hands it to hlaPredict() as attr(cl, "proc_ptr") (R/HIBAG.R:707) or
 * to HIBAG_NewClassifiers as its last argument (src/HIBAG.cpp:601-602); see
 * INTEGRATION.md.  Failures inside these void entries throw `const char *`, which
 * the host's CORE_CATCH turns into an R error (src/HIBAG.cpp:41-60).
 * The predict entries are called once per SAMPLE by the host; they run kernels of their own that
 * take their parallelism from the model instead of a batch (thread = allele-pair cell; no work items
 * are cut, so the launch status above does not apply to them) -- about 0.1 ms per call.
 * The table serves ONE model and ONE training state per process at a time, on the device
 * selected with hibag_hip_set_device by the calling thread: predict_init replaces the model of
 * the previous predict_init, build_init the previous build state -- the same restriction as the
 * reference's single staging buffer (gpu_geno_buf, src/LibHLA.h:680), which is why the host must
 * call with nthread = 1. */
const void *hibag_hip_gpu_ext_proc(void);

/* predict_avg_prob is ONE kernel whose workgroups meet at a barrier, so they must all be resident at once.  On a device the
 * process shares (another stream, another process, a profiler) that cannot be promised: a launch whose workgroups give up
 * waiting (a fifth of a second) is repeated on a single workgroup, which waits for nobody -- same results, slower -- and so
 * are the following calls, with the full width tried again every 64th.  Returns how many calls since the last predict_init
 * took that route in a row (0: the device is the process's own). */
long long hibag_hip_plugin_degraded_calls(void);

/* The loop an unmodified HIBAG runs around predict_avg_prob (src/LibHLA.cpp:2362-2411, :2433-2441), compiled like the host's:
 * n_samp calls through the table -- geno: TGenotype [n_samp][n_classifier] (48 bytes each), weight [n_samp][n_classifier] --
 * each followed by BestGuessEnsemble's scan (:1549-1566).  Returns the elapsed time of the loop in *seconds, the winning
 * posterior cell (-1: none) and the matching value per sample.  predict_init must have been called.  For measurements
 * (bench.py: what the zero-change route costs a compiled host, no interpreter between the calls) and tests. */
int hibag_hip_test_time_avg_prob(const void *geno, const double *weight, int n_samp, int n_classifier, int n_cell,
	int32_t *best_cell, double *matching, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* HIBAG_HIP_H_ */
