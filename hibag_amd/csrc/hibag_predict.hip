// hibag_predict.hip -- the batch driver of libhibag_hip.so, what replaces CAttrBag_Model::PredictHLA
// (src/LibHLA.cpp:2317-2412): the per-batch workspace, the kernel sequence of a step (pack, pass 1, pass 2 or the vote,
// finish), the device-pointer entries, the host-pointer entries with their slice pipeline, PLINK BED input, the partial
// sums of classifier shards, the launch status of failed hand-overs, and the per-kernel timers.
//
// There is no CPU fallback here: every compute entry runs the HIP kernels or fails with an error code.

#include "hibag_internal.h"

namespace hibag_detail {

// Samples per batch: bounds the workspace (the stored cell sums of pass 1 dominate: 8 bytes per
// classifier and non-empty cell) to roughly 16 GB of the 288 while keeping batches large enough to
// fill the 256 CUs many times over.
int batch_limit(const hibag_hip_model *m)
{
	const double per_sample = 8.0 * (m->view.n_cell + 3) + 24.0 * m->view.n_classifier +
		4.0 * m->mask_rows + 4.0 * m->view.n_classifier + 16.0 * m->bt_rows + 24.0 * m->view.n_classifier + 8.0 * m->cell_rows;
	double cap = 16e9 / std::max(per_sample, 1.0);
	// k_accum addresses the operand, weight and 1/total arrays through raw buffers with 32-bit offsets: each stays below 4 GB
	cap = std::min(cap, 3.5e9 / (16.0 * std::max(m->bt_rows, 1)));
	cap = std::min(cap, 3.5e9 / (16.0 * std::max(m->view.n_classifier, 1)));     // (winv: 16 bytes per classifier and sample)
	int lim = (int)std::min(cap, 1e9);
	lim = std::max(64, std::min(lim, 1 << 17));
	return lim / 64 * 64;
}

constexpr size_t WS_ERR_BYTES = 16 + 8 * 2040;     // HibagBatchView::err_dev: error word, counter, list (HIBAG_NAN_CAP entries)

int make_batch(hibag_hip_model *m, int n_samp, bool need_best, HibagBatchView &B)
{
	const int n_pad = round_up(std::max(n_samp, 1), HIBAG_WAVE);
	const size_t C = (size_t)std::max(m->view.n_classifier, 1);
	if (int rc = m->ws_planes.reserve((size_t)std::max(m->mask_rows, 1) * n_pad * sizeof(uint32_t))) return rc;
	if (int rc = m->ws_cw.reserve(C * n_pad * sizeof(double))) return rc;
	if (int rc = m->ws_tot.reserve(C * n_pad * sizeof(double))) return rc;
	if (int rc = m->ws_inv.reserve(C * n_pad * sizeof(double))) return rc;
	if (int rc = m->ws_winv.reserve(2 * C * n_pad * sizeof(double))) return rc;
	if (int rc = m->ws_part.reserve((size_t)(m->view.n_cell + 3) * n_pad * sizeof(double))) return rc;
	if (int rc = m->ws_codes.reserve((size_t)std::max(m->n_snp, 1) * n_pad)) return rc;
	// (two rows more than the model has: k_accum reads rows bt and bt + 1 of every block header it passes, whatever the block holds)
	if (int rc = m->ws_bt.reserve((size_t)(std::max(m->bt_rows, 1) + 2) * n_pad * sizeof(uint4))) return rc;
	if (int rc = m->ws_bias.reserve(2 * C * n_pad * sizeof(int))) return rc;
	if (int rc = m->ws_cells.reserve((size_t)std::max(m->cell_rows, 1) * n_pad * sizeof(double))) return rc;
	if (need_best) {
		if (int rc = m->ws_best.reserve(C * n_pad * sizeof(int))) return rc;
		if (int rc = m->ws_vrec.reserve(C * 8 * n_pad * sizeof(uint4))) return rc;      // pass 1's record log (HibagBatchView::vrec)
	}
	{
		// hand-over flags: one per pass-2 item (8 XCDs x group quads x tiles)
		const size_t n_gq = ((size_t)(n_pad / HIBAG_WAVE + 7) / 8 + HIBAG_ACCUM_WAVES - 1) / HIBAG_ACCUM_WAVES;
		const size_t n_flag2 = 8 * n_gq * (size_t)std::max(m->view.n_tile, 1);
		const size_t n_flag1 = (size_t)((n_pad / HIBAG_WAVE + HIBAG_BLOCK_WAVES - 1) / HIBAG_BLOCK_WAVES) * (size_t)std::max(std::max(m->view.n_item_whole, m->view.n_item_split), 1);
		const size_t n_flag = n_flag2 + n_flag1;
		const size_t had = m->ws_sync.cap;
		if (int rc = m->ws_sync.reserve(n_flag * sizeof(unsigned long long))) return rc;
		if (!m->ws_err.p) { if (int rc = m->ws_err.reserve(WS_ERR_BYTES)) return rc; HIP_TRY(hipMemset(m->ws_err.p, 0, WS_ERR_BYTES)); }
		if (m->ws_sync.cap != had) {               // new flags: the epochs start over (and so must the device error word)
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipMemset(m->ws_sync.p, 0, m->ws_sync.cap)); HIP_TRY(hipMemset(m->ws_err.p, 0, 16)); m->epoch = 0;
		}
		if (!m->h_err) {
			HIP_TRY(hipHostMalloc((void **)&m->h_err, sizeof(int), hipHostMallocMapped));
			*m->h_err = 0;
		}
		if (++m->epoch == 0) {
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipMemset(m->ws_sync.p, 0, m->ws_sync.cap)); HIP_TRY(hipMemset(m->ws_err.p, 0, 16)); m->epoch = 1;
		}
	}
	B.sync = m->ws_sync.as<unsigned long long>(); B.epoch = m->epoch; B.err = m->h_err;
	B.err_dev = m->ws_err.as<uint32_t>();
	B.spin_limit = m->spin_limit;
	B.tail_k = m->no_chunks ? 1 : 0;
	B.drop_post = m->drop_next;
	if (m->drop_next) { B.spin_limit = 4096; m->drop_next = 0; }     // (the injected fault should not take the full time-out)
	B.sync_total = B.sync + 8 * (((size_t)(n_pad / HIBAG_WAVE + 7) / 8 + HIBAG_ACCUM_WAVES - 1) / HIBAG_ACCUM_WAVES) * (size_t)std::max(m->view.n_tile, 1);
	B.n_samp = n_samp; B.n_pad = n_pad;
	B.masks = m->ws_planes.as<uint32_t>();
	B.cw = m->ws_cw.as<double>(); B.tot = m->ws_tot.as<double>(); B.inv = m->ws_inv.as<double>(); B.winv = m->ws_winv.as<double>();
	B.part = m->ws_part.as<double>();
	B.bt = m->ws_bt.as<uint4>(); B.bias = m->ws_bias.as<int>();
	B.bt_rows = std::max(m->bt_rows, 1) + 2;
	B.cells = m->ws_cells.as<double>();
	B.vrec = need_best ? m->ws_vrec.as<uint4>() : nullptr;
	return 0;
}

// Passes 1 and 2 (+ majority-vote variant) and the ensemble scalars for a
// batch whose planes / weights are already on the device.
// HIBAG_DEBUG_SYNC=1: wait for the stream behind every stage and name it on stderr (which kernel a device fault belongs to)
static void debug_stage(const char *what, hipStream_t st)
{
	static const bool on = getenv("HIBAG_DEBUG_SYNC") != nullptr;
	if (!on) return;
	const hipError_t e = hipStreamSynchronize(st);
	fprintf(stderr, "[hibag stage] %s: %s\n", what, hipGetErrorString(e));
	fflush(stderr);
}

void run_core(hibag_hip_model *m, HibagBatchView &B, int vote_method, double *d_part, hipStream_t st)
{
	KernelTimer &T = m->timer;
	B.part = d_part;
	debug_stage("pack", st);
	T.begin(HIBAG_HIP_K_TOTAL, st, true);      // (callers enqueue nothing between their pack and this)
	hibag_launch_total(m->view, B, st, m->side, vote_method == 2);
	T.end(st);
	debug_stage("pass 1", st);
	// HIBAG_DEBUG_THRASH_MB=<mb> (measurement only, profiles/r06_notes.txt): overwrite that much scratch memory between the
	// passes, so that pass 2 finds nothing of pass 1's stored sums in the L2s or the Infinity Cache -- what a schedule that keeps
	// them cache-resident could gain at most is the difference to the run without.
	static const long thrash_mb = getenv("HIBAG_DEBUG_THRASH_MB") ? atol(getenv("HIBAG_DEBUG_THRASH_MB")) : 0;
	if (thrash_mb > 0 && m->ws_thrash.reserve((size_t)thrash_mb << 20) == 0)
		(void)hipMemsetAsync(m->ws_thrash.p, 0x5a, (size_t)thrash_mb << 20, st);
	T.begin(HIBAG_HIP_K_ACCUM, st, thrash_mb <= 0);
	if (vote_method == 1) {
		hibag_launch_accum(m->view, B, st);
		debug_stage("pass 2 (accumulate)", st);
		hibag_launch_scalars(m->view, B, nullptr, st);
	} else {
		hibag_launch_vote(m->view, B, m->ws_best.as<int>(), st);
		hibag_launch_scalars(m->view, B, m->ws_best.as<int>(), st);
	}
	T.end(st);
	debug_stage("pass 2", st);
}

int check_predict_args(hibag_hip_model *m, const void *geno, int n_samp, int vote_method,
	const void *H1, const void *H2)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	if (!m->finalized) return hibag_fail(HIBAG_HIP_ESTATE, "model not finalized");
	if (vote_method < 1 || vote_method > 2)
		return hibag_fail(HIBAG_HIP_EINVAL, "Invalid 'vote_method'.");   // src/LibHLA.cpp:2321-2322
	if (n_samp < 0) return hibag_fail(HIBAG_HIP_EINVAL, "n_samp < 0");
	if (n_samp > 0 && !geno) return hibag_fail(HIBAG_HIP_EINVAL, "geno is NULL");
	if ((H1 == nullptr) != (H2 == nullptr)) return hibag_fail(HIBAG_HIP_EINVAL, "H1 and H2 must be given together");
	if (!m->have_snpidx)
		return hibag_fail(HIBAG_HIP_ESTATE, "model was built without SNP indices: raw genotypes cannot be packed");
	return 0;
}

// ---- failed hand-overs ------------------------------------------------------------------------------
// The kernels report a hand-over that never arrived through the host-mapped word (and poison the batch's outputs on the
// device, HibagBatchView::err_dev).  Whoever looks at the word first records it: the fault is counted, the model stops
// cutting work items (K = 1: nothing left to hand over), and -- for launches whose results already went to the caller
// through a device-pointer entry -- it becomes the model's sticky status.
bool take_fault(hibag_hip_model *m)
{
	if (!m->h_err || !*m->h_err) return false;
	*m->h_err = 0;
	m->fault_count++;
	m->no_chunks = true;
	return true;
}

int sticky_fault(hibag_hip_model *m)
{
	if (take_fault(m)) m->fault = HIBAG_HIP_EHANDOVER;
	if (m->fault)
		return hibag_fail(m->fault, "a hand-over between workgroups failed in an earlier launch on this model: the outputs of that "
			"call were poisoned (NA / NaN) and must be computed again; hibag_hip_model_clear_status() re-arms the model, "
			"which from now on launches without hand-overs");
	return 0;
}

// Device-pointer entries share the model's one workspace: chain them on the device, whatever streams they use.
int workspace_enter(hibag_hip_model *m, hipStream_t st)
{
	if (!m->ws_done) HIP_TRY(hipEventCreateWithFlags(&m->ws_done, hipEventDisableTiming));
	if (m->ws_pending) HIP_TRY(hipStreamWaitEvent(st, m->ws_done, 0));
	return 0;
}

int workspace_leave(hibag_hip_model *m, hipStream_t st)
{
	HIP_TRY(hipEventRecord(m->ws_done, st));
	m->ws_pending = true;
	return 0;
}

// Records ws_done when a device-pointer entry returns -- also on its error paths, once anything has been enqueued.
struct WorkspaceGuard {
	hibag_hip_model *m;
	hipStream_t st;
	bool enqueued = false, left = false;
	int leave() { left = true; return workspace_leave(m, st); }
	~WorkspaceGuard() { if (enqueued && !left && m->ws_done) { (void)hipEventRecord(m->ws_done, st); m->ws_pending = true; } }
};

// Where a batch's genotypes come from: the int32 matrix, or a PLINK BED payload.
struct PackSource {
	const int32_t *d_geno = nullptr;       // [n_samp][row_len]
	int row_len = 0;                       // SNPs per sample in d_geno (0: the model's n_snp, model order)
	size_t ld = 0;                         // != 0: d_geno is SNP-MAJOR, [rows][ld] with one row of genotypes per SNP (k_codes_rows); d_col = row of each model SNP
	const int32_t *d_col = nullptr;        // [n_snp] column of each model SNP in d_geno (-1 = absent), nullptr = identity
	const uint8_t *d_bed = nullptr;        // payload rows (see k_bed_codes)
	int mode = 0;
	size_t stride = 0;
	int samp0 = 0;                         // BED sample index of the call's sample 0
	const int32_t *d_row = nullptr, *d_flip = nullptr;
};

int predict_device_locked(hibag_hip_model *m, const PackSource &src, int n_samp, int vote_method,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, hipStream_t st)
{
	HIP_TRY(hipSetDevice(m->device));
	if (int rc = workspace_enter(m, st)) return rc;
	// Whatever way this call ends, work it has enqueued still uses the shared workspace: the next call on another stream
	// must be chained behind it (an error in a later batch used to skip the record).
	WorkspaceGuard guard{m, st};
	const int lim = batch_limit(m);
	const size_t P = (size_t)m->view.n_cell;
	for (int s0 = 0; s0 < n_samp; s0 += lim) {
		const int n = std::min(lim, n_samp - s0);
		HibagBatchView B;
		if (int rc = make_batch(m, n, vote_method == 2, B)) return rc;
		guard.enqueued = true;
		m->timer.begin(HIBAG_HIP_K_PACK, st);
		if (src.d_bed)
			hibag_launch_pack_bed(m->view, B, src.d_bed, src.mode, src.stride, src.samp0 + s0, src.d_row, src.d_flip,
				m->ws_codes.as<uint8_t>(), st);
		else if (src.ld)
			hibag_launch_pack_rows(m->view, B, src.d_geno + s0, src.ld, src.d_col, src.d_flip, m->ws_codes.as<uint8_t>(), st);
		else
			hibag_launch_pack(m->view, B, src.d_geno + (size_t)s0 * (src.d_col ? src.row_len : m->n_snp), src.row_len,
				src.d_col, src.d_flip, m->ws_codes.as<uint8_t>(), st);
		m->timer.end(st);
		run_core(m, B, vote_method, m->ws_part.as<double>(), st);
		m->timer.begin(HIBAG_HIP_K_FINISH, st, true);
		hibag_launch_finish(m->view, B, B.part,
			d_H1 ? d_H1 + s0 : nullptr, d_H2 ? d_H2 + s0 : nullptr,
			d_max_prob ? d_max_prob + s0 : nullptr, d_matching ? d_matching + s0 : nullptr,
			d_dosage ? d_dosage + (size_t)s0 * m->n_hla : nullptr,
			d_postprob ? d_postprob + (size_t)s0 * P : nullptr, st);
		m->timer.end(st);
	}
	HIP_TRY(hipGetLastError());
	return guard.leave();
}

// Host-pointer driver.  The cohort is cut into slices (bounded workspace, bounded genotype staging); consecutive slices
// are pipelined over three streams of the model's -- upload of slice i+1 and download of slice i-1 beside the kernels of
// slice i, genotype and output buffers doubled -- so that for cohorts of several slices only the first upload and the last
// download are exposed (SURVEY.md section 8d's protocol counts both).  Genotypes come from the host int32 matrix or from a
// BED payload already on the device.  A failed hand-over (poisoned outputs) is repaired here: the call is run again
// with undivided work items, in this process, before anything is returned.
int staged_streams(hibag_hip_model *m, StagedStreams **out)
{
	StagedStreams *ss = &m->staged;
	if (!m->staged_ready) {
		if (!getenv("HIBAG_STAGED_NULL")) HIP_TRY(hipStreamCreateWithFlags(&ss->run, hipStreamNonBlocking));    // (diagnostic: the null stream)
		HIP_TRY(hipStreamCreateWithFlags(&ss->in, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&ss->out, hipStreamNonBlocking));
		for (int i = 0; i < 2; i++) {
			HIP_TRY(hipEventCreateWithFlags(&ss->up[i], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ss->ran[i], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ss->down[i], hipEventDisableTiming));
		}
		m->staged_ready = true;
	}
	*out = ss;
	return 0;
}

// Samples per slice of the host-pointer entries: the workspace bound, at most ~1 GB of staged genotypes (a cohort matrix
// may carry every SNP of the genome: `row_len` is the cohort's, not the model's), and -- for cohorts worth pipelining --
// 12,288 samples: measured on the benchmark model at 100,000 samples (tools/host_path_probe.py, profiles/r03_staged_slices.txt)
// slices of 10-12k give 17.0 ms against 15.4 with the cohort resident in HBM; 25k: 18.0, 50k: 19.4, one slice: 18.5 (what is
// exposed is the first upload and the last download, and a batch of 12k runs within 2 % of the speed of one of 100k).
int staged_slice(const hibag_hip_model *m, int n_samp, size_t row_len)
{
	long long slice = std::min<long long>(batch_limit(m), ((long long)std::max(n_samp, 1) + 63) / 64 * 64);
	const long long by_geno = (long long)((1ull << 30) / (std::max<size_t>(row_len, 1) * sizeof(int32_t)));
	slice = std::min(slice, std::max<long long>(64, by_geno));
	if (n_samp >= 2 * 12288) slice = std::min<long long>(slice, 12288);
	if (const char *e = getenv("HIBAG_STAGED_SLICE")) slice = std::min<long long>(batch_limit(m), std::max(64, atoi(e)));     // (diagnostic)
	return (int)std::max<long long>(64, (slice + 63) / 64 * 64);
}

// A SNP-major host matrix (hibag_hip_predict_snp_major): geno[rows[r] * ld + s] is staged as row r of a slice's
// [rows.size()][n] device matrix -- only the rows the model uses travel.
struct HostRows {
	size_t ld = 0;
	std::vector<size_t> rows;
	bool consecutive = false;              // rows[r] = rows[0] + r: a slice is one strided block of the caller's matrix
};

int predict_staged_locked(hibag_hip_model *m, const int32_t *geno, const PackSource *bed, int n_samp, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob,
	const PackSource *map = nullptr, bool is_retry = false, const HostRows *hr = nullptr)
{
	// `map`: geno is the cohort's own matrix (map->row_len SNPs per sample); map->d_col / d_flip sit on the device
	// `hr` (with `map` for d_col / d_flip): the cohort's matrix is SNP-major, see HostRows
	// A device-pointer launch still running on another stream may yet fail a hand-over: wait for it, so that its fault
	// becomes the model's sticky status (its caller's to see) instead of being taken for this call's own and repaired away.
	if (m->ws_pending && m->ws_done && !is_retry) HIP_TRY(hipEventSynchronize(m->ws_done));
	if (int rc = sticky_fault(m)) return rc;
	const size_t P = (size_t)m->view.n_cell, nh = (size_t)m->n_hla,
		S = hr ? std::max<size_t>(hr->rows.size(), 1) : map ? (size_t)map->row_len : (size_t)m->n_snp;
	const int slice = staged_slice(m, n_samp, bed ? 1 : S);
	const size_t geno_bytes = ((size_t)slice * std::max<size_t>(S, 1) * sizeof(int32_t) + 255) / 256 * 256;
	const size_t o_h1 = 0, o_h2 = o_h1 + (size_t)slice * 4, o_mp = (o_h2 + (size_t)slice * 4 + 7) / 8 * 8,
		o_mt = o_mp + (size_t)slice * 8, o_ds = o_mt + (size_t)slice * 8, o_pp = o_ds + (size_t)slice * nh * 8,
		out_bytes = (o_pp + (postprob ? (size_t)slice * P * 8 : 0) + 255) / 256 * 256;
	// A pipelined run starts with a shorter slice: what nothing overlaps with is the staging and upload of the FIRST slice,
	// and a third of a slice costs the kernels less (their last rounds are emptier) than the wait it saves.
	const bool piped = n_samp > slice;
	static const int first_env = getenv("HIBAG_STAGED_FIRST") ? atoi(getenv("HIBAG_STAGED_FIRST")) : 0;     // (diagnostic)
	const int first = piped ? std::max(64, std::min(slice, (first_env > 0 ? first_env : slice / 3) / 64 * 64)) : slice;
	const int n_slice = piped ? 1 + (n_samp - first + slice - 1) / slice : 1;
	const int nbuf = piped ? 2 : 1;
	if (!bed)
		if (int rc = m->ws_geno.reserve(geno_bytes * nbuf)) return rc;
	if (int rc = m->ws_out.reserve(out_bytes * nbuf)) return rc;
	StagedStreams *ss;
	if (int rc = staged_streams(m, &ss)) return rc;
	if (piped) {
		// pinned staging on the host side, so that every copy call returns at once and the host thread's own work -- filling
		// and draining the staging buffers, ~50 GB/s -- runs beside the kernels too (transfers from / to the caller's pageable
		// memory are as fast on this platform, but the calls block: tools/copy_probe, profiles/r03_copy_probe.txt)
		if (!bed) if (int rc = m->pin_geno.reserve(geno_bytes * 2)) return rc;
		if (int rc = m->pin_out.reserve(out_bytes * 2)) return rc;
	} else if (hr && !hr->consecutive) {
		if (int rc = m->pin_geno.reserve(geno_bytes)) return rc;          // (scattered rows are gathered on the host side)
	}
	auto slice_of = [&](int i, int &s0, int &n) {
		if (i == 0) { s0 = 0; n = std::min(first, n_samp); }
		else { s0 = first + (i - 1) * slice; n = std::min(slice, n_samp - s0); }
	};
	auto upload = [&](int i) -> int {
		if (bed) return 0;
		int s0, n; slice_of(i, s0, n);
		const size_t bytes = (size_t)n * S * sizeof(int32_t);
		char *dst = m->ws_geno.as<char>() + (size_t)(i % nbuf) * geno_bytes;
		if (hr) {
			// SNP-major source: row r of the slice's device matrix [rows][n] = n genotypes of the caller's row rows[r] from sample s0
			const size_t nr = hr->rows.size(), w = (size_t)n * sizeof(int32_t);
			if (nr == 0) return 0;
			const int32_t *first = geno + hr->rows[0] * hr->ld + (size_t)s0;
			if (!piped && hr->consecutive) {
				// one block of the caller's matrix (the whole of it when the cohort's SNPs are the model's): no host copy
				if (hr->ld == (size_t)n) HIP_TRY(hipMemcpyAsync(dst, first, nr * w, hipMemcpyHostToDevice, ss->run));
				else HIP_TRY(hipMemcpy2DAsync(dst, w, first, hr->ld * sizeof(int32_t), w, nr, hipMemcpyHostToDevice, ss->run));
				return 0;
			}
			char *pin = (char *)m->pin_geno.p + (piped ? (size_t)(i & 1) * geno_bytes : 0);
			if (piped && i >= 2) HIP_TRY(hipEventSynchronize(ss->up[i & 1]));
			for (size_t r = 0; r < nr; r++) memcpy(pin + r * w, geno + hr->rows[r] * hr->ld + (size_t)s0, w);
			if (piped && i >= 2) HIP_TRY(hipStreamWaitEvent(ss->in, ss->ran[i & 1], 0));
			HIP_TRY(hipMemcpyAsync(dst, pin, nr * w, hipMemcpyHostToDevice, piped ? ss->in : ss->run));
			if (piped) HIP_TRY(hipEventRecord(ss->up[i & 1], ss->in));
			return 0;
		}
		if (!piped) {
			HIP_TRY(hipMemcpyAsync(dst, geno + (size_t)s0 * S, bytes, hipMemcpyHostToDevice, ss->run));
			return 0;
		}
		char *pin = (char *)m->pin_geno.p + (size_t)(i & 1) * geno_bytes;
		if (i >= 2) HIP_TRY(hipEventSynchronize(ss->up[i & 1]));              // the transfer of slice i - 2 has left the staging buffer
		memcpy(pin, geno + (size_t)s0 * S, bytes);
		if (i >= 2) HIP_TRY(hipStreamWaitEvent(ss->in, ss->ran[i & 1], 0));   // ... and its kernels have read the device buffer
		HIP_TRY(hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, ss->in));
		HIP_TRY(hipEventRecord(ss->up[i & 1], ss->in));
		return 0;
	};
	bool small_staged = false;
	if (!piped)
		if (int rc = m->pin_out.reserve(o_ds)) return rc;
	// device -> host of slice i's outputs: straight into the caller's arrays (one slice), or into the pinned staging buffer
	auto download = [&](int i) -> int {
		int s0, n; slice_of(i, s0, n);
		const char *o = m->ws_out.as<char>() + (size_t)(i % nbuf) * out_bytes;
		if (piped) {
			HIP_TRY(hipStreamWaitEvent(ss->out, ss->ran[i & 1], 0));
			const size_t used = (postprob ? o_pp + (size_t)n * P * 8 : dosage ? o_ds + (size_t)n * nh * 8 : o_ds);
			HIP_TRY(hipMemcpyAsync((char *)m->pin_out.p + (size_t)(i & 1) * out_bytes, o, used, hipMemcpyDeviceToHost, ss->out));
			HIP_TRY(hipEventRecord(ss->down[i & 1], ss->out));
			return 0;
		}
		hipStream_t st = ss->run;
		// One slice: the four per-sample vectors (calls, probability, matching: 24 bytes per sample, contiguous on the device)
		// come down in ONE copy into pinned staging and are handed out behind the final synchronisation -- a copy into the
		// caller's pageable memory holds the calling thread for ~12 us whatever its size, and there were four of them; the
		// large ones (dosage, posterior) go straight to the caller's arrays.
		if (H1 || max_prob || matching) {
			HIP_TRY(hipMemcpyAsync(m->pin_out.p, o, o_ds, hipMemcpyDeviceToHost, st));
			small_staged = true;
		}
		if (dosage) HIP_TRY(hipMemcpyAsync(dosage + (size_t)s0 * nh, o + o_ds, (size_t)n * nh * 8, hipMemcpyDeviceToHost, st));
		if (postprob) HIP_TRY(hipMemcpyAsync(postprob + (size_t)s0 * P, o + o_pp, (size_t)n * P * 8, hipMemcpyDeviceToHost, st));
		return 0;
	};
	// staging buffer -> the caller's arrays (pipelined runs)
	auto drain = [&](int i) -> int {
		int s0, n; slice_of(i, s0, n);
		HIP_TRY(hipEventSynchronize(ss->down[i & 1]));
		const char *o = (const char *)m->pin_out.p + (size_t)(i & 1) * out_bytes;
		if (H1) { memcpy(H1 + s0, o + o_h1, (size_t)n * 4); memcpy(H2 + s0, o + o_h2, (size_t)n * 4); }
		if (max_prob) memcpy(max_prob + s0, o + o_mp, (size_t)n * 8);
		if (matching) memcpy(matching + s0, o + o_mt, (size_t)n * 8);
		if (dosage) memcpy(dosage + (size_t)s0 * nh, o + o_ds, (size_t)n * nh * 8);
		if (postprob) memcpy(postprob + (size_t)s0 * P, o + o_pp, (size_t)n * P * 8);
		return 0;
	};
	static const bool trace = getenv("HIBAG_STAGED_TRACE") != nullptr;     // diagnostic: host time of each phase on stderr
	auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tr[6] = {now(), 0, 0, 0, 0, 0};
	if (int rc = upload(0)) return rc;
	if (trace) { if (!piped) (void)hipStreamSynchronize(ss->run); tr[1] = now(); }
	for (int i = 0; i < n_slice; i++) {
		int s0, n; slice_of(i, s0, n);
		char *o = m->ws_out.as<char>() + (size_t)(i % nbuf) * out_bytes;
		PackSource src;
		if (bed) {
			src = *bed;
			src.samp0 = bed->samp0 + s0;
		} else {
			if (map) src = *map;
			src.d_geno = (const int32_t *)(m->ws_geno.as<char>() + (size_t)(i % nbuf) * geno_bytes);
			if (hr) src.ld = (size_t)n;
			if (piped) HIP_TRY(hipStreamWaitEvent(ss->run, ss->up[i & 1], 0));
		}
		if (piped && i >= 2) HIP_TRY(hipStreamWaitEvent(ss->run, ss->down[i & 1], 0));    // slice i - 2 has left the device output buffer
		if (int rc = predict_device_locked(m, src, n, vote_method,
				H1 ? (int32_t *)(o + o_h1) : nullptr, H2 ? (int32_t *)(o + o_h2) : nullptr,
				max_prob ? (double *)(o + o_mp) : nullptr, matching ? (double *)(o + o_mt) : nullptr,
				dosage ? (double *)(o + o_ds) : nullptr, postprob ? (double *)(o + o_pp) : nullptr, ss->run))
			return rc;
		if (piped) HIP_TRY(hipEventRecord(ss->ran[i & 1], ss->run));
		if (trace && piped) fprintf(stderr, "[hibag staged] slice %d enqueued at %.3f ms\n", i, now() - tr[0]);
		if (trace && !piped) { tr[2] = now(); (void)hipStreamSynchronize(ss->run); tr[3] = now(); }
		// With the kernels of slice i enqueued, the host fills the next staging buffer and starts its transfer -- BEFORE the
		// download of slice i is queued: the copy engine takes transfers in submission order, and a download that waits for
		// its kernels would hold up every upload submitted behind it (measured: no overlap at all the other way round).
		if (i + 1 < n_slice) if (int rc = upload(i + 1)) return rc;
		if (piped && i >= 2) if (int rc = drain(i - 2)) return rc;          // (frees the staging buffer download(i) writes)
		if (int rc = download(i)) return rc;
		if (trace && piped) fprintf(stderr, "[hibag staged] slice %d: download queued, next upload staged at %.3f ms\n", i, now() - tr[0]);
	}
	if (piped) {
		if (n_slice >= 2) if (int rc = drain(n_slice - 2)) return rc;
		if (int rc = drain(n_slice - 1)) return rc;
	}
	if (trace) tr[4] = now();
	HIP_TRY(hipStreamSynchronize(ss->run));
	if (small_staged) {                                  // (one slice: s0 = 0, n = n_samp)
		const char *o = (const char *)m->pin_out.p;
		if (H1) { memcpy(H1, o + o_h1, (size_t)n_samp * 4); memcpy(H2, o + o_h2, (size_t)n_samp * 4); }
		if (max_prob) memcpy(max_prob, o + o_mp, (size_t)n_samp * 8);
		if (matching) memcpy(matching, o + o_mt, (size_t)n_samp * 8);
	}
	if (trace) {
		tr[5] = now();
		if (!piped) fprintf(stderr, "[hibag staged] n=%d upload %.3f  enqueue %.3f  kernels %.3f  download calls %.3f  final sync %.3f ms\n", n_samp,
			tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4]);
		else fprintf(stderr, "[hibag staged] n=%d in %d slices of %d: %.3f ms\n", n_samp, n_slice, slice, tr[5] - tr[0]);
	}
	if (take_fault(m)) {
		// poisoned outputs: once more, now without hand-overs (take_fault switched them off) -- never returned to the caller
		if (is_retry) return hibag_fail(HIBAG_HIP_EHANDOVER, "a hand-over between workgroups failed in a launch without hand-overs");
		return predict_staged_locked(m, geno, bed, n_samp, vote_method, H1, H2, max_prob, matching, dosage, postprob, map, true, hr);
	}
	return 0;
}

// ---------------------------------------------------------------------------
// PLINK BED files (HIBAG_BEDFlag / HIBAG_ConvBED, src/HIBAG.cpp:1068-1191)

// Host image of the part of a BED file a call needs.  SNP-major files keep only
// the rows of the wanted SNPs (a cohort file holds the whole genome, a model
// ~10^2-10^3 SNPs); individual-major files are kept whole.
struct BedImage {
	int mode = 0;
	size_t stride = 0;                 // bytes per row
	std::vector<uint8_t> rows;         // payload
	std::vector<int32_t> index;        // per wanted SNP: row (SNP-major) / column (individual-major) in `rows`, -1 = absent
};

int read_bed_prefix(FILE *f, int *mode)
{
	unsigned char prefix[3];
	if (fread(prefix, 1, 3, f) != 3 || prefix[0] != 0x6C || prefix[1] != 0x1B)
		return hibag_fail(HIBAG_HIP_EINVAL, "Invalid prefix in the PLINK BED file.");   // src/HIBAG.cpp:1077-1078, :1112-1113
	*mode = prefix[2];
	return 0;
}

// want[n_want]: BED SNP indices (0-based, -1 = none).
int load_bed(const char *fn, int n_samp, int n_snp, const int32_t *want, int n_want, BedImage &img)
{
	if (!fn) return hibag_fail(HIBAG_HIP_EINVAL, "bed file name is NULL");
	if (n_samp < 0 || n_snp < 0) return hibag_fail(HIBAG_HIP_EINVAL, "negative dimensions (n_samp=%d, n_snp=%d)", n_samp, n_snp);
	FILE *f = fopen(fn, "rb");
	if (!f) return hibag_fail(HIBAG_HIP_EINVAL, "Fail to open the file \"%s\".", fn);   // src/HIBAG.cpp:1106-1107
	struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
	if (int rc = read_bed_prefix(f, &img.mode)) return rc;
	for (int j = 0; j < n_want; j++)
		if (want[j] >= n_snp) return hibag_fail(HIBAG_HIP_EINVAL, "SNP index %d outside the BED file's %d SNPs", want[j], n_snp);
	img.index.assign(n_want, -1);
	const char *short_msg = "the PLINK BED file holds fewer than %d x %d genotypes";
	if (img.mode == 0) {
		img.stride = ((size_t)n_snp + 3) / 4;
		img.rows.resize(img.stride * (size_t)n_samp);
		if (!img.rows.empty() && fread(img.rows.data(), 1, img.rows.size(), f) != img.rows.size())
			return hibag_fail(HIBAG_HIP_EINVAL, short_msg, n_samp, n_snp);
		for (int j = 0; j < n_want; j++) img.index[j] = want[j];
	} else {
		img.stride = ((size_t)n_samp + 3) / 4;
		int n_row = 0, lo = n_snp, hi = -1;
		for (int j = 0; j < n_want; j++) if (want[j] >= 0) { n_row++; lo = std::min(lo, want[j]); hi = std::max(hi, want[j]); }
		if (n_row > 0 && (size_t)(hi - lo + 1) <= 2 * (size_t)n_row) {
			// the wanted rows lie close together (a model's SNPs are one region of the chromosome): ONE read of the range they
			// span instead of a seek and a read per row -- 150 system calls were a tenth of a 10,000-sample call
			img.rows.resize(img.stride * (size_t)(hi - lo + 1));
			if (fseeko(f, (off_t)3 + (off_t)img.stride * lo, SEEK_SET) != 0 ||
				(!img.rows.empty() && fread(img.rows.data(), 1, img.rows.size(), f) != img.rows.size()))
				return hibag_fail(HIBAG_HIP_EINVAL, short_msg, n_samp, n_snp);
			for (int j = 0; j < n_want; j++) img.index[j] = want[j] >= 0 ? want[j] - lo : -1;
			return 0;
		}
		img.rows.resize(img.stride * (size_t)n_row);
		int r = 0;
		for (int j = 0; j < n_want; j++) {
			if (want[j] < 0) continue;
			if (fseeko(f, (off_t)3 + (off_t)img.stride * want[j], SEEK_SET) != 0 ||
				(img.stride && fread(img.rows.data() + img.stride * (size_t)r, 1, img.stride, f) != img.stride))
				return hibag_fail(HIBAG_HIP_EINVAL, short_msg, n_samp, n_snp);
			img.index[j] = r++;
		}
	}
	return 0;
}

} // namespace hibag_detail

// ===========================================================================
// C ABI: prediction, status, timing

extern "C" {

int hibag_hip_predict_device(hibag_hip_model *m, const int32_t *d_geno, int n_samp, int vote_method,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, void *stream)
{
	if (int rc = check_predict_args(m, d_geno, n_samp, vote_method, d_H1, d_H2)) return rc;
	std::lock_guard<std::mutex> g(m->lock);
	if (int rc = sticky_fault(m)) return rc;
	PackSource src;
	src.d_geno = d_geno;
	return predict_device_locked(m, src, n_samp, vote_method, d_H1, d_H2, d_max_prob, d_matching,
		d_dosage, d_postprob, (hipStream_t)stream);
}

int hibag_hip_predict(hibag_hip_model *m, const int32_t *geno, int n_samp, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (int rc = check_predict_args(m, geno, n_samp, vote_method, H1, H2)) return rc;
	if (n_samp == 0) return 0;
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	return predict_staged_locked(m, geno, nullptr, n_samp, vote_method, H1, H2, max_prob, matching, dosage, postprob);
}

// samples per batch of the device-pointer entries that take ONE batch (hibag_hip_predict_partial_device); 0 = not finalized
int hibag_hip_model_batch_limit(const hibag_hip_model *m) { return m && m->finalized ? batch_limit(m) : 0; }

int hibag_hip_multi_slice(int n_samp, int n_models, int i, int *first, int *count)
{
	if (n_samp < 0 || n_models <= 0 || i < 0 || i >= n_models) return hibag_fail(HIBAG_HIP_EINVAL, "bad slice query (n_samp=%d, n_models=%d, i=%d)", n_samp, n_models, i);
	// contiguous slices whose boundaries fall on multiples of 64 samples (a wavefront's worth) wherever the cohort allows
	const long long groups = ((long long)n_samp + 63) / 64;
	const long long a = std::min<long long>(n_samp, groups * i / n_models * 64), b = std::min<long long>(n_samp, groups * (i + 1) / n_models * 64);
	if (first) *first = (int)a;
	if (count) *count = (int)(b - a);
	return 0;
}

int hibag_hip_predict_multi(hibag_hip_model *const *models, int n_models, const int32_t *geno, int n_samp, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (!models || n_models <= 0) return hibag_fail(HIBAG_HIP_EINVAL, "no models given");
	for (int i = 0; i < n_models; i++) {
		if (int rc = check_predict_args(models[i], geno, n_samp, vote_method, H1, H2)) return rc;
		if (models[i]->n_hla != models[0]->n_hla || models[i]->n_snp != models[0]->n_snp || models[i]->cls.size() != models[0]->cls.size())
			return hibag_fail(HIBAG_HIP_EINVAL, "model %d is not a replica of model 0", i);
	}
	if (n_samp == 0) return 0;
	const size_t S = (size_t)models[0]->n_snp, nh = (size_t)models[0]->n_hla, P = nh * (nh + 1) / 2;
	// One host thread per replica: each drives its own device through the ordinary host-pointer entry on its slice of the
	// cohort and writes its slice of every output in place -- samples are independent (src/LibHLA.cpp:2362-2411), nothing is
	// merged.  The first non-empty slice runs on the calling thread.  No C++ exception leaves this function (thread
	// creation and the vectors below can throw): threads already started are joined, the call fails with ENOMEM.
	std::vector<std::thread> th;
	int code = 0, who = -1;
	std::string text;
	try {
		std::vector<int> rc(n_models, 0);
		std::vector<std::string> msg(n_models);
		auto run = [&](int i, int first, int count) {
			rc[i] = hibag_hip_predict(models[i], geno + (size_t)first * S, count, vote_method,
				H1 ? H1 + first : nullptr, H2 ? H2 + first : nullptr, max_prob ? max_prob + first : nullptr,
				matching ? matching + first : nullptr, dosage ? dosage + (size_t)first * nh : nullptr,
				postprob ? postprob + (size_t)first * P : nullptr);
			if (rc[i]) { try { msg[i] = hibag_hip_last_error(); } catch (...) {} }
		};
		int mine = -1, mine_first = 0, mine_count = 0;
		th.reserve(n_models);
		for (int i = 0; i < n_models; i++) {
			int first = 0, count = 0;
			(void)hibag_hip_multi_slice(n_samp, n_models, i, &first, &count);
			if (count == 0) continue;
			if (mine < 0) { mine = i; mine_first = first; mine_count = count; continue; }
			th.emplace_back(run, i, first, count);
		}
		if (mine >= 0) run(mine, mine_first, mine_count);
		for (auto &t : th) t.join();
		th.clear();
		for (int i = 0; i < n_models && !code; i++)
			if (rc[i]) { code = rc[i]; who = i; text = msg[i]; }
	} catch (...) {
		for (auto &t : th) if (t.joinable()) t.join();
		return hibag_fail(HIBAG_HIP_ENOMEM, "hibag_hip_predict_multi: could not start a host thread per replica");
	}
	if (code) return hibag_fail(code, "replica %d (device %d): %s", who, models[who]->device, text.c_str());
	return 0;
}

int hibag_hip_predict_mapped(hibag_hip_model *m, const int32_t *geno, int n_samp, int n_geno_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (int rc = check_predict_args(m, geno, n_samp, vote_method, H1, H2)) return rc;
	if (n_geno_snp <= 0) return hibag_fail(HIBAG_HIP_EINVAL, "n_geno_snp must be positive");
	if (!snp_col && m->n_snp > 0) return hibag_fail(HIBAG_HIP_EINVAL, "snp_col is NULL");
	for (int k = 0; k < m->n_snp; k++)
		if (snp_col[k] >= n_geno_snp) return hibag_fail(HIBAG_HIP_EINVAL, "snp_col[%d] = %d outside the %d SNPs of the genotype matrix", k, snp_col[k], n_geno_snp);
	if (n_samp == 0) return 0;
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	const size_t S = (size_t)std::max(m->n_snp, 1);
	std::vector<int32_t> idx(2 * S, 0);
	for (int k = 0; k < m->n_snp; k++) {
		idx[k] = snp_col[k] < 0 ? -1 : snp_col[k];
		idx[S + k] = flip ? (flip[k] != 0) : 0;
	}
	if (int rc = m->ws_bedidx.reserve(idx.size() * sizeof(int32_t))) return rc;
	HIP_TRY(hipMemcpyAsync(m->ws_bedidx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, 0));
	HIP_TRY(hipStreamSynchronize(0));            // `idx` is pageable host memory about to go out of scope
	PackSource map;
	map.row_len = n_geno_snp;
	map.d_col = m->ws_bedidx.as<int32_t>();
	map.d_flip = m->ws_bedidx.as<int32_t>() + S;
	return predict_staged_locked(m, geno, nullptr, n_samp, vote_method, H1, H2, max_prob, matching, dosage, postprob, &map);
}

int hibag_hip_predict_mapped_device(hibag_hip_model *m, const int32_t *d_geno, int n_samp, int n_geno_snp,
	const int32_t *d_snp_col, const int32_t *d_flip, int vote_method,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, void *stream)
{
	if (int rc = check_predict_args(m, d_geno, n_samp, vote_method, d_H1, d_H2)) return rc;
	if (n_geno_snp <= 0 || !d_snp_col) return hibag_fail(HIBAG_HIP_EINVAL, "n_geno_snp must be positive and d_snp_col given");
	std::lock_guard<std::mutex> g(m->lock);
	if (int rc = sticky_fault(m)) return rc;
	PackSource src;
	src.d_geno = d_geno; src.row_len = n_geno_snp; src.d_col = d_snp_col; src.d_flip = d_flip;
	return predict_device_locked(m, src, n_samp, vote_method, d_H1, d_H2, d_max_prob, d_matching,
		d_dosage, d_postprob, (hipStream_t)stream);
}

// The cohort's matrix SNP-major: geno[row][sample] with `ld` elements between rows.  Only the model's rows are uploaded.
int hibag_hip_predict_snp_major(hibag_hip_model *m, const int32_t *geno, size_t ld, int n_samp, int n_geno_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (int rc = check_predict_args(m, geno, n_samp, vote_method, H1, H2)) return rc;
	if (n_geno_snp <= 0) return hibag_fail(HIBAG_HIP_EINVAL, "n_geno_snp must be positive");
	if (ld < (size_t)n_samp) return hibag_fail(HIBAG_HIP_EINVAL, "ld = %zu is smaller than n_samp = %d", ld, n_samp);
	if (!snp_col && m->n_snp > n_geno_snp)
		return hibag_fail(HIBAG_HIP_EINVAL, "snp_col is NULL (rows in model order) but the matrix has %d rows for the model's %d SNPs", n_geno_snp, m->n_snp);
	for (int k = 0; snp_col && k < m->n_snp; k++)
		if (snp_col[k] >= n_geno_snp) return hibag_fail(HIBAG_HIP_EINVAL, "snp_col[%d] = %d outside the %d SNPs of the genotype matrix", k, snp_col[k], n_geno_snp);
	if (n_samp == 0) return 0;
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	const size_t S = (size_t)std::max(m->n_snp, 1);
	HostRows hr;
	std::vector<int32_t> idx;
	try {
		idx.assign(2 * S, 0);
		hr.ld = ld;
		hr.rows.reserve(S);
		for (int k = 0; k < m->n_snp; k++) {
			const int c = snp_col ? snp_col[k] : k;
			idx[k] = c < 0 ? -1 : (int32_t)hr.rows.size();           // (the staged matrix holds the model's rows only, in model order)
			if (c >= 0) hr.rows.push_back((size_t)c);
			idx[S + k] = flip ? (flip[k] != 0) : 0;
		}
	} catch (...) { return hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); }
	hr.consecutive = true;
	for (size_t r = 1; r < hr.rows.size(); r++) if (hr.rows[r] != hr.rows[0] + r) { hr.consecutive = false; break; }
	if (int rc = m->ws_bedidx.reserve(idx.size() * sizeof(int32_t))) return rc;
	HIP_TRY(hipMemcpyAsync(m->ws_bedidx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, 0));
	HIP_TRY(hipStreamSynchronize(0));            // `idx` is pageable host memory about to go out of scope
	PackSource map;
	map.d_col = m->ws_bedidx.as<int32_t>();
	map.d_flip = flip ? m->ws_bedidx.as<int32_t>() + S : nullptr;
	return predict_staged_locked(m, geno, nullptr, n_samp, vote_method, H1, H2, max_prob, matching, dosage, postprob, &map, false, &hr);
}

int hibag_hip_predict_snp_major_device(hibag_hip_model *m, const int32_t *d_geno, size_t ld, int n_samp, int n_geno_snp,
	const int32_t *d_snp_col, const int32_t *d_flip, int vote_method,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, void *stream)
{
	if (int rc = check_predict_args(m, d_geno, n_samp, vote_method, d_H1, d_H2)) return rc;
	if (n_geno_snp <= 0 || ld < (size_t)std::max(n_samp, 0)) return hibag_fail(HIBAG_HIP_EINVAL, "n_geno_snp must be positive and ld >= n_samp");
	if (!d_snp_col && m->n_snp > n_geno_snp) return hibag_fail(HIBAG_HIP_EINVAL, "d_snp_col is NULL but the matrix has fewer rows than the model SNPs");
	std::lock_guard<std::mutex> g(m->lock);
	if (int rc = sticky_fault(m)) return rc;
	PackSource src;
	src.d_geno = d_geno; src.ld = std::max<size_t>(ld, 1); src.d_col = d_snp_col; src.d_flip = d_flip;
	return predict_device_locked(m, src, n_samp, vote_method, d_H1, d_H2, d_max_prob, d_matching,
		d_dosage, d_postprob, (hipStream_t)stream);
}

// ---- PLINK BED ------------------------------------------------------------

int hibag_hip_bed_flag(const char *bed_fn)
{
	if (!bed_fn) return hibag_fail(HIBAG_HIP_EINVAL, "bed file name is NULL");
	FILE *f = fopen(bed_fn, "rb");
	if (!f) return hibag_fail(HIBAG_HIP_EINVAL, "Cannot open the file %s.", bed_fn);   // src/HIBAG.cpp:1073-1074
	int mode = 0;
	const int rc = read_bed_prefix(f, &mode);
	fclose(f);
	return rc ? rc : mode;
}

int hibag_hip_conv_bed(const char *bed_fn, int n_samp, int n_snp, int n_save_snp, const int32_t *snp_flag,
	int32_t *geno)
{
	if (!snp_flag && n_snp > 0) return hibag_fail(HIBAG_HIP_EINVAL, "snp_flag is NULL");
	std::vector<int32_t> want;
	for (int j = 0; j < n_snp; j++) if (snp_flag[j]) want.push_back(j);
	if ((int)want.size() != n_save_snp)
		return hibag_fail(HIBAG_HIP_EINVAL, "snp_flag selects %zu SNPs, n_save_snp is %d", want.size(), n_save_snp);
	BedImage img;
	if (int rc = load_bed(bed_fn, n_samp, n_snp, want.data(), n_save_snp, img)) return rc;
	if (n_samp == 0 || n_save_snp == 0) return 0;
	if (!geno) return hibag_fail(HIBAG_HIP_EINVAL, "geno is NULL");
	if (hibag_hip_device_count() <= hibag_selected_device()) return hibag_fail(HIBAG_HIP_ENODEV, "no HIP device available");
	HIP_TRY(hipSetDevice(hibag_selected_device()));
	DevBuf d_rows, d_sel, d_geno;
	struct Free { DevBuf &a, &b, &c; ~Free() { a.release(); b.release(); c.release(); } } fr{d_rows, d_sel, d_geno};
	const size_t out_bytes = (size_t)n_samp * n_save_snp * sizeof(int32_t);
	if (int rc = d_rows.reserve(std::max<size_t>(img.rows.size(), 1))) return rc;
	if (int rc = d_sel.reserve((size_t)n_save_snp * sizeof(int32_t))) return rc;
	if (int rc = d_geno.reserve(out_bytes)) return rc;
	HIP_TRY(hipMemcpyAsync(d_rows.p, img.rows.data(), img.rows.size(), hipMemcpyHostToDevice, 0));
	HIP_TRY(hipMemcpyAsync(d_sel.p, img.index.data(), (size_t)n_save_snp * sizeof(int32_t), hipMemcpyHostToDevice, 0));
	hibag_launch_bed_geno(d_rows.as<uint8_t>(), img.mode, img.stride, n_samp, n_save_snp, d_sel.as<int32_t>(),
		d_geno.as<int32_t>(), 0);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(geno, d_geno.p, out_bytes, hipMemcpyDeviceToHost, 0));
	HIP_TRY(hipStreamSynchronize(0));
	return 0;
}

int hibag_hip_predict_bed(hibag_hip_model *m, const char *bed_fn, int n_samp, int n_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (int rc = check_predict_args(m, bed_fn, n_samp, vote_method, H1, H2)) return rc;
	if (!snp_col && m->n_snp > 0) return hibag_fail(HIBAG_HIP_EINVAL, "snp_col is NULL");
	BedImage img;
	if (int rc = load_bed(bed_fn, n_samp, n_snp, snp_col, m->n_snp, img)) return rc;
	if (n_samp == 0) return 0;
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	const size_t S = (size_t)std::max(m->n_snp, 1);
	std::vector<int32_t> idx(2 * S, 0);
	for (int k = 0; k < m->n_snp; k++) {
		idx[k] = img.index[k];
		idx[S + k] = flip ? (flip[k] != 0) : 0;
	}
	if (int rc = m->ws_bed.reserve(std::max<size_t>(img.rows.size(), 1))) return rc;
	if (int rc = m->ws_bedidx.reserve(idx.size() * sizeof(int32_t))) return rc;
	HIP_TRY(hipMemcpyAsync(m->ws_bed.p, img.rows.data(), img.rows.size(), hipMemcpyHostToDevice, 0));
	HIP_TRY(hipMemcpyAsync(m->ws_bedidx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, 0));
	HIP_TRY(hipStreamSynchronize(0));            // `img` and `idx` are pageable host memory about to go out of scope
	PackSource src;
	src.d_bed = m->ws_bed.as<uint8_t>();
	src.mode = img.mode;
	src.stride = img.stride;
	src.d_row = m->ws_bedidx.as<int32_t>();
	src.d_flip = m->ws_bedidx.as<int32_t>() + S;
	return predict_staged_locked(m, nullptr, &src, n_samp, vote_method, H1, H2, max_prob, matching, dosage, postprob);
}

int hibag_hip_predict_partial_device(hibag_hip_model *m, const int32_t *d_geno, int n_samp,
	double *d_partial, void *stream)
{
	if (int rc = check_predict_args(m, d_geno, n_samp, 1, nullptr, nullptr)) return rc;
	if (!d_partial) return hibag_fail(HIBAG_HIP_EINVAL, "d_partial is NULL");
	if (n_samp > batch_limit(m))
		return hibag_fail(HIBAG_HIP_EINVAL, "n_samp %d exceeds the batch limit %d of the partial entry", n_samp, batch_limit(m));
	if (n_samp == 0) return 0;
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	hipStream_t st = (hipStream_t)stream;
	if (int rc = sticky_fault(m)) return rc;
	if (int rc = workspace_enter(m, st)) return rc;
	HibagBatchView B;
	if (int rc = make_batch(m, n_samp, false, B)) return rc;
	m->timer.begin(HIBAG_HIP_K_PACK, st);
	hibag_launch_pack(m->view, B, d_geno, 0, nullptr, nullptr, m->ws_codes.as<uint8_t>(), st);
	m->timer.end(st);
	run_core(m, B, 1, d_partial, st);
	HIP_TRY(hipGetLastError());
	return workspace_leave(m, st);
}

int hibag_hip_finish_device(hibag_hip_model *m, const double *d_partial, int n_samp,
	int32_t *d_H1, int32_t *d_H2, double *d_max_prob, double *d_matching, double *d_dosage,
	double *d_postprob, void *stream)
{
	if (!m || !m->finalized) return hibag_fail(HIBAG_HIP_ESTATE, "model not finalized");
	if (!d_partial) return hibag_fail(HIBAG_HIP_EINVAL, "d_partial is NULL");
	if ((d_H1 == nullptr) != (d_H2 == nullptr)) return hibag_fail(HIBAG_HIP_EINVAL, "H1 and H2 must be given together");
	if (n_samp <= 0) return n_samp == 0 ? 0 : hibag_fail(HIBAG_HIP_EINVAL, "n_samp < 0");
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	hipStream_t st = (hipStream_t)stream;
	if (int rc = sticky_fault(m)) return rc;
	HibagBatchView B{};
	B.n_samp = n_samp; B.n_pad = round_up(n_samp, HIBAG_WAVE);
	m->timer.begin(HIBAG_HIP_K_FINISH, st);
	hibag_launch_finish(m->view, B, (double *)d_partial, d_H1, d_H2, d_max_prob, d_matching, d_dosage, d_postprob, st);
	m->timer.end(st);
	HIP_TRY(hipGetLastError());
	return 0;
}

int hibag_hip_set_timing(hibag_hip_model *m, int enabled)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	(void)hipSetDevice(m->device);
	m->timer.resolve();
	m->timer.enabled = enabled != 0;
	m->timer.mask = enabled > 1 ? ((unsigned)enabled >> 1) & 0xfu : 0xfu;     // 1: every kernel class; 2 * bits: only those
	return 0;
}

int hibag_hip_get_timing(hibag_hip_model *m, int k, double *ms_total, int64_t *launches)
{
	if (!m || k < 0 || k >= HIBAG_HIP_K_COUNT) return hibag_fail(HIBAG_HIP_EINVAL, "bad timing query");
	std::lock_guard<std::mutex> g(m->lock);
	(void)hipSetDevice(m->device);
	m->timer.resolve();
	if (ms_total) *ms_total = m->timer.ms[k];
	if (launches) *launches = m->timer.n[k];
	return sticky_fault(m);                      // (the events have been waited for: a failed hand-over of a timed launch shows here)
}

int hibag_hip_model_status(hibag_hip_model *m)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	if (m->ws_pending) { HIP_TRY(hipEventSynchronize(m->ws_done)); m->ws_pending = false; }
	return sticky_fault(m);
}

int hibag_hip_model_clear_status(hibag_hip_model *m)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	(void)take_fault(m);
	m->fault = 0;
	return 0;
}

int64_t hibag_hip_model_handover_faults(const hibag_hip_model *m) { return m ? m->fault_count : 0; }

int hibag_hip_test_inject_handover_fault(hibag_hip_model *m, int pass)
{
	if (!m || pass < 0 || pass > 2) return hibag_fail(HIBAG_HIP_EINVAL, "pass must be 0 (none), 1 or 2");
	std::lock_guard<std::mutex> g(m->lock);
	m->drop_next = pass;
	return 0;
}

// Diagnostic builds of the kernels (-DHIBAG_ACCUM_STAMPS) sum clock differences in the tail of the model's error buffer
// (entries 2000 .. of the list behind byte 16): read `n` of them and zero them.  All zero with the shipped kernels.
int hibag_hip_test_read_diag(hibag_hip_model *m, unsigned long long *out, int n)
{
	if (!m || !out || n < 0 || n > 40) return hibag_fail(HIBAG_HIP_EINVAL, "bad arguments");
	if (!m->ws_err.p) { for (int i = 0; i < n; i++) out[i] = 0; return 0; }
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	HIP_TRY(hipDeviceSynchronize());
	char *at = m->ws_err.as<char>() + 16 + 8 * 2000;
	HIP_TRY(hipMemcpy(out, at, (size_t)n * 8, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemset(at, 0, (size_t)n * 8));
	return 0;
}

int hibag_hip_reset_timing(hibag_hip_model *m)
{
	if (!m) return hibag_fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	(void)hipSetDevice(m->device);
	m->timer.reset();
	return 0;
}

} // extern "C"
