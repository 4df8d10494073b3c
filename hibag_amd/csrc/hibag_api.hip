// hibag_api.hip -- host side of libhibag_hip.so: the model container, the
// batch driver that replaces CAttrBag_Model::PredictHLA, the C ABI declared in
// include/hibag_hip.h and the TypeGPUExtProc-compatible plugin table.
//
// There is no CPU fallback here: every compute entry runs the HIP kernels or
// fails with an error code.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hibag_hip.h"
#include "hibag_device.h"
#include "hibag_kernels.h"
#include "hibag_plugin.h"

namespace {

thread_local std::string g_last_error;
thread_local int g_device = 0;

int fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}

#define HIP_TRY(expr)                                                                   \
	do {                                                                                \
		hipError_t e_ = (expr);                                                         \
		if (e_ != hipSuccess)                                                           \
			return fail(e_ == hipErrorOutOfMemory ? HIBAG_HIP_ENOMEM : HIBAG_HIP_ENODEV, \
				"%s failed: %s", #expr, hipGetErrorString(e_));                         \
	} while (0)

// Grow-only pinned host buffer (staging of the pipelined host-pointer entries).
struct PinBuf {
	void *p = nullptr;
	size_t cap = 0;
	int reserve(size_t bytes);
	void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Grow-only device buffer.
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	int reserve(size_t bytes)
	{
		if (bytes <= cap) return 0;
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		HIP_TRY(hipMalloc(&p, bytes));
		cap = bytes;
		return 0;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T *)p; }
};

int PinBuf::reserve(size_t bytes)
{
	if (bytes <= cap) return 0;
	release();
	HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
	cap = bytes;
	return 0;
}

struct HostClassifier {
	std::vector<int> snpidx;         // may be empty for plugin-built models
	int n_snp = 0;
	std::vector<uint64_t> bits;      // [H][2], bits >= n_snp cleared
	std::vector<double> freq;
	std::vector<int> hla;
};

struct KernelTimer {
	struct Pending { int k; hipEvent_t a, b; bool a_shared; };
	bool enabled = false;
	unsigned mask = 0xf;               // kernel classes that get events (bit k); the others run unobserved
	bool open = false;                 // begin() recorded something that end() has to close
	bool chainable = false;            // the last timer operation was an end() that recorded an event ...
	hipStream_t chain_stream = nullptr; // ... on this stream
	std::vector<Pending> pending;
	std::vector<hipEvent_t> pool;
	double ms[HIBAG_HIP_K_COUNT] = {0, 0, 0, 0};
	int64_t n[HIBAG_HIP_K_COUNT] = {0, 0, 0, 0};

	hipEvent_t get()
	{
		if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
		hipEvent_t e;
		(void)hipEventCreate(&e);
		return e;
	}
	// `chain`: the caller has enqueued nothing on `st` since the end() before -- that end's event is this begin's too
	// (an event record is a barrier packet of its own on the queue: five per batch instead of eight).
	void begin(int k, hipStream_t st, bool chain = false)
	{
		open = false;
		if (!enabled || !((mask >> k) & 1u)) { chainable = false; return; }
		Pending p;
		p.k = k;
		p.a_shared = chain && chainable && chain_stream == st && !pending.empty();
		p.a = p.a_shared ? pending.back().b : get();
		p.b = get();
		if (!p.a_shared) (void)hipEventRecord(p.a, st);
		pending.push_back(p);
		open = true;
		chainable = false;
	}
	void end(hipStream_t st)
	{
		if (!open) return;
		(void)hipEventRecord(pending.back().b, st);
		open = false;
		chainable = true;
		chain_stream = st;
	}
	void resolve()
	{
		for (auto &p : pending) {
			(void)hipEventSynchronize(p.b);
			float t = 0;
			if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { ms[p.k] += t; n[p.k]++; }
		}
		for (auto &p : pending) {
			if (!p.a_shared) pool.push_back(p.a);
			pool.push_back(p.b);
		}
		pending.clear();
		chainable = false;
		open = false;
	}
	void reset()
	{
		resolve();
		for (int k = 0; k < HIBAG_HIP_K_COUNT; k++) { ms[k] = 0; n[k] = 0; }
	}
	void destroy()
	{
		resolve();
		for (auto e : pool) (void)hipEventDestroy(e);
		pool.clear();
	}
};

// streams and events of the host-pointer entries' slice pipeline (predict_staged_locked)
struct StagedStreams { hipStream_t run = nullptr, in = nullptr, out = nullptr; hipEvent_t up[2] = {}, ran[2] = {}, down[2] = {}; };

} // namespace

// for the other translation units of the library (hibag_train.hip, hibag_build.hip)
int hibag_selected_device() { return g_device; }        // the calling thread's hibag_hip_set_device() choice

int hibag_fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}

struct hibag_hip_model {
	int device = 0;
	int n_hla = 0, n_snp = 0;
	bool finalized = false;
	bool have_snpidx = true;
	bool use_mfma = true;                  // matrix-core engine for classifiers with <= 112 SNPs (HIBAG_ENGINE=valu disables)
	bool use_fp4 = true;                   // its FP4 form for <= 30 SNPs (HIBAG_ENGINE=i8 keeps every classifier on the int8 form)
	std::vector<HostClassifier> cls;
	std::vector<int> snp_weight_override;   // classifier-sharded runs
	int64_t pair_evals = 0;
	double tab[HIBAG_TAB_N];

	// device model
	DevBuf d_int, d_stream, d_tile, d_tab, d_blk, d_pfac, d_phdr, d_parow;
	HibagModelView view{};
	int mask_rows = 0, bt_rows = 0, cell_rows = 0;
	size_t stream_bytes = 0;

	// per-batch workspace (grow-only)
	DevBuf ws_planes, ws_cw, ws_tot, ws_inv, ws_winv, ws_part, ws_best, ws_vrec, ws_geno, ws_out, ws_codes, ws_bt, ws_bias, ws_cells, ws_sync;
	std::vector<int> engine_of, steps_of;  // per classifier: HIBAG_HIP_ENGINE_* and K steps, as finalized
	int store_mode = 0;                    // which cell sums pass 1 stores for pass 2 (HibagModelView::store_cells)
	int64_t second_pass_pairs = 0;         // haplotype pairs per sample pass 2 evaluates again
	HibagSideStream side;                  // second stream for pass 1 of the classifiers with several K steps (created at finalize if any)
	uint32_t epoch = 0;                    // batch counter for the hand-over flags (HibagBatchView::epoch)
	int *h_err = nullptr;                  // host-mapped error word of the hand-overs
	DevBuf ws_err;                         // its device twin: epoch of the last batch with a failed hand-over (HibagBatchView::err_dev)
	// A failed hand-over: `fault` is sticky until hibag_hip_model_clear_status(); from the first one on
	// the model launches without hand-overs (`no_chunks`: every work item undivided -- nothing left that could fail).
	int fault = 0;
	int64_t fault_count = 0;
	bool no_chunks = false;
	int drop_next = 0;                     // fault injection (hibag_hip_test_inject_handover_fault): pass whose first hand-over the next batch drops
	uint32_t spin_limit = 1u << 19;        // polls a waiting workgroup makes before it gives up (set at finalize from the longest item)
	// The workspace is one per model: calls on different streams are chained on the device through this event, each
	// waits for the one enqueued before it.
	hipEvent_t ws_done = nullptr;
	bool ws_pending = false;
	StagedStreams staged;                  // the host-pointer entries' slice pipeline (created on first use)
	PinBuf pin_geno, pin_out;              // its pinned staging buffers (two slices each)
	bool staged_ready = false;
	// PLINK BED payload + SNP map of hibag_hip_predict_bed
	DevBuf ws_bed, ws_bedidx;

	KernelTimer timer;
	std::mutex lock;

	~hibag_hip_model()
	{
		(void)hipSetDevice(device);
		timer.destroy();
		if (h_err) (void)hipHostFree(h_err);
		if (ws_done) (void)hipEventDestroy(ws_done);
		pin_geno.release(); pin_out.release();
		for (hipStream_t st : {staged.run, staged.in, staged.out}) if (st) (void)hipStreamDestroy(st);
		for (int i = 0; i < 2; i++)
			for (hipEvent_t e : {staged.up[i], staged.ran[i], staged.down[i]}) if (e) (void)hipEventDestroy(e);
		if (side.fork) (void)hipEventDestroy(side.fork);
		if (side.join) (void)hipEventDestroy(side.join);
		if (side.stream) (void)hipStreamDestroy(side.stream);
		for (DevBuf *b : {&d_int, &d_stream, &d_tile, &d_tab, &d_blk, &d_pfac, &d_phdr, &d_parow, &ws_bt, &ws_bias, &ws_cells, &ws_sync, &ws_err, &ws_planes, &ws_cw, &ws_tot, &ws_inv, &ws_winv,
		                  &ws_part, &ws_best, &ws_vrec, &ws_geno, &ws_out, &ws_codes, &ws_bed, &ws_bedidx})
			b->release();
	}
};

namespace {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

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
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
	if (m->finalized) return fail(HIBAG_HIP_ESTATE, "model already finalized");
	if (n_snp_c < 0 || n_snp_c > HIBAG_HIP_MAX_SNP_IN_CLASSIFIER)
		return fail(HIBAG_HIP_EINVAL, "there are too many SNP markers in a classifier (%d > %d).",
			n_snp_c, HIBAG_HIP_MAX_SNP_IN_CLASSIFIER);
	if (n_haplo < 0 || (n_haplo > 0 && (!freq || !hla)))
		return fail(HIBAG_HIP_EINVAL, "invalid haplotype list");
	if (snpidx)
		for (int i = 0; i < n_snp_c; i++)
			if (snpidx[i] < 0 || snpidx[i] >= m->n_snp)
				return fail(HIBAG_HIP_EINVAL, "SNP index %d out of range [0,%d)", snpidx[i], m->n_snp);
	for (int i = 0; i < n_haplo; i++) {
		if (hla[i] < 0 || hla[i] >= m->n_hla)
			return fail(HIBAG_HIP_EINVAL, "HLA allele index %d out of range [0,%d)", hla[i], m->n_hla);
		if (i > 0 && hla[i] < hla[i - 1])
			return fail(HIBAG_HIP_EINVAL, "haplotypes must be grouped by ascending HLA allele index");
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
void build_tiles(int P, const std::vector<uint64_t> &cell_work, std::vector<int> &tile_p0, std::vector<int> &tile_n)
{
	uint64_t total = 0;
	for (int p = 0; p < P; p++) total += cell_work[p] + 1;
	const int min_tiles = (P + HIBAG_TILE - 1) / HIBAG_TILE;
	const uint64_t target = std::max<uint64_t>(1, total / (uint64_t)std::max(min_tiles, 1));
	// A tile is a wavefront of pass 2, and four tiles make a workgroup: every tile beyond the minimum
	// is another wavefront per sample group (and possibly another, mostly empty, workgroup).  A tile is
	// therefore closed early for balance only while the cells it leaves unused still fit into the
	// minimum number of tiles.
	int slack = min_tiles * HIBAG_TILE - P;
	tile_p0.clear(); tile_n.clear();
	int p = 0;
	while (p < P) {
		int n = 0;
		uint64_t w = 0;
		while (p + n < P && n < HIBAG_TILE) {
			const uint64_t cw = cell_work[p + n] + 1;
			if (n > 0 && w + cw > target + target / 4 && HIBAG_TILE - n <= slack) break;
			w += cw; n++;
		}
		if (p + n < P) slack -= HIBAG_TILE - n;
		tile_p0.push_back(p); tile_n.push_back(n);
		p += n;
	}
}

int finalize_model(hibag_hip_model *m)
{
	if (m->finalized) return fail(HIBAG_HIP_ESTATE, "model already finalized");
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
					if (n > 0xFFFFFFull * HIBAG_CHUNK) return fail(HIBAG_HIP_EINVAL, "an allele pair of classifier %d has too many haplotype pairs", c);
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
	if (hap.size() * sizeof(uint32_t) > 0x7FFFFF00ull) return fail(HIBAG_HIP_EINVAL, "the model's haplotype tables exceed 2 GB");

	std::vector<int> tile_p0, tile_n;
	build_tiles(P, cell_work, tile_p0, tile_n);
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
			if (off > 0xFFFFFFFFull) return fail(HIBAG_HIP_EINVAL, "classifier %d has too many haplotype pairs", c);
			me[1] = (uint32_t)off;
			int k = 0;
			uint64_t jpack = 0;
			for (int j = 0; j < tile_n[t]; j++) {
				const uint32_t n = cell_chunks[c][tile_p0[t] + j];
				if (n > 0xFFFFFFu) return fail(HIBAG_HIP_EINVAL, "an allele pair of classifier %d has too many haplotype pairs", c);
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
				uint32_t closes = 0;
				if (b < nb)
					for (int i = 0; i < HIBAG_PLIST_DWORDS; i++) closes += plist[first + (size_t)b * HIBAG_PLIST_DWORDS + i] >> 31;
				const int nsb = std::max(0, std::min(HIBAG_STORED_PER_VISIT, ns - HIBAG_STORED_PER_VISIT * b));
				if (c > 0xFFFF) return fail(HIBAG_HIP_EINVAL, "too many classifiers (%d) for the second pass's block headers", C);
				// (pass 2 requests the operand rows of every block it passes, also of blocks that only carry stored sums: a
				// classifier of the vector engine has no rows -- bt_row[c] is then the NEXT classifier's first row, or one past
				// the last row of the batch's array for the model's last classifiers: rows 0 and 1 instead)
				const uint32_t bt = (uint32_t)(HIBAG_ENGINE_ROWS(engine[c], n_snp_c[c]) > 0 ? bt_row[c] : 0);
				if (bt > 0xFFFFu) return fail(HIBAG_HIP_EINVAL, "too many classifiers for the matrix engine's operand rows");
				const uint32_t h[8] = {
					(uint32_t)c | (bt << 16), srow | ((uint32_t)nsb << 25),
					0u, 0u,                                   // (the next block's first two words: filled in below)
					(uint32_t)jp, (uint32_t)(jp >> 32),
					(uint32_t)jps, 0u};
				ehdr.insert(ehdr.end(), h, h + 8);
				jp = closes >= 16 ? 0 : jp >> (4 * closes);
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
			if (bt_row[c] > 0xFFFF) return fail(HIBAG_HIP_EINVAL, "too many classifiers for the matrix engine's operand rows");
			(void)me;
			const int k_last = n_snp_c[c] - HIBAG_FP4_STEP_SNPS * (n_step[c] - 1);        // SNPs of the last K step (all of them for one step)
			r[0] = (uint32_t)mfma_nkb[c] | ((uint32_t)k_last << 2 & 0xFCu) | (tile_nlist[(size_t)c * n_tile + t] << 8) |
			       ((uint32_t)(n_step[c] - 1) << 13) | ((uint32_t)bt_row[c] << 16);
			r[1] = hap_off[c];
			r[2] = (uint32_t)off; r[3] = (uint32_t)(off >> 32);
			r[4] = seg_nblk[(size_t)c * n_tile + t];
			// first stored row of the (classifier, tile) among all stored cells of the model
			const uint64_t row = (uint64_t)cell_row[c] + tile_k0[(size_t)c * n_tile + t];
			if (row >> 27) return fail(HIBAG_HIP_EINVAL, "the model has too many allele pairs to store their sums");
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
		return fail(HIBAG_HIP_EINVAL, "the model stores too many cell sums per sample (%d) for the second pass", cell_row[C]);
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
		const size_t n_gq = ((size_t)(n_pad / HIBAG_WAVE + 7) / 8 + 3) / 4;
		const size_t n_flag2 = 8 * n_gq * (size_t)std::max(m->view.n_tile, 1);
		const size_t n_flag1 = (size_t)((n_pad / HIBAG_WAVE + 3) / 4) * (size_t)std::max(std::max(m->view.n_item_whole, m->view.n_item_split), 1);
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
	B.sync_total = B.sync + 8 * (((size_t)(n_pad / HIBAG_WAVE + 7) / 8 + 3) / 4) * (size_t)std::max(m->view.n_tile, 1);
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
	T.begin(HIBAG_HIP_K_ACCUM, st, true);
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
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
	if (!m->finalized) return fail(HIBAG_HIP_ESTATE, "model not finalized");
	if (vote_method < 1 || vote_method > 2)
		return fail(HIBAG_HIP_EINVAL, "Invalid 'vote_method'.");   // src/LibHLA.cpp:2321-2322
	if (n_samp < 0) return fail(HIBAG_HIP_EINVAL, "n_samp < 0");
	if (n_samp > 0 && !geno) return fail(HIBAG_HIP_EINVAL, "geno is NULL");
	if ((H1 == nullptr) != (H2 == nullptr)) return fail(HIBAG_HIP_EINVAL, "H1 and H2 must be given together");
	if (!m->have_snpidx)
		return fail(HIBAG_HIP_ESTATE, "model was built without SNP indices: raw genotypes cannot be packed");
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
		return fail(m->fault, "a hand-over between workgroups failed in an earlier launch on this model: the outputs of that "
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

int predict_staged_locked(hibag_hip_model *m, const int32_t *geno, const PackSource *bed, int n_samp, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob,
	const PackSource *map = nullptr, bool is_retry = false)
{
	// `map`: geno is the cohort's own matrix (map->row_len SNPs per sample); map->d_col / d_flip sit on the device
	// A device-pointer launch still running on another stream may yet fail a hand-over: wait for it, so that its fault
	// becomes the model's sticky status (its caller's to see) instead of being taken for this call's own and repaired away.
	if (m->ws_pending && m->ws_done && !is_retry) HIP_TRY(hipEventSynchronize(m->ws_done));
	if (int rc = sticky_fault(m)) return rc;
	const size_t P = (size_t)m->view.n_cell, nh = (size_t)m->n_hla, S = map ? (size_t)map->row_len : (size_t)m->n_snp;
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
		if (H1) {
			HIP_TRY(hipMemcpyAsync(H1 + s0, o + o_h1, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(H2 + s0, o + o_h2, (size_t)n * 4, hipMemcpyDeviceToHost, st));
		}
		if (max_prob) HIP_TRY(hipMemcpyAsync(max_prob + s0, o + o_mp, (size_t)n * 8, hipMemcpyDeviceToHost, st));
		if (matching) HIP_TRY(hipMemcpyAsync(matching + s0, o + o_mt, (size_t)n * 8, hipMemcpyDeviceToHost, st));
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
	if (trace) {
		tr[5] = now();
		if (!piped) fprintf(stderr, "[hibag staged] n=%d upload %.3f  enqueue %.3f  kernels %.3f  download calls %.3f  final sync %.3f ms\n", n_samp,
			tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4]);
		else fprintf(stderr, "[hibag staged] n=%d in %d slices of %d: %.3f ms\n", n_samp, n_slice, slice, tr[5] - tr[0]);
	}
	if (take_fault(m)) {
		// poisoned outputs: once more, now without hand-overs (take_fault switched them off) -- never returned to the caller
		if (is_retry) return fail(HIBAG_HIP_EHANDOVER, "a hand-over between workgroups failed in a launch without hand-overs");
		return predict_staged_locked(m, geno, bed, n_samp, vote_method, H1, H2, max_prob, matching, dosage, postprob, map, true);
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
		return fail(HIBAG_HIP_EINVAL, "Invalid prefix in the PLINK BED file.");   // src/HIBAG.cpp:1077-1078, :1112-1113
	*mode = prefix[2];
	return 0;
}

// want[n_want]: BED SNP indices (0-based, -1 = none).
int load_bed(const char *fn, int n_samp, int n_snp, const int32_t *want, int n_want, BedImage &img)
{
	if (!fn) return fail(HIBAG_HIP_EINVAL, "bed file name is NULL");
	if (n_samp < 0 || n_snp < 0) return fail(HIBAG_HIP_EINVAL, "negative dimensions (n_samp=%d, n_snp=%d)", n_samp, n_snp);
	FILE *f = fopen(fn, "rb");
	if (!f) return fail(HIBAG_HIP_EINVAL, "Fail to open the file \"%s\".", fn);   // src/HIBAG.cpp:1106-1107
	struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
	if (int rc = read_bed_prefix(f, &img.mode)) return rc;
	for (int j = 0; j < n_want; j++)
		if (want[j] >= n_snp) return fail(HIBAG_HIP_EINVAL, "SNP index %d outside the BED file's %d SNPs", want[j], n_snp);
	img.index.assign(n_want, -1);
	const char *short_msg = "the PLINK BED file holds fewer than %d x %d genotypes";
	if (img.mode == 0) {
		img.stride = ((size_t)n_snp + 3) / 4;
		img.rows.resize(img.stride * (size_t)n_samp);
		if (!img.rows.empty() && fread(img.rows.data(), 1, img.rows.size(), f) != img.rows.size())
			return fail(HIBAG_HIP_EINVAL, short_msg, n_samp, n_snp);
		for (int j = 0; j < n_want; j++) img.index[j] = want[j];
	} else {
		img.stride = ((size_t)n_samp + 3) / 4;
		int n_row = 0;
		for (int j = 0; j < n_want; j++) if (want[j] >= 0) n_row++;
		img.rows.resize(img.stride * (size_t)n_row);
		int r = 0;
		for (int j = 0; j < n_want; j++) {
			if (want[j] < 0) continue;
			if (fseeko(f, (off_t)3 + (off_t)img.stride * want[j], SEEK_SET) != 0 ||
				(img.stride && fread(img.rows.data() + img.stride * (size_t)r, 1, img.stride, f) != img.stride))
				return fail(HIBAG_HIP_EINVAL, short_msg, n_samp, n_snp);
			img.index[j] = r++;
		}
	}
	return 0;
}

} // namespace

// ===========================================================================
// C ABI

extern "C" {

int hibag_hip_abi_version(void) { return HIBAG_HIP_ABI_VERSION; }

const char *hibag_hip_last_error(void) { return g_last_error.c_str(); }

int hibag_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

int hibag_hip_set_device(int device)
{
	const int n = hibag_hip_device_count();
	if (device < 0 || device >= n)
		return fail(HIBAG_HIP_ENODEV, "HIP device %d not available (%d visible)", device, n);
	g_device = device;
	return 0;
}

int hibag_hip_set_kernel_target(const char *target, char *info, size_t info_len)
{
	if (!target || strcmp(target, "hip") != 0)
		return fail(HIBAG_HIP_EINVAL, "this library implements the kernel target \"hip\" only (got \"%s\")",
			target ? target : "(null)");
	const int n = hibag_hip_device_count();
	if (n <= 0 || g_device >= n) return fail(HIBAG_HIP_ENODEV, "no HIP device available");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, g_device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(HIBAG_HIP_ENODEV, "device %d is %s; the kernels are built for gfx950 only", g_device, prop.gcnArchName);
	if (info && info_len)
		snprintf(info, info_len, "HIP, %s, %s, %d CUs", prop.gcnArchName, prop.name, prop.multiProcessorCount);
	return 0;
}

hibag_hip_model *hibag_hip_model_new(int n_hla, int n_snp)
{
	if (n_hla <= 0 || n_hla > 46340 || n_snp < 0) {
		fail(HIBAG_HIP_EINVAL, "invalid model dimensions (n_hla=%d, n_snp=%d)", n_hla, n_snp);
		return nullptr;
	}
	hibag_hip_model *m = new (std::nothrow) hibag_hip_model;
	if (!m) { fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	m->device = g_device;
	const char *engine = getenv("HIBAG_ENGINE");         // "valu": bit logic + popcount on the vector ALU for every classifier
	m->use_mfma = !(engine && strcmp(engine, "valu") == 0);
	m->use_fp4 = !(engine && strcmp(engine, "i8") == 0);
	m->n_hla = n_hla;
	m->n_snp = n_snp;
	build_table(m->tab);
	return m;
}

int hibag_hip_model_add_classifier(hibag_hip_model *m, int n_snp_c, const int32_t *snpidx,
	int n_haplo, const double *freq, const int32_t *hla, const char *const *haplo)
{
	if (int rc = check_classifier_args(m, n_snp_c, snpidx, n_haplo, freq, hla)) return rc;
	if (n_snp_c > 0 && !snpidx) return fail(HIBAG_HIP_EINVAL, "snpidx is NULL");
	if (n_haplo > 0 && !haplo) return fail(HIBAG_HIP_EINVAL, "haplo is NULL");
	std::vector<uint64_t> bits((size_t)n_haplo * 2, 0);
	for (int i = 0; i < n_haplo; i++) {
		const char *s = haplo[i];
		const size_t len = s ? strlen(s) : 0;
		if (len > HIBAG_HIP_MAX_SNP_IN_CLASSIFIER)   // src/LibHLA.cpp:328-329
			return fail(HIBAG_HIP_EINVAL, "THaplotype::StrToHaplo, the input string is too long.");
		if ((int)len != n_snp_c)
			return fail(HIBAG_HIP_EINVAL, "haplotype %d has %zu alleles, expected %d", i, len, n_snp_c);
		for (size_t j = 0; j < len; j++) {
			if (s[j] == '1') bits[2 * (size_t)i + (j >> 6)] |= (uint64_t)1 << (j & 63);
			else if (s[j] != '0')                    // src/LibHLA.cpp:333-334
				return fail(HIBAG_HIP_EINVAL, "THaplotype::StrToHaplo, the input string should be '0' or '1'");
		}
	}
	push_classifier(m, n_snp_c, snpidx, n_haplo, freq, hla, std::move(bits));
	return 0;
}

int hibag_hip_model_add_classifier_packed(hibag_hip_model *m, int n_snp_c, const int32_t *snpidx,
	int n_haplo, const double *freq, const int32_t *hla, const uint64_t *bits_in)
{
	if (int rc = check_classifier_args(m, n_snp_c, snpidx, n_haplo, freq, hla)) return rc;
	if (n_haplo > 0 && !bits_in) return fail(HIBAG_HIP_EINVAL, "bits is NULL");
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
	if (!m || !snp_weight) return fail(HIBAG_HIP_EINVAL, "NULL argument");
	if (m->finalized) return fail(HIBAG_HIP_ESTATE, "model already finalized");
	m->snp_weight_override.assign(snp_weight, snp_weight + std::max(m->n_snp, 1));
	return 0;
}

int hibag_hip_model_finalize(hibag_hip_model *m)
{
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
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
	if (!m || !out) return fail(HIBAG_HIP_EINVAL, "NULL argument");
	memcpy(out, m->tab, sizeof(m->tab));
	return 0;
}

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

// ---- several devices ------------------------------------------------------------------------------------

hibag_hip_model *hibag_hip_model_replicate(const hibag_hip_model *src, int device)
{
	if (!src) { fail(HIBAG_HIP_EINVAL, "model is NULL"); return nullptr; }
	const int n = hibag_hip_device_count();
	if (device < 0 || device >= n) { fail(HIBAG_HIP_ENODEV, "HIP device %d not available (%d visible)", device, n); return nullptr; }
	hibag_hip_model *m = new (std::nothrow) hibag_hip_model;
	if (!m) { fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
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
	if (!src) { fail(HIBAG_HIP_EINVAL, "model is NULL"); return nullptr; }
	int first = 0, count = 0;
	if (hibag_hip_shard_bounds((int)src->cls.size(), n_shards, shard, &first, &count)) return nullptr;
	const int n = hibag_hip_device_count();
	if (device < 0 || device >= n) { fail(HIBAG_HIP_ENODEV, "HIP device %d not available (%d visible)", device, n); return nullptr; }
	hibag_hip_model *m = new (std::nothrow) hibag_hip_model;
	if (!m) { fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
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
	} catch (...) { delete m; fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	if (src->finalized && hibag_hip_model_finalize(m)) { delete m; return nullptr; }
	return m;
}

// samples per batch of the device-pointer entries that take ONE batch (hibag_hip_predict_partial_device); 0 = not finalized
int hibag_hip_model_batch_limit(const hibag_hip_model *m) { return m && m->finalized ? batch_limit(m) : 0; }

int hibag_hip_multi_slice(int n_samp, int n_models, int i, int *first, int *count)
{
	if (n_samp < 0 || n_models <= 0 || i < 0 || i >= n_models) return fail(HIBAG_HIP_EINVAL, "bad slice query (n_samp=%d, n_models=%d, i=%d)", n_samp, n_models, i);
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
	if (!models || n_models <= 0) return fail(HIBAG_HIP_EINVAL, "no models given");
	for (int i = 0; i < n_models; i++) {
		if (int rc = check_predict_args(models[i], geno, n_samp, vote_method, H1, H2)) return rc;
		if (models[i]->n_hla != models[0]->n_hla || models[i]->n_snp != models[0]->n_snp || models[i]->cls.size() != models[0]->cls.size())
			return fail(HIBAG_HIP_EINVAL, "model %d is not a replica of model 0", i);
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
		return fail(HIBAG_HIP_ENOMEM, "hibag_hip_predict_multi: could not start a host thread per replica");
	}
	if (code) return fail(code, "replica %d (device %d): %s", who, models[who]->device, text.c_str());
	return 0;
}

int hibag_hip_predict_mapped(hibag_hip_model *m, const int32_t *geno, int n_samp, int n_geno_snp,
	const int32_t *snp_col, const int32_t *flip, int vote_method,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (int rc = check_predict_args(m, geno, n_samp, vote_method, H1, H2)) return rc;
	if (n_geno_snp <= 0) return fail(HIBAG_HIP_EINVAL, "n_geno_snp must be positive");
	if (!snp_col && m->n_snp > 0) return fail(HIBAG_HIP_EINVAL, "snp_col is NULL");
	for (int k = 0; k < m->n_snp; k++)
		if (snp_col[k] >= n_geno_snp) return fail(HIBAG_HIP_EINVAL, "snp_col[%d] = %d outside the %d SNPs of the genotype matrix", k, snp_col[k], n_geno_snp);
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
	if (n_geno_snp <= 0 || !d_snp_col) return fail(HIBAG_HIP_EINVAL, "n_geno_snp must be positive and d_snp_col given");
	std::lock_guard<std::mutex> g(m->lock);
	if (int rc = sticky_fault(m)) return rc;
	PackSource src;
	src.d_geno = d_geno; src.row_len = n_geno_snp; src.d_col = d_snp_col; src.d_flip = d_flip;
	return predict_device_locked(m, src, n_samp, vote_method, d_H1, d_H2, d_max_prob, d_matching,
		d_dosage, d_postprob, (hipStream_t)stream);
}

// ---- PLINK BED ------------------------------------------------------------

int hibag_hip_bed_flag(const char *bed_fn)
{
	if (!bed_fn) return fail(HIBAG_HIP_EINVAL, "bed file name is NULL");
	FILE *f = fopen(bed_fn, "rb");
	if (!f) return fail(HIBAG_HIP_EINVAL, "Cannot open the file %s.", bed_fn);   // src/HIBAG.cpp:1073-1074
	int mode = 0;
	const int rc = read_bed_prefix(f, &mode);
	fclose(f);
	return rc ? rc : mode;
}

int hibag_hip_conv_bed(const char *bed_fn, int n_samp, int n_snp, int n_save_snp, const int32_t *snp_flag,
	int32_t *geno)
{
	if (!snp_flag && n_snp > 0) return fail(HIBAG_HIP_EINVAL, "snp_flag is NULL");
	std::vector<int32_t> want;
	for (int j = 0; j < n_snp; j++) if (snp_flag[j]) want.push_back(j);
	if ((int)want.size() != n_save_snp)
		return fail(HIBAG_HIP_EINVAL, "snp_flag selects %zu SNPs, n_save_snp is %d", want.size(), n_save_snp);
	BedImage img;
	if (int rc = load_bed(bed_fn, n_samp, n_snp, want.data(), n_save_snp, img)) return rc;
	if (n_samp == 0 || n_save_snp == 0) return 0;
	if (!geno) return fail(HIBAG_HIP_EINVAL, "geno is NULL");
	if (hibag_hip_device_count() <= g_device) return fail(HIBAG_HIP_ENODEV, "no HIP device available");
	HIP_TRY(hipSetDevice(g_device));
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
	if (!snp_col && m->n_snp > 0) return fail(HIBAG_HIP_EINVAL, "snp_col is NULL");
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
	if (!d_partial) return fail(HIBAG_HIP_EINVAL, "d_partial is NULL");
	if (n_samp > batch_limit(m))
		return fail(HIBAG_HIP_EINVAL, "n_samp %d exceeds the batch limit %d of the partial entry", n_samp, batch_limit(m));
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
	if (!m || !m->finalized) return fail(HIBAG_HIP_ESTATE, "model not finalized");
	if (!d_partial) return fail(HIBAG_HIP_EINVAL, "d_partial is NULL");
	if ((d_H1 == nullptr) != (d_H2 == nullptr)) return fail(HIBAG_HIP_EINVAL, "H1 and H2 must be given together");
	if (n_samp <= 0) return n_samp == 0 ? 0 : fail(HIBAG_HIP_EINVAL, "n_samp < 0");
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
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	(void)hipSetDevice(m->device);
	m->timer.resolve();
	m->timer.enabled = enabled != 0;
	m->timer.mask = enabled > 1 ? ((unsigned)enabled >> 1) & 0xfu : 0xfu;     // 1: every kernel class; 2 * bits: only those
	return 0;
}

int hibag_hip_get_timing(hibag_hip_model *m, int k, double *ms_total, int64_t *launches)
{
	if (!m || k < 0 || k >= HIBAG_HIP_K_COUNT) return fail(HIBAG_HIP_EINVAL, "bad timing query");
	std::lock_guard<std::mutex> g(m->lock);
	(void)hipSetDevice(m->device);
	m->timer.resolve();
	if (ms_total) *ms_total = m->timer.ms[k];
	if (launches) *launches = m->timer.n[k];
	return sticky_fault(m);                      // (the events have been waited for: a failed hand-over of a timed launch shows here)
}

int hibag_hip_model_status(hibag_hip_model *m)
{
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	if (m->ws_pending) { HIP_TRY(hipEventSynchronize(m->ws_done)); m->ws_pending = false; }
	return sticky_fault(m);
}

int hibag_hip_model_clear_status(hibag_hip_model *m)
{
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	(void)take_fault(m);
	m->fault = 0;
	return 0;
}

int64_t hibag_hip_model_handover_faults(const hibag_hip_model *m) { return m ? m->fault_count : 0; }

int hibag_hip_test_inject_handover_fault(hibag_hip_model *m, int pass)
{
	if (!m || pass < 0 || pass > 2) return fail(HIBAG_HIP_EINVAL, "pass must be 0 (none), 1 or 2");
	std::lock_guard<std::mutex> g(m->lock);
	m->drop_next = pass;
	return 0;
}

// Diagnostic builds of the kernels (-DHIBAG_ACCUM_STAMPS) sum clock differences in the tail of the model's error buffer
// (entries 2000 .. of the list behind byte 16): read `n` of them and zero them.  All zero with the shipped kernels.
int hibag_hip_test_read_diag(hibag_hip_model *m, unsigned long long *out, int n)
{
	if (!m || !out || n < 0 || n > 40) return fail(HIBAG_HIP_EINVAL, "bad arguments");
	if (!m->ws_err.p) { for (int i = 0; i < n; i++) out[i] = 0; return 0; }
	std::lock_guard<std::mutex> g(m->lock);
	HIP_TRY(hipSetDevice(m->device));
	HIP_TRY(hipDeviceSynchronize());
	char *at = m->ws_err.as<char>() + 16 + 8 * 2000;
	HIP_TRY(hipMemcpy(out, at, (size_t)n * 8, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemset(at, 0, (size_t)n * 8));
	return 0;
}

int hibag_hip_model_engine(const hibag_hip_model *m, int classifier, int *engine, int *k_steps)
{
	if (!m || !m->finalized) return fail(HIBAG_HIP_ESTATE, "model not finalized");
	if (classifier < 0 || classifier >= (int)m->cls.size()) return fail(HIBAG_HIP_EINVAL, "classifier %d out of range", classifier);
	if (engine) *engine = m->engine_of[classifier];
	if (k_steps) *k_steps = m->steps_of[classifier];
	return 0;
}

int hibag_hip_reset_timing(hibag_hip_model *m)
{
	if (!m) return fail(HIBAG_HIP_EINVAL, "model is NULL");
	std::lock_guard<std::mutex> g(m->lock);
	(void)hipSetDevice(m->device);
	m->timer.reset();
	return 0;
}

} // extern "C"

// ===========================================================================
// HIBAG plugin table: layout-compatible with HLA_LIB::TypeGPUExtProc
// (inst/include/LibHLA_ext.h:358-388).  The host calls predict_init once per
// PredictHLA (src/LibHLA.cpp:2498-2523), predict_avg_prob once per sample with
// nthread = 1 (:2433-2441) and predict_done from a destructor (:2525-2531).
// A failure cannot be returned through these void signatures; like a C++
// plugin would, it throws `const char *`, which the host's CORE_CATCH turns
// into an R error (src/HIBAG.cpp:41-60).

namespace {

// The predict entries live in hibag_sample.hip (a per-sample form of the path: lane = allele-pair cell), the build entries in
// hibag_build.hip.  One model and one training state per process at a time, like the reference's single staging buffer.
const PluginTable g_plugin_table = {
	hibag_build_init, hibag_build_done, hibag_build_set_bootstrap, hibag_build_haplomatch,
	hibag_build_set_haplo_geno, hibag_build_acc_oob, hibag_build_acc_ib,
	hibag_sample_init, hibag_sample_done, hibag_sample_avg_prob,
};

} // namespace

extern "C" const void *hibag_hip_gpu_ext_proc(void) { return &g_plugin_table; }

// predict_avg_prob calls since the last predict_init whose full-width launch could not get all its workgroups resident (the
// device is shared) and that were therefore repeated on a single workgroup; 0 on a device the process has to itself.
extern "C" long long hibag_hip_plugin_degraded_calls(void) { return hibag_sample_degraded_calls(); }

// The loop an unmodified HIBAG runs around predict_avg_prob (src/LibHLA.cpp:2362-2411 with :2433-2441), in C like the host's: for
// every sample one call through the table with its packed genotypes and weights, then BestGuessEnsemble's scan of the posterior
// (:1549-1566: first strict maximum in cell order, -1 when nothing is positive).  Measurement and tests: the time of the n_samp
// calls as a compiled host sees them (no interpreter between the calls), and the per-sample best cell to check them with.
extern "C" int hibag_hip_test_time_avg_prob(const void *geno, const double *weight, int n_samp, int n_classifier, int n_cell,
	int32_t *best_cell, double *matching, double *seconds)
{
	if (!geno || !weight || n_samp < 0 || n_classifier < 0 || n_cell <= 0) return fail(HIBAG_HIP_EINVAL, "bad arguments");
	std::vector<double> prob;
	try { prob.assign((size_t)n_cell, 0.0); } catch (...) { return fail(HIBAG_HIP_ENOMEM, "out of host memory"); }
	const PluginGenotype *g = (const PluginGenotype *)geno;
	double match = 0;
	const auto t0 = std::chrono::steady_clock::now();
	try {
		for (int i = 0; i < n_samp; i++) {
			g_plugin_table.predict_avg_prob(g + (size_t)i * n_classifier, weight + (size_t)i * n_classifier, prob.data(), &match);
			double best = 0;
			int cell = -1;
			for (int p = 0; p < n_cell; p++)
				if (best < prob[p]) { best = prob[p]; cell = p; }
			if (best_cell) best_cell[i] = cell;
			if (matching) matching[i] = match;
		}
	} catch (const char *msg) {
		return fail(HIBAG_HIP_ENODEV, "%s", msg);
	}
	if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return 0;
}
