// hibag_internal.h -- what the host-side translation units of libhibag_hip.so share: the model container behind the opaque
// `hibag_hip_model` of include/hibag_hip.h, its device / pinned buffers, the per-kernel timers, and the few functions that
// cross the files:
//   hibag_api.hip      error state, device selection, kernel target, the plugin table
//   hibag_model.hip    the model: classifiers in, the device layout out (hibag_hip_model_new ... _finalize, replicas, shards)
//   hibag_predict.hip  the batch driver that replaces CAttrBag_Model::PredictHLA: workspace, kernel sequence, host-pointer
//                      pipeline, BED input, partial sums, launch status, timing
#ifndef HIBAG_INTERNAL_H_
#define HIBAG_INTERNAL_H_

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

// records the calling thread's last error (hibag_hip_last_error) and returns `code` (hibag_api.hip)
int hibag_fail(int code, const char *fmt, ...);
int hibag_selected_device();                 // the calling thread's hibag_hip_set_device() choice

#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess)                                                                 \
			return hibag_fail(e_ == hipErrorOutOfMemory ? HIBAG_HIP_ENOMEM : HIBAG_HIP_ENODEV, \
				"%s failed: %s", #expr, hipGetErrorString(e_));                               \
	} while (0)

namespace hibag_detail {

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

inline int PinBuf::reserve(size_t bytes)
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


inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

} // namespace hibag_detail
using namespace hibag_detail;

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
	DevBuf ws_thrash;                      // HIBAG_DEBUG_THRASH_MB (hibag_predict.hip run_core): scratch a measurement overwrites between the passes

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
		                  &ws_part, &ws_best, &ws_vrec, &ws_geno, &ws_out, &ws_codes, &ws_bed, &ws_bedidx, &ws_thrash})
			b->release();
	}
};

namespace hibag_detail {

void build_table(double *tab);                               // hibag_model.hip: exp(d * log(1e-5)), the host libm's
int finalize_model(hibag_hip_model *m);                      // hibag_model.hip
int batch_limit(const hibag_hip_model *m);                   // hibag_predict.hip: samples per batch (workspace bound)

} // namespace hibag_detail

#endif
