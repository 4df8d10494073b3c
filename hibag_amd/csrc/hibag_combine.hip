// hibag_combine.hip -- the combiner of hibag_combine.h: per device two lanes (short operations, EM fits), each a list of
// pending operations, three streams (batches in flight) and a leader-takes-all protocol; the batch's copies as one kernel
// each way; the leader's sleeping wait; the host-thread budget of shared trainers.
// Compiled with -fgpu-default-stream=per-thread like the files whose work it launches: stream 0 below is the CALLING
// THREAD's own stream.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <mutex>
#include <sys/prctl.h>
#include <vector>
#include "hibag_combine.h"

namespace {

HibagOpLaunch g_launch[HIBAG_OP_KINDS] = {};
thread_local bool t_shared = false;
thread_local bool t_has_slot = false;
thread_local char t_msg[300];

[[noreturn]] void combine_throw(const char *what, hipError_t e)
{
	snprintf(t_msg, sizeof(t_msg), "HIBAG HIP trainer: %s: %s", what, hipGetErrorString(e));
	throw (const char *)t_msg;
}

// ---- host-thread budget -------------------------------------------------------------------------------
std::mutex g_slot_m;
std::condition_variable g_slot_cv;
int g_slot_cap = 0, g_slot_used = 0;          // cap 0: no limit

void slot_acquire()
{
	if (t_has_slot) return;
	std::unique_lock<std::mutex> lk(g_slot_m);
	g_slot_cv.wait(lk, [] { return g_slot_cap <= 0 || g_slot_used < g_slot_cap; });
	g_slot_used++;
	t_has_slot = true;
}

void slot_release()
{
	if (!t_has_slot) return;
	{
		std::lock_guard<std::mutex> lk(g_slot_m);
		g_slot_used--;
	}
	t_has_slot = false;
	g_slot_cv.notify_one();
}

// ---- execution of a batch on a stream -------------------------------------------------------------------
std::mutex g_stat_m;
long long g_stat_launch[HIBAG_OP_STAT_N] = {}, g_stat_ops[HIBAG_OP_STAT_N] = {};
double g_stat_time[12] = {};                  // seconds an operation took from hand-over to results, by kind [0..7]; seconds and number of batches, by lane [8..11]
double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

// ---- the copies of a batch as ONE kernel each way --------------------------------------------------------------------
// An operation's inputs and results are a few small arrays between pinned host memory and the device.  As hipMemcpyAsync
// calls -- six per growth step and trainer -- they were what a node of trainers saturated first: 25,000 copies a second
// through the runtime's copy path, each tens of microseconds of queue time, whatever the number of streams
// (profiles/r06_notes.txt).  Pinned host memory is addressable from the device, so a batch's uploads are ONE kernel that
// reads the staging areas over PCIe and writes the device arenas, and its downloads one more: workgroup = 16 KB piece of one
// copy, 16 bytes per lane where both ends are aligned for it.  Ordered with the batch's kernels by the stream; the host's
// staging writes are visible to a kernel launched after them, the kernel's writes to host memory once its event is done.
#define COPY_MAX 48                             // copies per launch (the table travels as kernel arguments)
#define COPY_PIECE 16384
struct CopyTable {
	int n;
	int first[COPY_MAX + 1];                    // workgroups [first[i], first[i + 1]) move copy i, COPY_PIECE bytes each
	void *dst[COPY_MAX];
	const void *src[COPY_MAX];
	unsigned long long bytes[COPY_MAX];
};

__global__ __launch_bounds__(256) void k_combine_copy(CopyTable T)
{
	int i = 0;
	while (i + 1 < T.n && (int)blockIdx.x >= T.first[i + 1]) i++;
	i = __builtin_amdgcn_readfirstlane(i);
	const unsigned long long lo = (unsigned long long)((int)blockIdx.x - T.first[i]) * COPY_PIECE;
	const unsigned long long hi = lo + COPY_PIECE < T.bytes[i] ? lo + COPY_PIECE : T.bytes[i];
	char *d = (char *)T.dst[i];
	const char *s = (const char *)T.src[i];
	const unsigned long long both = (unsigned long long)(uintptr_t)d | (unsigned long long)(uintptr_t)s;
	if ((both & 15) == 0) {
		unsigned long long at = lo + 16ull * threadIdx.x;
		for (; at + 16 <= hi; at += 16ull * 256) *(uint4 *)(d + at) = *(const uint4 *)(s + at);
		// (the last piece's tail, byte by byte: at most 15 bytes)
		const unsigned long long tail = hi & ~15ull;
		if (tail >= lo && tail + threadIdx.x < hi && threadIdx.x < 16) d[tail + threadIdx.x] = s[tail + threadIdx.x];
	} else if ((both & 3) == 0) {
		unsigned long long at = lo + 4ull * threadIdx.x;
		for (; at + 4 <= hi; at += 4ull * 256) *(uint32_t *)(d + at) = *(const uint32_t *)(s + at);
		const unsigned long long tail = hi & ~3ull;
		if (tail >= lo && tail + threadIdx.x < hi && threadIdx.x < 4) d[tail + threadIdx.x] = s[tail + threadIdx.x];
	} else {
		for (unsigned long long at = lo + threadIdx.x; at < hi; at += 256) d[at] = s[at];
	}
}

// enqueue the copies `c[0 .. n)` on `st`, COPY_MAX per launch
void launch_copies(const HibagCopy *c, int n, hipStream_t st)
{
	CopyTable T;
	T.n = 0;
	int at = 0;
	auto flush = [&]() {
		if (T.n == 0) return;
		for (int i = T.n; i <= COPY_MAX; i++) T.first[i] = at;
		hipLaunchKernelGGL(k_combine_copy, dim3(at), dim3(256), 0, st, T);
		T.n = 0; at = 0;
	};
	for (int i = 0; i < n; i++) {
		if (c[i].bytes == 0) continue;
		T.first[T.n] = at;
		T.dst[T.n] = c[i].dst; T.src[T.n] = c[i].src; T.bytes[T.n] = c[i].bytes;
		at += (int)((c[i].bytes + COPY_PIECE - 1) / COPY_PIECE);
		if (++T.n == COPY_MAX) flush();
	}
	flush();
}

// The leader's wait for its batch.  hipEventSynchronize spins on the host even for an event created with hipEventBlockingSync
// (ROCm 7.2: a leader burnt a core for the length of every batch, profiles/r06_notes.txt), and the trainers' threads are
// meant to cost the host next to nothing: poll the event between short sleeps instead (timer slack of the thread lowered
// to a microsecond, so that a 25 us sleep is not rounded up to 75).
// `expect_s`: the shortest of the lane's recent batches -- the first sleep is half of that (a poll costs the host a few
// microseconds, and an EM batch lasts forty naps), then 25 us naps.  (Sleeping most of the AVERAGE batch instead was measured:
// batches differ too much, the overshoot cost 6 % of the throughput.)
hipError_t sleepy_wait(hipEvent_t ev, double expect_s)
{
	static thread_local bool slack_set = false;
	if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); slack_set = true; }
	if (expect_s > 200e-6) {
		const long ns = (long)(0.5 * std::min(expect_s, 2e-3) * 1e9);
		const timespec first{0, ns};
		nanosleep(&first, nullptr);
	}
	const timespec nap{0, 25000};
	for (;;) {
		const hipError_t q = hipEventQuery(ev);
		if (q != hipErrorNotReady) return q;
		nanosleep(&nap, nullptr);
	}
}

// Everything of the batch is enqueued, then waited for once.  `done`: the event the leader of shared trainers sleeps on
// (sleepy_wait; it gives up its host-thread slot meanwhile); nullptr: a trainer that runs alone waits on its own stream.
hipError_t execute(HibagOp *const ops[], int n, hipStream_t st, hipEvent_t done, double expect_s = 0)
{
	const bool give_up_slot = done != nullptr;
	hipError_t err = hipSuccess;
	auto ok = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; return e == hipSuccess; };
	std::vector<HibagCopy> cp;
	for (int i = 0; i < n; i++) cp.insert(cp.end(), ops[i]->up.begin(), ops[i]->up.end());
	launch_copies(cp.data(), (int)cp.size(), st);
	ok(hipGetLastError());
	long long nl[HIBAG_OP_KINDS] = {}, no[HIBAG_OP_KINDS] = {};
	for (int kind = 0; kind < HIBAG_OP_KINDS && err == hipSuccess; kind++) {
		const HibagOp *sel[HIBAG_COMBINE_MAX];
		int m = 0;
		for (int i = 0; i <= n; i++) {
			if (i < n && ops[i]->kind == kind) sel[m++] = ops[i];
			if (m == HIBAG_COMBINE_MAX || (i == n && m > 0)) {
				if (!g_launch[kind]) { err = hipErrorInvalidValue; break; }
				g_launch[kind](sel, m, st);
				nl[kind]++; no[kind] += m;
				m = 0;
			}
		}
		ok(hipGetLastError());
	}
	if (err == hipSuccess) {
		cp.clear();
		for (int i = 0; i < n; i++) cp.insert(cp.end(), ops[i]->down.begin(), ops[i]->down.end());
		launch_copies(cp.data(), (int)cp.size(), st);
		ok(hipGetLastError());
	}
	if (give_up_slot) slot_release();
	// (also after a failed enqueue: nothing of the batch may still be in flight)
	if (done && err == hipSuccess && ok(hipEventRecord(done, st))) ok(sleepy_wait(done, expect_s));
	else ok(hipStreamSynchronize(st));
	if (give_up_slot) slot_acquire();
	{
		std::lock_guard<std::mutex> lk(g_stat_m);
		for (int k = 0; k < HIBAG_OP_KINDS; k++) { g_stat_launch[k] += nl[k]; g_stat_ops[k] += no[k]; }
	}
	return err;
}

// ---- the lanes ----------------------------------------------------------------------------------------------
constexpr int LANE_SLOTS = 6;                     // at most (the arrays' size); in use: lane_slots() below
// batches of a lane in flight at a time (a stream and an event each): three (measured: profiles/r06_notes.txt item 4);
// HIBAG_COMBINE_SLOTS_SHORT / _EM override for measurements
int lane_slots(int lane)
{
	static const int n[2] = {
		getenv("HIBAG_COMBINE_SLOTS_SHORT") ? std::max(1, std::min(LANE_SLOTS, atoi(getenv("HIBAG_COMBINE_SLOTS_SHORT")))) : 3,
		getenv("HIBAG_COMBINE_SLOTS_EM") ? std::max(1, std::min(LANE_SLOTS, atoi(getenv("HIBAG_COMBINE_SLOTS_EM")))) : 3};
	return n[lane];
}
struct Lane {
	std::mutex m;
	std::vector<HibagOp *> pending;
	bool busy[LANE_SLOTS] = {};
	double recent[8] = {};                        // how long the lane's last batches took (seconds; 0 = none yet)
	int recent_at = 0;
	hipStream_t st[LANE_SLOTS] = {};
	hipEvent_t ev[LANE_SLOTS] = {};
};
constexpr int MAX_DEV = 64;
Lane g_lane[MAX_DEV][2];                          // [device][0 = short operations, 1 = EM fits]

} // namespace

void hibag_combine_register(int kind, HibagOpLaunch fn) { if (kind >= 0 && kind < HIBAG_OP_KINDS) g_launch[kind] = fn; }
void hibag_combine_set_shared(bool on) { t_shared = on; }
bool hibag_combine_shared() { return t_shared; }

void hibag_combine_set_budget(int n)
{
	{
		std::lock_guard<std::mutex> lk(g_slot_m);
		g_slot_cap = n > 0 ? n : 0;
	}
	g_slot_cv.notify_all();
}

void hibag_combine_enter() { if (t_shared) slot_acquire(); }
void hibag_combine_leave() { slot_release(); }

void hibag_combine_stats(long long launches[HIBAG_OP_STAT_N], long long ops[HIBAG_OP_STAT_N], int reset)
{
	std::lock_guard<std::mutex> lk(g_stat_m);
	for (int k = 0; k < HIBAG_OP_STAT_N; k++) {
		if (launches) launches[k] = g_stat_launch[k];
		if (ops) ops[k] = g_stat_ops[k];
		if (reset) g_stat_launch[k] = g_stat_ops[k] = 0;
	}
}

void hibag_combine_times(double out[12], int reset)
{
	std::lock_guard<std::mutex> lk(g_stat_m);
	for (int k = 0; k < 12; k++) { if (out) out[k] = g_stat_time[k]; if (reset) g_stat_time[k] = 0; }
}

void hibag_combine_run(HibagOp &op)
{
	op.done = false; op.err = hipSuccess;
	if (!t_shared) {
		HibagOp *one[1] = {&op};
		const hipError_t e = execute(one, 1, 0, nullptr);        // stream 0 = this thread's own (per-thread default stream)
		if (e != hipSuccess) combine_throw("device operation", e);
		return;
	}
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess || dev < 0 || dev >= MAX_DEV) combine_throw("hipGetDevice", e != hipSuccess ? e : hipErrorInvalidDevice);
	const int lane = op.kind == HIBAG_OP_EM ? 1 : 0;
	Lane &L = g_lane[dev][lane];
	const double t_in = now_s();
	std::unique_lock<std::mutex> lk(L.m);
	L.pending.push_back(&op);
	while (!op.done) {
		int slot = -1;
		for (int i = 0; i < lane_slots(lane); i++) if (!L.busy[i]) { slot = i; break; }
		// (this operation may already be part of a batch another leader took: then it is no longer pending and all there is to do is wait)
		bool mine_pending = false;
		for (HibagOp *o : L.pending) if (o == &op) { mine_pending = true; break; }
		if (slot < 0 || !mine_pending) {
			slot_release();
			op.cv.wait(lk);
			lk.unlock();                                         // (never hold the lane while waiting for a host-thread slot)
			slot_acquire();
			lk.lock();
			continue;
		}
		// lead: everything that is pending -- this thread's operation among it -- is one batch on the free slot's stream
		std::vector<HibagOp *> batch;
		batch.swap(L.pending);
		L.busy[slot] = true;
		if (!L.st[slot]) {
			// (the short operations' streams at the highest priority: streams of a priority class have hardware queues of their own
			// -- an EM fit at the head of a queue never holds a pair list up -- and a short kernel goes ahead of a long one)
			int pr_lo = 0, pr_hi = 0;
			(void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
			if (hipStreamCreateWithPriority(&L.st[slot], hipStreamNonBlocking, lane == 0 ? pr_hi : pr_lo) != hipSuccess) L.st[slot] = nullptr;
			if (L.st[slot] && hipEventCreateWithFlags(&L.ev[slot], hipEventDisableTiming) != hipSuccess) L.ev[slot] = nullptr;
		}
		const hipStream_t st = L.st[slot];
		const hipEvent_t ev = L.ev[slot];
		double expect = 1e9;
		for (double r : L.recent) expect = std::min(expect, r > 0 ? r : 0.0);      // (the shortest recent batch; 0 until eight are known)
		const double t_b = now_s();
		lk.unlock();
		// (nothing may leave this block by exception: the batch's owners sleep until they are told, and the slot stays taken)
		hipError_t be = hipErrorOutOfMemory;
		try {
			if (st && ev) be = execute(batch.data(), (int)batch.size(), st, ev, expect);
		} catch (...) {
			if (st) (void)hipStreamSynchronize(st);
			slot_acquire();
		}
		const double took = now_s() - t_b;
		{
			std::lock_guard<std::mutex> sl(g_stat_m);
			g_stat_time[8 + lane] += took; g_stat_time[10 + lane] += 1;
		}
		lk.lock();
		L.recent[L.recent_at++ & 7] = took;
		// wake the batch's owners -- each on its own condition variable: nobody is woken for nothing -- and, if operations piled
		// up meanwhile, ONE of their owners to lead them (whoever leads takes everything that is pending)
		for (HibagOp *o : batch) { o->err = be; o->done = true; if (o != &op) o->cv.notify_one(); }
		L.busy[slot] = false;
		if (!L.pending.empty()) L.pending.front()->cv.notify_one();
	}
	lk.unlock();
	{
		std::lock_guard<std::mutex> sl(g_stat_m);
		g_stat_time[op.kind] += now_s() - t_in;
	}
	if (op.err != hipSuccess) combine_throw("device operation (combined launch)", op.err);
}
