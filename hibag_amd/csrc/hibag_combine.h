// hibag_combine.h -- one launch per kind of device work for ALL trainers of a process that run side by side.
//
// A growth step of hlaAttrBagging's search (CVariableSelection::Search, src/LibHLA.cpp:1981-2122) makes four device round
// trips -- the candidate pair lists in two passes (_PrepHaploMatch, :1569-1637), the EM fits of the step's candidates
// (CAlg_EM, :1127-1255), their scoring (_BestGuess / _PostProb, :1639-1767) -- each a few small copies and one or two short
// kernels.  With several trainers of one process on one device (hibag_amd.train.grow_concurrently: the decomposition of
// hlaParallelAttrBagging's workers, R/HIBAG.R:329-390) every trainer used to issue those on a stream of its own: sixteen
// streams share the runtime's four hardware queues, a 0.6 ms EM kernel at the head of a queue holds up every short operation
// mapped behind it, and a growth step took 4 ms instead of 1.2 (profiles/r06_notes.txt: 8 trainers, 2 kernels in flight on
// average).  Here a trainer instead hands each operation -- its copies, the argument block ("view") of its kernels -- to a
// COMBINER: whichever trainer thread finds the combiner idle becomes its leader, takes every operation that is pending, sends
// all their uploads, launches ONE fused kernel per kind (grid = the operations' workgroups back to back, each workgroup
// looks up whose it is), queues all downloads, waits once and wakes the owners.  While it waits the next operations pile up
// for the next leader.  Two combiners per device -- the long EM fits on one, the short operations on the other, so a pair
// list never queues behind an EM fit (the short operations' streams at the highest priority: hardware queues of their own) --
// with three streams each: another leader may start its batch while the first one's runs.  A batch's copies are ONE kernel
// each way (pinned host memory is device-addressable; per-operation hipMemcpyAsync calls were what saturated first), and the
// leader SLEEPS between event queries while the device works (hipEventSynchronize spins, also for a blocking-sync event: the
// trainers' threads are meant to cost the host next to nothing).  Results are bit-identical: the kernels' per-workgroup work
// is unchanged.  Measurements of every step of that: profiles/r06_notes.txt item 4.
//
// A trainer that runs alone (the default, and the plugin-table entries an unmodified HIBAG drives) uses the same code with a
// batch of one on its own thread's stream: hibag_combine_run() is the only way this library launches training work.
#ifndef HIBAG_COMBINE_H_
#define HIBAG_COMBINE_H_

#include <hip/hip_runtime.h>
#include <condition_variable>
#include <vector>

#define HIBAG_COMBINE_MAX 16                 // operations fused into one launch (their views travel as kernel arguments)

// MATCH0 / MATCH1: the two passes of the candidate pair lists as operations of their own (the host sizes the second from the
// first's counts); MATCH: both passes in one operation (the second finds its offsets on the device, the read-back is sized by
// an upper bound) -- what the driver uses whenever the bound is small; EVAL: the scoring of a step's candidates; EM: their fits.
enum { HIBAG_OP_MATCH0 = 0, HIBAG_OP_MATCH1, HIBAG_OP_EVAL, HIBAG_OP_EM, HIBAG_OP_MATCH, HIBAG_OP_KINDS };
#define HIBAG_OP_STAT_N 8                    // length of the statistics arrays below (>= HIBAG_OP_KINDS)

struct HibagCopy { void *dst; const void *src; size_t bytes; };

struct HibagOp {
	int kind = 0;
	const void *view = nullptr;              // the kind's argument block (MatchView, BatchView, EmView)
	std::vector<HibagCopy> up, down;         // pinned host -> device before the kernels, device -> pinned host behind them
	// the combiner's
	bool done = false;
	hipError_t err = hipSuccess;
	std::condition_variable cv;              // the owner sleeps here: woken when its batch is done, or to lead the next one
};

// The fused launch of n <= HIBAG_COMBINE_MAX operations of one kind on `st` (registered by the file that owns the kernels).
typedef void (*HibagOpLaunch)(const HibagOp *const ops[], int n, hipStream_t st);
void hibag_combine_register(int kind, HibagOpLaunch fn);

// Runs the operation -- uploads, kernels, downloads -- and returns when its results are in host memory.  On the calling
// thread's own stream, or through the device's combiners when the thread is in shared mode.  Throws `const char *`.
void hibag_combine_run(HibagOp &op);

// Shared mode of the calling thread (a trainer that runs beside others: hibag_hip_trainer_set_shared).
void hibag_combine_set_shared(bool on);
bool hibag_combine_shared();

// Host-thread budget of the shared trainers: at most `n` of their threads are runnable at a time (0 = no limit).  A thread
// gives its slot up while it waits for the device or for a leader and takes one again before it goes on.
void hibag_combine_set_budget(int n);
void hibag_combine_enter();                  // a shared trainer's thread starts / ends its work
void hibag_combine_leave();

// statistics since the last reset: fused launches, operations in them (launches x mean batch size), by kind
void hibag_combine_stats(long long launches[HIBAG_OP_STAT_N], long long ops[HIBAG_OP_STAT_N], int reset);
// shared mode only: seconds the operations took from hand-over to results, summed by kind [0 .. 7]; seconds spent in batches
// and number of batches, by lane (short operations, EM fits) [8..9], [10..11]
void hibag_combine_times(double out[12], int reset);

// ---- device side: the views of a fused launch and the owner of a workgroup ----
template <class V>
struct HibagMulti {
	int n;
	int first[HIBAG_COMBINE_MAX + 1];        // workgroups [first[j], first[j + 1]) belong to operation j
	V v[HIBAG_COMBINE_MAX];
};

#ifdef __HIPCC__
template <class V>
__device__ __forceinline__ int hibag_multi_owner(const HibagMulti<V> &M, int block)
{
	int j = 0;
	while (j + 1 < M.n && block >= M.first[j + 1]) j++;
	return __builtin_amdgcn_readfirstlane(j);
}
#endif

#endif
