// hibag_pool.h -- the trainer's persistent host threads (hibag_train.hip): the candidate SNPs of a growth step are fitted
// concurrently (CAlg_EM, src/LibHLA.cpp:1127-1255, once per candidate), and a step lasts about a millisecond, so creating
// threads per step would cost as much as the work.  Header-only so that tests/native/pool_test.cpp can run it under
// ThreadSanitizer on a CPU-only box (tools/run_sanitizers.sh).
#ifndef HIBAG_POOL_H_
#define HIBAG_POOL_H_

#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Persistent helper threads for the per-step fits: a growth step lasts about a millisecond, so
// creating threads per step would cost as much as the work.
class Pool {
	std::vector<std::thread> th;
	std::mutex m;
	std::condition_variable wake, done;
	std::function<void()> job;
	unsigned long gen = 0;
	int pending = 0;
	bool stop = false;
	std::exception_ptr failed;                  // first exception of a job, rethrown by run() on the caller
	void note_failure()
	{
		std::lock_guard<std::mutex> lk(m);
		if (!failed) failed = std::current_exception();
	}
public:
	explicit Pool(int n_helpers)
	{
		for (int i = 0; i < n_helpers; i++)
			th.emplace_back([this] {
				unsigned long seen = 0;
				for (;;) {
					std::function<void()> f;
					{
						std::unique_lock<std::mutex> lk(m);
						wake.wait(lk, [&] { return stop || gen != seen; });
						if (stop) return;
						seen = gen;
						f = job;
					}
					try { f(); } catch (...) { note_failure(); }     // never let an exception leave the thread
					{
						std::lock_guard<std::mutex> lk(m);
						if (--pending == 0) done.notify_one();
					}
				}
			});
	}
	~Pool()
	{
		{ std::lock_guard<std::mutex> lk(m); stop = true; }
		wake.notify_all();
		for (std::thread &t : th) t.join();
	}
	// runs f on every helper and on the caller; returns when all are done
	void run(const std::function<void()> &f)
	{
		{
			std::lock_guard<std::mutex> lk(m);
			job = f; gen++; pending = (int)th.size();
		}
		wake.notify_all();
		try { f(); } catch (...) { note_failure(); }
		// always wait for the helpers: they use objects on the caller's stack
		std::unique_lock<std::mutex> lk(m);
		done.wait(lk, [&] { return pending == 0; });
		if (failed) {
			std::exception_ptr e = failed;
			failed = nullptr;
			lk.unlock();
			std::rethrow_exception(e);
		}
	}
};

#endif
