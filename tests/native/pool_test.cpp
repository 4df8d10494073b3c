// Stand-alone harness for the trainer's host thread pool (hibag_amd/csrc/hibag_pool.h), built with -fsanitize=thread (and
// address,undefined) by tools/run_sanitizers.sh and by tests/test_sanitized_threads.py: the use the trainer makes of it
// -- every helper and the caller pull candidate indices from one atomic counter and write their own slots of a result
// array, thousands of short rounds -- plus jobs that throw, and pools created and destroyed while idle or right after work.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <vector>
#include "hibag_pool.h"

int main()
{
	long long checksum = 0;
	for (int helpers : {1, 3, 15}) {
		Pool pool(helpers);
		for (int round = 0; round < 2000; round++) {
			const int n = 18;                         // candidates of a growth step (mtry)
			std::vector<double> fit(n, 0.0);
			std::atomic<int> next{0};
			pool.run([&] {
				for (int i; (i = next.fetch_add(1)) < n;) {
					double v = 0;
					for (int k = 0; k < 50; k++) v += (i + 1) * 1e-3 * k;      // (a fit's worth of arithmetic, scaled down)
					fit[i] = v;
				}
			});
			for (int i = 0; i < n; i++) checksum += (long long)(fit[i] * 1000);
		}
		// a job that throws on one thread: run() must still wait for everybody and rethrow on the caller
		int caught = 0;
		for (int round = 0; round < 50; round++) {
			std::atomic<int> next{0};
			try {
				pool.run([&] { if (next.fetch_add(1) == 0) throw std::runtime_error("fit failed"); });
			} catch (const std::runtime_error &) { caught++; }
		}
		if (caught != 50) { std::printf("pool_test: %d of 50 exceptions reached the caller\n", caught); return 1; }
	}
	for (int i = 0; i < 200; i++) { Pool idle(4); }        // created and destroyed without work
	std::printf("pool_test OK (checksum %lld)\n", checksum);
	return 0;
}
