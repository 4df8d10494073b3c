#!/bin/bash
# CPU-side sanitizer runs (GPU sanitizers are not available on the pool): the oracle -- the scalar restatement, the threaded
# AVX2 port and the training driver -- under AddressSanitizer + UBSan, driven by the CPU test-suite (every test that touches
# the oracle); the threaded code -- the AVX2 port's sample threads, the trainer's host thread pool
# (hibag_amd/csrc/hibag_pool.h) -- under ThreadSanitizer in stand-alone harnesses (tests/native/).  Logs: profiles/rNN_sanitizers_*.txt.   usage: tools/run_sanitizers.sh r04
set -u
cd "$(dirname "$0")/.."
tag=${1:-r04}
which=${2:-both}
make -C oracle -s SAN=asan || exit 1
TESTS="tests/test_oracle.py tests/test_oracle_pin.py tests/test_oracle_train.py tests/test_bed_host.py tests/test_printed_example.py tests/test_sanitized_threads.py"
if [ "$which" != tsan ]; then
echo "== ASan + UBSan: $TESTS" > profiles/${tag}_sanitizers_asan.txt
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  HIBAG_ORACLE_LIBRARY=$PWD/oracle/libhibag_oracle_asan.so timeout 3000 python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider >> profiles/${tag}_sanitizers_asan.txt 2>&1
echo "exit code $?" >> profiles/${tag}_sanitizers_asan.txt
grep -c "ERROR: AddressSanitizer\|runtime error:" profiles/${tag}_sanitizers_asan.txt | sed 's/^/sanitizer reports: /' >> profiles/${tag}_sanitizers_asan.txt
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -pthread -I hibag_amd/csrc tests/native/pool_test.cpp -o /tmp/pool_test_asan && /tmp/pool_test_asan >> profiles/${tag}_sanitizers_asan.txt 2>&1
echo "pool_test exit code $?" >> profiles/${tag}_sanitizers_asan.txt
fi
# ThreadSanitizer: stand-alone harnesses (the Python interpreter is not instrumented and hangs under the TSan runtime):
# the oracle's threaded AVX2 port against the scalar oracle at 1..16 threads, and the trainer's host thread pool
OSRC="oracle/hibag_oracle.c oracle/hibag_oracle_avx2.c"
echo "== TSan: the oracle's threaded AVX2 port (tests/native/oracle_threads_test.c + $OSRC)" > profiles/${tag}_sanitizers_tsan.txt
gcc -O1 -g -fsanitize=thread -ffp-contract=off -pthread tests/native/oracle_threads_test.c $OSRC -o /tmp/oracle_threads_tsan -lm \
  && TSAN_OPTIONS=halt_on_error=0 timeout 900 /tmp/oracle_threads_tsan >> profiles/${tag}_sanitizers_tsan.txt 2>&1
echo "exit code $?" >> profiles/${tag}_sanitizers_tsan.txt
echo "== TSan: host thread pool of the trainer (tests/native/pool_test.cpp)" >> profiles/${tag}_sanitizers_tsan.txt
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -I hibag_amd/csrc tests/native/pool_test.cpp -o /tmp/pool_test_tsan && timeout 900 /tmp/pool_test_tsan >> profiles/${tag}_sanitizers_tsan.txt 2>&1
echo "exit code $?" >> profiles/${tag}_sanitizers_tsan.txt
grep -c "WARNING: ThreadSanitizer" profiles/${tag}_sanitizers_tsan.txt | sed 's/^/sanitizer reports: /' >> profiles/${tag}_sanitizers_tsan.txt
if [ "$which" != tsan ]; then
echo "== ASan + UBSan: the same harness" >> profiles/${tag}_sanitizers_asan.txt
gcc -O1 -g -fsanitize=address,undefined -ffp-contract=off -pthread tests/native/oracle_threads_test.c $OSRC -o /tmp/oracle_threads_asan -lm \
  && timeout 900 /tmp/oracle_threads_asan >> profiles/${tag}_sanitizers_asan.txt 2>&1
echo "oracle_threads_test exit code $?" >> profiles/${tag}_sanitizers_asan.txt
fi
for f in asan tsan; do tail -n 4 profiles/${tag}_sanitizers_$f.txt; done
