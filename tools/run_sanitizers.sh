#!/bin/bash
# CPU-side sanitizer runs (GPU sanitizers are not available on the pool): the oracle -- the scalar restatement, the threaded
# AVX2 port and the training driver -- under AddressSanitizer + UBSan and under ThreadSanitizer, driven by the CPU
# test-suite (every test that touches the oracle), plus the trainer's host thread pool (hibag_amd/csrc/hibag_pool.h) in a
# stand-alone TSan harness.  Logs: profiles/rNN_sanitizers_*.txt.   usage: tools/run_sanitizers.sh r04
set -u
cd "$(dirname "$0")/.."
tag=${1:-r04}
which=${2:-both}
make -C oracle -s SAN=asan && make -C oracle -s SAN=tsan || exit 1
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
echo "== TSan: $TESTS" > profiles/${tag}_sanitizers_tsan.txt
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 \
  HIBAG_ORACLE_LIBRARY=$PWD/oracle/libhibag_oracle_tsan.so timeout 3000 python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider >> profiles/${tag}_sanitizers_tsan.txt 2>&1
echo "exit code $?" >> profiles/${tag}_sanitizers_tsan.txt
grep -c "WARNING: ThreadSanitizer" profiles/${tag}_sanitizers_tsan.txt | sed 's/^/sanitizer reports: /' >> profiles/${tag}_sanitizers_tsan.txt
echo "== TSan: host thread pool of the trainer (tests/native/pool_test.cpp)" >> profiles/${tag}_sanitizers_tsan.txt
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -I hibag_amd/csrc tests/native/pool_test.cpp -o /tmp/pool_test_tsan && /tmp/pool_test_tsan >> profiles/${tag}_sanitizers_tsan.txt 2>&1
echo "exit code $?" >> profiles/${tag}_sanitizers_tsan.txt
tail -3 profiles/${tag}_sanitizers_asan.txt profiles/${tag}_sanitizers_tsan.txt
