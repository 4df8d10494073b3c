"""The threaded CPU-side code under load (what tools/run_sanitizers.sh runs under ASan / UBSan / TSan; without a sanitizer
these are plain functional tests): the AVX2 port predicts with one thread per slice of samples and must equal the scalar
oracle whatever the thread count; the trainer's host thread pool (hibag_amd/csrc/hibag_pool.h) runs thousands of short
rounds, rethrows a job's exception on the caller and can be destroyed idle."""

import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_threaded_avx2_port_equals_the_scalar_oracle_at_every_thread_count(oracle):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small")
    G, _ = synth.make_samples(founders, af, 257)
    fm = oracle.flatten(model)
    want = oracle.predict(fm, G, vote_method=1)
    for threads in (1, 2, 3, 8, 16):
        for vote in (1, 2):
            ref = want if vote == 1 else oracle.predict(fm, G, vote_method=2)
            got = oracle.predict(fm, G, vote_method=vote, avx2=True, n_threads=threads)
            for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
                assert np.array_equal(got[k], ref[k], equal_nan=True), (threads, vote, k)


def test_trainer_thread_pool_harness(tmp_path):
    exe = str(tmp_path / "pool_test")
    san = os.environ.get("HIBAG_POOL_TEST_SANITIZER", "")           # e.g. "thread" or "address,undefined"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-I", os.path.join(ROOT, "hibag_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "pool_test.cpp"), "-o", exe] + ([f"-fsanitize={san}"] if san else [])
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}   # (a preloaded sanitizer runtime is for the Python process)
    subprocess.check_call(cmd, env=env)
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "pool_test OK" in p.stdout


def test_oracle_threads_harness(tmp_path):
    """tests/native/oracle_threads_test.c (what tools/run_sanitizers.sh builds with -fsanitize=thread): the threaded AVX2 port
    equals the scalar oracle at every thread count."""
    exe = str(tmp_path / "oracle_threads_test")
    san = os.environ.get("HIBAG_POOL_TEST_SANITIZER", "")
    src = [os.path.join(ROOT, "tests", "native", "oracle_threads_test.c"), os.path.join(ROOT, "oracle", "hibag_oracle.c"),
           os.path.join(ROOT, "oracle", "hibag_oracle_avx2.c")]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    subprocess.check_call(["gcc", "-O1", "-g", "-ffp-contract=off", "-pthread"] + ([f"-fsanitize={san}"] if san else []) + src + ["-o", exe, "-lm"], env=env)
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "oracle_threads_test OK" in p.stdout or "skipped" in p.stdout
