"""The HOST side of the C ABI under sanitizers, without a GPU (round-5 verdict, item 4).

dsdtm_amd/csrc/api.cpp — stream rings and pair counters, recover slots, the team ring's tag epoch, the sharded / streamed
entries, frames and their buffer pool, the staging blocks every single-call entry and dsdtm_track_frame pack — is compiled with
g++ against a FAKE HIP runtime (tests/fake_hip/: "device" memory in the host heap; every asynchronous operation queued and run
only when something waits for it; fake kernel launchers that touch every buffer a launch names, check the invariants the real
kernels rely on and raise their timeout word on request; any HIP call can be told to fail) and driven through the scenarios of
tests/fake_hip/driver.cpp under AddressSanitizer + UndefinedBehaviorSanitizer (+ LeakSanitizer), and under ThreadSanitizer for
two contexts on two threads. Each of the five lifetime / staleness defects the round-4 advisor found by reading fails one of
these scenarios when its fix is reverted (shown when the job was added: a_no_forget -> heap-use-after-free in
sharded_error_midway; re-run on the context's stream -> recover_slots_past_64; no sticky flag -> evicted_timeout_is_reported;
no second download -> sharded_refetches_after_rerun; no re-zeroing at the epoch wrap -> team_epoch_wrap_hazard_is_real)."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_hip")
SCENARIOS = ["ring_table_past_16_streams", "evicted_timeout_is_reported", "recover_slots_past_64", "team_epoch_wrap_is_cleared",
             "team_epoch_wrap_hazard_is_real", "sharded_refetches_after_rerun", "sharded_error_midway", "streamed_entry", "frame_lifetime",
             "single_call_entries", "track_frame_packing", "track_frame_failures", "two_contexts_two_threads"]


@pytest.fixture(scope="module")
def drivers():
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)
    return os.path.join(HERE, "_build", "driver_asan"), os.path.join(HERE, "_build", "driver_tsan")


def _run(exe, env, scenarios=()):
    r = subprocess.run([exe, *scenarios], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok ", "FAILED "))]
    return r, lines


def test_host_bookkeeping_under_address_and_ub_sanitizers(drivers):
    r, lines = _run(drivers[0], {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert lines == ["ok " + s for s in SCENARIOS], lines
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr


def test_two_contexts_on_two_threads_under_thread_sanitizer(drivers):
    r, lines = _run(drivers[1], {"TSAN_OPTIONS": "halt_on_error=1"},
                    ["two_contexts_two_threads", "ring_table_past_16_streams", "recover_slots_past_64", "sharded_refetches_after_rerun", "streamed_entry"])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert len(lines) == 5 and all(l.startswith("ok ") for l in lines), lines
    assert "ThreadSanitizer" not in r.stderr
