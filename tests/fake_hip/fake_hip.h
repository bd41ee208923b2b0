// tests/fake_hip/fake_hip.h — control interface of the fake HIP runtime (test infrastructure only).
#pragma once
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

enum { FAKE_SA_ONE_CU = 0, FAKE_SA_DUO = 1, FAKE_SA_TEAM = 2, FAKE_OTHER = 3 };
struct fake_launch {
    int kind = FAKE_OTHER;
    hipStream_t stream = nullptr;       // the stream the launch was enqueued on
    double* T_cur_w = nullptr;
    long seq = 0;                        // order of EXECUTION
    int team_k = 0, n_pairs = 0, max_features = 0;
    unsigned epoch = 0;
    bool timed_out = false;
    std::string name;
};
void fake_hip_reset();                                        // drains everything, clears counters / log / errors / injections
void fake_hip_fail(const char* api, long nth_call_from_now);  // the n-th call of `api` from now returns hipErrorUnknown (once)
void fake_hip_timeout_next(long multi_cu, long one_cu);       // the next launches of that kind raise their timeout word when they run
void fake_hip_drain_all();
long fake_hip_calls(const char* api);
std::vector<fake_launch> fake_hip_log();
std::vector<std::string> fake_hip_errors();                  // invariant violations seen by the fake kernels
size_t fake_hip_live_allocations();
size_t fake_hip_pending();
