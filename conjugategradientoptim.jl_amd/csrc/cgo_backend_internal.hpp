// cgo_backend_internal.hpp — what the translation units of the device backend share (round 4: cgo_hip_backend.hip split along its
// families): cgo_hip_backend.hip (context, reductions to the host, the stored-gradient k_fused launches, results, raw entry points),
// cgo_backend_cg.hip (k_cg / k_chain launches, reduction tails, on-device controller, resident solver), cgo_backend_lbfgs.hip
// (log-sum-exp + L-BFGS ring) and cgo_backend_place.hip (buffer placement search, stream-mix harness).
#pragma once

#include "cgo_hip_backend.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

namespace cgo {

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            set_error(std::string("HIP error: ") + hipGetErrorString(e__) + " at " #expr); \
            return CGO_EHIP;                                                               \
        }                                                                                  \
    } while (0)

static inline double now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

// streaming policy and grids (cgo_hip_backend.hip)
double big_bytes_for(double forced, bool read_only = false);
double env_big_bytes();
bool is_big(int obj_kind, int mode, int64_t n, bool hp, double forced);
int grid_cg(int64_t n, int npts = 1);
// ALGORITHMIC bytes of a k_cg / k_chain launch (cgo_backend_cg.hip)
double bytes_r(int obj_kind, int mode, int64_t n, bool has_param);
void unpack_r(const double *s, int k, Scal *out, bool dir);
// host side of the publish protocols (cgo_hip_backend.hip)
int launch_module(hipFunction_t f, void *params, int grid, hipStream_t st);
int wait_word(HipCtx *ctx, unsigned long long *word, unsigned long long want);
int wait_checked(HipCtx *ctx, unsigned long long *word, unsigned long long want, const double *block, int ns, double *dst);

}  // namespace cgo
