// cgo_rtc.hpp — run-time compilation of user-supplied element-wise objectives (hiprtc, gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>

namespace cgo {

// A loaded code object holding k_cg<UserObjective, …> and k_fused<UserObjective, …> instantiations.
struct RtcModule {
    hipModule_t mod = nullptr;
    std::map<std::string, hipFunction_t> fn;
    ~RtcModule();
    hipFunction_t cg(int mode, int npts, bool big) const;
    hipFunction_t fused(int mode, bool big) const;
    hipFunction_t resident(int npts) const;   // k_resident<UserObjective, npts> (npts = 3 only), or nullptr
    hipFunction_t spec(bool big, bool push) const;   // k_lbfgs_combine_spec<UserObjective, big, push>
    hipFunction_t lite(bool big) const;              // k_lbfgs_push_lite<UserObjective, big>
};

// `source`: either a complete `struct UserObjective { … };` (functor interface of
// cgo_kernels.hip.hpp) or just the statements of an element-wise body that compute `fi` and
// `gi` from `x`, `p`, `s0`.  Returns CGO_OK / CGO_EINVAL (compile log in `log`) / CGO_EHIP.
int rtc_compile_objective(int device, const std::string &source, bool has_param,
                          std::shared_ptr<RtcModule> &out, std::string &log);

}  // namespace cgo
