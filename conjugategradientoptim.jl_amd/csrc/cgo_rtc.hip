// cgo_rtc.hip — "user-supplied element-wise f/∇f" on the GPU path.
//
// The reference takes an arbitrary Julia closure `f = fdf!(g, x)` (src/engine/optim.jl:25,
// src/cg_utils.jl:19).  A host closure cannot run inside a HIP kernel, so the device-side
// equivalent is SOURCE: the user hands over the element-wise body (or a full functor struct),
// and the same kernel templates that serve the built-in objectives (cgo_kernels.hip.hpp,
// cgo_kernels_cg.hip.hpp — embedded in this library as text) are instantiated for it at run time
// with hiprtc for gfx950, with the same flags as the ahead-of-time build (-ffp-contract=off).
#include "cgo_rtc.hpp"

#include <hip/hiprtc.h>

#include <vector>

#include "cgo_hip_backend.hpp"
#include "cgo_rtc_sources.inc"

namespace cgo {

RtcModule::~RtcModule() {
    if (mod) (void)hipModuleUnload(mod);
}

static std::string key_cg(int mode, int npts, bool big) {
    return "cg:" + std::to_string(mode) + ":" + std::to_string(npts) + ":" + (big ? "1" : "0");
}
static std::string key_fused(int mode, bool big) {
    return "fused:" + std::to_string(mode) + ":" + (big ? "1" : "0");
}
hipFunction_t RtcModule::cg(int mode, int npts, bool big) const {
    auto it = fn.find(key_cg(mode, npts, big));
    return it == fn.end() ? nullptr : it->second;
}
hipFunction_t RtcModule::resident(int npts) const {
    auto it = fn.find("res:" + std::to_string(npts));
    return it == fn.end() ? nullptr : it->second;
}
hipFunction_t RtcModule::spec(bool big, bool push) const {
    auto it = fn.find(std::string("spec:") + (big ? "1" : "0") + (push ? "1" : "0"));
    return it == fn.end() ? nullptr : it->second;
}
hipFunction_t RtcModule::lite(bool big) const {
    auto it = fn.find(std::string("lite:") + (big ? "1" : "0"));
    return it == fn.end() ? nullptr : it->second;
}
hipFunction_t RtcModule::fused(int mode, bool big) const {
    auto it = fn.find(key_fused(mode, big));
    return it == fn.end() ? nullptr : it->second;
}

int rtc_compile_objective(int device, const std::string &source, bool has_param,
                          std::shared_ptr<RtcModule> &out, std::string &log) {
    if (hipSetDevice(device) != hipSuccess) { log = "hipSetDevice failed"; return CGO_EHIP; }
    std::string src;
    for (const char *c : kRtcKernelSourceChunks) src += c;
    src += "\nnamespace cgo { namespace dev {\n";
    if (source.find("struct UserObjective") != std::string::npos) {
        src += source;
    } else {
        // element-wise body: statements computing `fi` (objective term) and `gi` (its derivative)
        // from `x` (the element), `p` (its parameter, 0 if none) and `s0` (a scalar)
        src += "struct UserObjective {\n";
        src += std::string("    static constexpr bool kParam = ") + (has_param ? "true" : "false") + ";\n";
        src += "    static constexpr bool kPairOnly = false;\n";
        src += "    __device__ static inline void eval1(double x, double p, double s0, double &f, double &g) {\n";
        src += "        double fi = 0.0, gi = 0.0;\n        {\n" + source + "\n        }\n        f += fi; g = gi;\n    }\n";
        src += "    __device__ static inline void eval2(d2 xx, d2 pp, double s0, double &f, d2 &gg) {\n";
        src += "        double g0, g1;\n        eval1(xx.x, pp.x, s0, f, g0);\n        eval1(xx.y, pp.y, s0, f, g1);\n";
        src += "        gg.x = g0; gg.y = g1;\n    }\n};\n";
    }
    src += "\n}}\n";

    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "cgo_user_objective.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return CGO_EHIP;
    }
    struct Want { std::string key, expr; };
    std::vector<Want> wants;
    const int cg_modes[] = {8, 4, 7, 3, 1, 16, 32, 64, 128, 2, 6, 256};  // RMode combinations the backend launches
    for (int big = 0; big < 2; ++big) {
        for (int m : cg_modes) {
            const int np_max = (m == 4 || m == 7 || m == 6) ? 4 : 1;
            for (int q = 0; q < np_max; ++q) {
                const int npts = 1 + 2 * q;
                wants.push_back({key_cg(m, npts, big),
                                 "cgo::dev::k_cg<cgo::dev::UserObjective, " + std::to_string(m) + ", " +
                                     std::to_string(npts) + ", " + (big ? "true" : "false") + ">"});
            }
        }
        for (int m : {16, 12, 4, 15})  // k_fused: INIT, TRIAL|BETA, TRIAL, ACCEPT|DIR|TRIAL|BETA
            wants.push_back({key_fused(m, big), "cgo::dev::k_fused<cgo::dev::UserObjective, " + std::to_string(m) + ", " +
                                                    (big ? "true" : "false") + ">"});
    }
    // L-BFGS in one pass over the ring per outer iteration (k_lbfgs_combine_spec, k_lbfgs_push_lite) for this objective
    for (int big = 0; big < 2; ++big) {
        for (int push = 0; push < 2; ++push)
            wants.push_back({std::string("spec:") + (big ? "1" : "0") + (push ? "1" : "0"),
                             std::string("cgo::dev::k_lbfgs_combine_spec<cgo::dev::UserObjective, ") + (big ? "true" : "false") + ", " + (push ? "true" : "false") + ">"});
        wants.push_back({std::string("lite:") + (big ? "1" : "0"),
                         std::string("cgo::dev::k_lbfgs_push_lite<cgo::dev::UserObjective, ") + (big ? "true" : "false") + ">"});
    }
    // the resident solver for this objective (cgo_kernels_resident.hip.hpp): whole outer iterations in one launch
    wants.push_back({"res:3", "cgo::dev::k_resident<cgo::dev::UserObjective, 3>"});
    for (auto &w : wants) hiprtcAddNameExpression(prog, w.expr.c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    const hiprtcResult cr = hiprtcCompileProgram(prog, 4, opts);
    size_t logsz = 0;
    hiprtcGetProgramLogSize(prog, &logsz);
    if (logsz > 1) {
        log.resize(logsz);
        hiprtcGetProgramLog(prog, &log[0]);
    }
    if (cr != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        if (log.empty()) log = hiprtcGetErrorString(cr);
        return CGO_EINVAL;
    }
    size_t codesz = 0;
    hiprtcGetCodeSize(prog, &codesz);
    std::vector<char> code(codesz);
    hiprtcGetCode(prog, code.data());
    auto mod = std::make_shared<RtcModule>();
    if (hipModuleLoadData(&mod->mod, code.data()) != hipSuccess) {
        hiprtcDestroyProgram(&prog);
        log = "hipModuleLoadData failed for the compiled user objective";
        return CGO_EHIP;
    }
    for (auto &w : wants) {
        const char *lowered = nullptr;
        if (hiprtcGetLoweredName(prog, w.expr.c_str(), &lowered) != HIPRTC_SUCCESS || !lowered) {
            hiprtcDestroyProgram(&prog);
            log = "no lowered name for " + w.expr;
            return CGO_EHIP;
        }
        hipFunction_t f = nullptr;
        if (hipModuleGetFunction(&f, mod->mod, lowered) != hipSuccess) {
            hiprtcDestroyProgram(&prog);
            log = std::string("kernel not found in module: ") + lowered;
            return CGO_EHIP;
        }
        mod->fn[w.key] = f;
    }
    hiprtcDestroyProgram(&prog);
    out = mod;
    return CGO_OK;
}

}  // namespace cgo
