# The reference's examples/min.jl workflow against the MI355X engine: only the objective line differs
# (a device descriptor instead of a Julia closure).  Needs an MI355X and lib/libcgo_hip.so (`make -C csrc`).
include(joinpath(@__DIR__, "..", "conjugategradientoptim.jl_amd", "julia", "ConjugateGradientOptimAMD.jl"))
const CGO = ConjugateGradientOptimAMD
using LinearAlgebra

fdf! = CGO.Booth()
x0 = [0.43; 1.23]

linesearch_config = CGO.setupStrongWolfeBisection(1e-5, 0.8; a_max_growth_factor = 2.0, max_iters = 1000, zoom_max_iters = 100)
config = CGO.setupCGConfig(1e-5, CGO.HagerZhang(), CGO.EnableTrace(); max_iters = 1000)

ret = CGO.minimizeobjective(fdf!, x0, config, linesearch_config)
@show ret.status ret.minimizer ret.objective norm(ret.gradient) ret.iters_ran sum(ret.trace.objective_evals)

# g(x) = 0 with the Hager–Zhang-type method of solve_system.jl
sys = CGO.solvesystem(CGO.QuadDiag(collect(range(1.0, 2.0; length = 1000))), ones(1000), config, CGO.setupLinesearchSolveSys(0.5))
@show sys.status sys.iters_ran
