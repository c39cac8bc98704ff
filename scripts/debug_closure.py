import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _cases import Case, run_oracle, run_gpu, O, first_divergence
n=1001
p=O.fill_uniform(n,3,0.5,4.0); x0=O.fill_uniform(n,4,-2.0,2.0); s0=0.75
def fdf(g,x):
    x2=x*x; g[:]=x2*x+p*x-s0
    return float(np.sum(0.25*(x2*x2)+0.5*(p*x2)-s0*x))
combos=[("DaiYuan",dict(c1=1e-5,c2=0.8)),("PolakRibiere",dict(c1=1e-5,c2=0.1)),("HagerZhang",dict(ls="WolfeBisection",cond="Wolfe",c1=1e-3,c2=0.9,ls_max_iters=100)),
 ("DaiYuan",dict(ls="Backtracking",c1=1e-3,discount=0.5,ls_max_iters=100,feas_max_iters=50)),("LBFGS",dict(c1=1e-5,c2=0.9,m=5))]
for b,kw in combos:
    c=Case("q","closure",n,x0,beta=b,eps=1e-12,max_iters=12,extra={"fdf":fdf},**kw)
    a=run_oracle(c); h=run_gpu(c)
    d=first_divergence(h,a,1e-12)
    print(b,kw.get("ls","SW"),len(a.log_a),len(h.log_a),a.status,h.status,a.iters_ran,h.iters_ran,"div",d)
    if d is not None:
        lo=max(0,d-2)
        print("  ref a", a.log_a[lo:d+3], "phi", a.log_phi[lo:d+3], "dphi", a.log_dphi[lo:d+3])
        print("  gpu a", h.log_a[lo:d+3], "phi", h.log_phi[lo:d+3], "dphi", h.log_dphi[lo:d+3])
        print("  evals ref", a.trace_objective_evals, "gpu", h.trace_objective_evals)
