"""How sensitive is the REFERENCE algorithm's Davidson trajectory to roundoff?

Runs the CPU oracle (identical arithmetic to sella/eigensolvers.py) twice on the SURVEY §8(d)
recipe at n = 768, once with P perturbed by 1 ulp (relative 1e-16), and prints the lowest Ritz
value after j vectors and the difference.  Measured: 6e-14, 6e-12, 2e-9, 1e-7, 6e-6, 6e-2 at
j = 2, 4, 6, 8, 12, 16 — the (P - theta I)^-1 correction with an interior shift amplifies roundoff
~10x per iteration, so step-for-step parity is only defined for the first few vectors.
Test infrastructure (uses oracle/)."""
import sys, numpy as np
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import oracle.sella_oracle as orc
from conftest import hessian_like
n=768
A,P,g = hessian_like(n,0)
rng=np.random.RandomState(9)
P2 = P*(1+1e-16*rng.normal(size=P.shape)); P2=(P2+P2.T)/2
for j in (2,4,6,8,12,16):
    l1,_,_ = orc.rayleigh_ritz(A,0.1,P,v0=g,maxiter=j)
    l2,_,_ = orc.rayleigh_ritz(A,0.1,P2,v0=g,maxiter=j)
    print(j, l1[0], abs(l1[0]-l2[0]))
