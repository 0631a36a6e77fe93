import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
from oracle.oracle import OracleSimplex
lp=P.sparse_lp(5000,20000,20,seed=21)
N=int(sys.argv[1]) if len(sys.argv)>1 else 2852
o=OracleSimplex(lp); o.set_option('max_iterations',N); o.dual()
g=ClpGpuSimplex().loadProblem(lp); g.set_option('max_iterations',N); g.dual()
wo,io=o.row_weights(); wg,ig=g.rowWeights()
print('pivots same', np.array_equal(o.pivot_log()['sequenceIn'], g.pivotLog()['sequenceIn']))
print('pivotVariable same', np.array_equal(o.pivot_variable(), g.pivotVariable()))
rel=lambda a,b: np.max(np.abs(a-b)/(1e-30+np.maximum(np.abs(a),np.abs(b))))
print('weights rel', rel(wo,wg), 'infeas rel', rel(np.where(io>1e-50,io,0),np.where(ig>1e-50,ig,0)))
so,sg=o.solution(),g.solution(); print('sol rel', rel(so,sg))
ro=np.where(io>1e-14, io/wo, 0); rg=np.where(ig>1e-14, ig/wg, 0)
for name,r in (('cpu',ro),('gpu',rg)):
    idx=np.argsort(-r)[:4]; print(name,'top rows',idx,'ratios',r[idx])
import os
os.makedirs('/root/repo/gpurun_out', exist_ok=True)
np.savez_compressed('/root/repo/gpurun_out/diverge.npz', wo=wo, io=io, wg=wg, ig=ig, so=so, sg=sg, pv=o.pivot_variable(), st=o.status(), stg=g.statusArray(), djo=o.reduced_costs(), djg=g.reducedCosts())
