import sys; sys.path.insert(0,'/root/repo')
import numpy as np, time
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
lp=P.sparse_lp()
g=ClpGpuSimplex().loadProblem(lp)
t=time.time(); g.dual_steps(2500); dt=time.time()-t
log=g.pivotLog()
nc=log['reserved']; fl=log['numberFlipped']
print('it/s', 2500/dt)
for name,a in (('candidates',nc),('flips',fl)):
    print(name,'mean %.1f'%a.mean(),'pcts', np.percentile(a,[10,50,90,99,100]).astype(int))
for lo,hi in ((0,500),(500,1500),(1500,2500)):
    print('iters',lo,hi,'nc mean %.0f'%nc[lo:hi].mean(),'flips mean %.1f'%fl[lo:hi].mean())
