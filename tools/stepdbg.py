import sys; sys.path.insert(0,'/root/repo')
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
lp=P.sparse_lp(2000,8000,10,seed=3)
g=ClpGpuSimplex().loadProblem(lp); g.set_option('check_every',16); g.set_option('max_pivots',0)
tot=0
for s in (200,100,37,500,16,1,300):
    st=g.dual_steps(s); tot+=s
    print('asked',s,'status',st,'iterations',g.numberIterations(),'expected',tot)
