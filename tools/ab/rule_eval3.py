import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle.oracle as oo
oo.build = lambda force=False: '/root/repo/tools/ab/liboracle_rule.so'
from oracle.oracle import OracleQp, default_opts, lib
from random_qp import random_structure_qp
from acados_amd import OcpQpGpuBatch, _lib
from hostsim.build import build
clib=_lib.bind(C.CDLL(build()))
sizes=[(6,3),(12,4),(24,6),(40,8)]
lo,hi,B=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
tot={1:[0,0,0],2:[0,0,0]}
for seed in range(lo,hi):
    nxm,num=sizes[seed%4]
    qp=random_structure_qp(seed,nx_max=nxm,nu_max=num,allow_general=(seed%5!=0),allow_slack=(seed%7!=0))
    g=np.random.default_rng(seed+9000)
    b=OcpQpGpuBatch.from_qps([qp]*B,_clib=clib)
    for k in range(qp.N+1):
        for f in ("q","r"):
            a0=b.get(f,k)
            if a0.shape[1]:
                b.set(f,k,a0*g.uniform(-2.0,3.0,(B,1))+0.3*g.standard_normal(a0.shape))
    qps=[b.to_qp(i) for i in range(B)]
    line=f'seed {seed}:'
    for rule in (1,2):
        its=[];fails=0
        for q in qps:
            o=OracleQp(q); rc=o.solve(default_opts(tol_stat=1e-8,iter_max=80,cond_pred_corr=rule))
            fails+= rc!=0
            if rc==0: its.append(o.iter)
        tot[rule][0]+=fails; tot[rule][1]+=sum(its); tot[rule][2]+=len(its)
        line+=f'  rule {rule}: fails {fails} mean {np.mean(its):.2f} max {max(its)}'
    print(line,flush=True)
for r in (1,2): print('rule',r,'fails',tot[r][0],'of',(hi-lo)*B,'mean iter of converged',tot[r][1]/tot[r][2])
