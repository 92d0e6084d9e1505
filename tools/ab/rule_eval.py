import sys, ctypes as C, time
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
def instances(seed,B,take):
    nxm,num=sizes[seed%4]
    qp=random_structure_qp(seed,nx_max=nxm,nu_max=num,allow_general=(seed%5!=0),allow_slack=(seed%7!=0))
    g=np.random.default_rng(seed+9000)
    b=OcpQpGpuBatch.from_qps([qp]*B,_clib=clib)
    for k in range(qp.N+1):
        for f in ("q","r"):
            a0=b.get(f,k)
            if a0.shape[1]:
                b.set(f,k,a0*g.uniform(-2.0,3.0,(B,1))+0.3*g.standard_normal(a0.shape))
    return [b.to_qp(i) for i in take]
seeds=[int(x) for x in sys.argv[1].split(',')]
B=1536; n=int(sys.argv[2])
for seed in seeds:
    qps=instances(seed,B,range(n))
    for rule in (1,2,3):
        its=[];fails=0
        g=(C.c_int*64).in_dll(lib(),'g_redo')
        for j in range(64): g[j]=0
        for q in qps:
            o=OracleQp(q); rc=o.solve(default_opts(tol_stat=1e-8,iter_max=80,cond_pred_corr=rule))
            fails+= rc!=0; its.append(o.iter)
        its=np.array(its)
        print(f'seed {seed} rule {rule}: fails {fails}/{n} mean iter {its.mean():.2f} max {its.max()} redos {sum(g)}',flush=True)
