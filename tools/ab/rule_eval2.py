import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle.oracle as oo
oo.build = lambda force=False: '/root/repo/tools/ab/liboracle_rule.so'
from oracle.oracle import OracleQp, default_opts, lib
from acados_amd.generators import random_lqr_batch, lqr_instance_qp, chain_soft_qp, multiphase_batch, multiphase_instance_qp
def run(name, qps):
    for rule in (1,2):
        its=[];fails=0
        g=(C.c_int*64).in_dll(lib(),'g_redo')
        for j in range(64): g[j]=0
        for q in qps:
            o=OracleQp(q); rc=o.solve(default_opts(tol_stat=1e-8,iter_max=50,cond_pred_corr=rule))
            fails+= rc!=0; its.append(o.iter)
        its=np.array(its)
        print(f'{name} rule {rule}: fails {fails}/{len(qps)} mean iter {its.mean():.3f} max {its.max()} hist {np.bincount(its).tolist()} redos {sum(g)}',flush=True)
n=int(sys.argv[1])
d=random_lqr_batch(N=50,batch=n,seed=0)
run('C2', [lqr_instance_qp(d,i,50) for i in range(n)])
run('C4', [chain_soft_qp(i, N=20) for i in range(min(n,256))])
d12=random_lqr_batch(N=50,nx=12,nu=3,batch=min(n,512),seed=203)
run('nx12 N50', [lqr_instance_qp(d12,i,50) for i in range(min(n,512))])
d24=random_lqr_batch(N=50,nx=24,nu=6,batch=min(n,256),seed=206)
run('nx24 N50', [lqr_instance_qp(d24,i,50) for i in range(min(n,256))])
d4=random_lqr_batch(N=50,nx=4,nu=1,batch=min(n,1024),seed=201)
run('nx4 N50', [lqr_instance_qp(d4,i,50) for i in range(min(n,1024))])
