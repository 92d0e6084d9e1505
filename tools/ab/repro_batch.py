import os, sys, ctypes
ROOT='/root/repo'; sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
import numpy as np
from random_qp import random_structure_qp
from acados_amd import OcpQpGpuBatch, _lib
from oracle.oracle import OracleQp, default_opts
from hostsim.build import build
clib=_lib.bind(ctypes.CDLL(build()))
seed=int(sys.argv[1]); B=int(sys.argv[2]) if len(sys.argv)>2 else 1536
sizes=[(6,3),(12,4),(24,6),(40,8)]
nxm,num=sizes[seed%4]
qp=random_structure_qp(seed,nx_max=nxm,nu_max=num,allow_general=(seed%5!=0),allow_slack=(seed%7!=0))
g=np.random.default_rng(seed+9000)
b=OcpQpGpuBatch.from_qps([qp]*B,_clib=clib)
for k in range(qp.N+1):
    for f in ("q","r"):
        a0=b.get(f,k)
        if a0.shape[1]:
            b.set(f,k,a0*g.uniform(-2.0,3.0,(B,1))+0.3*g.standard_normal(a0.shape))
for f in ("tol_stat","tol_eq","tol_ineq","tol_comp"): b.opts_set(f,1e-8)
b.opts_set("iter_max",80)
bad=b.solve()
st=b.info('status'); it=b.info('iter')
print('kernel',b.kernel_name,'bad',bad,'failing',np.nonzero(st)[0][:10], 'N',qp.N)
for i in list(np.nonzero(st)[0][:3]):
    qi=b.to_qp(int(i)); o=OracleQp(qi)
    rc=o.solve(default_opts(tol_stat=1e-8,iter_max=80))
    print('instance',i,'device iters',it[i],'res',[float(b.info(n)[i]) for n in ("res_stat","res_eq","res_ineq","res_comp")],'oracle rc',rc,'iters',o.iter)
    s=o.stat() if hasattr(o,'stat') else None
    if s is not None: print(np.array2string(s[-6:,:8],precision=2))
    ds=b.stat(int(i))
    print('device stat tail'); print(np.array2string(ds[-6:,:11],precision=2))
