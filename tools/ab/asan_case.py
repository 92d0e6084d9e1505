import os, sys, ctypes
ROOT='/root/repo'; sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
import numpy as np
from random_qp import random_structure_qp
from acados_amd import OcpQpGpuBatch, _lib
clib=_lib.bind(ctypes.CDLL(ROOT+'/tools/ab/libgqp_hostsim_asan.so'))
seed=int(sys.argv[1]); B=int(sys.argv[2])
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
print('kernel',b.kernel_name,flush=True)
bad=b.solve()
print('bad',bad)
