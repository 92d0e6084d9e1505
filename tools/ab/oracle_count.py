import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle.oracle import OracleQp, default_opts
from random_qp import random_structure_qp
from acados_amd import OcpQpGpuBatch, _lib
from hostsim.build import build
clib=_lib.bind(C.CDLL(build()))
sizes=[(6,3),(12,4),(24,6),(40,8)]
for seed in [int(x) for x in sys.argv[1].split(',')]:
    B=1536
    nxm,num=sizes[seed%4]
    qp=random_structure_qp(seed,nx_max=nxm,nu_max=num,allow_general=(seed%5!=0),allow_slack=(seed%7!=0))
    g=np.random.default_rng(seed+9000)
    b=OcpQpGpuBatch.from_qps([qp]*B,_clib=clib)
    for k in range(qp.N+1):
        for f in ("q","r"):
            a0=b.get(f,k)
            if a0.shape[1]:
                b.set(f,k,a0*g.uniform(-2.0,3.0,(B,1))+0.3*g.standard_normal(a0.shape))
    fails=[]
    for i in range(B):
        o=OracleQp(b.to_qp(i))
        if o.solve(default_opts(tol_stat=1e-8,iter_max=80))!=0: fails.append(i)
    print('seed',seed,'oracle failures',len(fails),fails[:10],flush=True)
