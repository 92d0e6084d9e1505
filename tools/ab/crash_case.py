import os, sys, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv)>1 and sys.argv[1]=="child":
    sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
    import numpy as np
    from random_qp import random_structure_qp
    from acados_amd import OcpQpGpuBatch
    seed=int(sys.argv[2]); B=int(sys.argv[3]); cpc=int(sys.argv[4])
    sizes=[(6,3),(12,4),(24,6),(40,8)]
    nxm,num=sizes[seed%4]
    qp=random_structure_qp(seed,nx_max=nxm,nu_max=num,allow_general=(seed%5!=0),allow_slack=(seed%7!=0))
    g=np.random.default_rng(seed+9000)
    b=OcpQpGpuBatch.from_qps([qp]*B)
    for k in range(qp.N+1):
        for f in ("q","r"):
            a0=b.get(f,k)
            if a0.shape[1]:
                b.set(f,k,a0*g.uniform(-2.0,3.0,(B,1))+0.3*g.standard_normal(a0.shape))
    for f in ("tol_stat","tol_eq","tol_ineq","tol_comp"): b.opts_set(f,1e-8)
    b.opts_set("iter_max",80); b.opts_set("cond_pred_corr",cpc)
    if len(sys.argv)>5: b.opts_set("print_level",int(sys.argv[5]))
    bad=b.solve()
    print("OK kernel",b.kernel_name,"bad",bad,"iters",int(b.info('iter').min()),int(b.info('iter').max()),"res",float(b.res_compute().max()),flush=True)
    sys.exit(0)
seed=sys.argv[1] if len(sys.argv)>1 else "7004"
for B in ("1536","64","256","1024"):
    for env,cpc in (({},1),({"ACADOS_AMD_EXT_UPDATE":"0"},1),({},0),({"ACADOS_AMD_W16T_GEN":"0"},1),({"ACADOS_AMD_WPI":"1"},1)):
        r=subprocess.run([sys.executable,__file__,"child",seed,B,str(cpc)],env=dict(os.environ,**env),capture_output=True,text=True)
        out=[l for l in r.stdout.splitlines() if l.startswith("OK")]
        print("B",B,env,"cond_pred_corr",cpc,"->",out[0] if out else "CRASH rc %d: %s"%(r.returncode,(r.stderr.strip().splitlines() or ['?'])[-1][:120]),flush=True)
