"""Development aid: approximate VGPR / AGPR liveness over the ISA of one kernel (hipcc -S output), treated as straight-line
code with a wrap-around for the stage loop: where is the register peak, and where were the values alive at it defined?
This is what located the three causes of the 126 spilled registers of ky_factor<24,3,4> (profiles/NOTES.md, round 3).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iacados_amd/csrc --cuda-device-only -S one.hip -o one.s
    python tools/isa_liveness.py one.s <mangled kernel name> [bucket] [from to] [peak_index]
  bucket            print max live VGPRs / AGPRs per `bucket` instructions
  from to           list the instructions of that range with the live counts in front
  peak_index        histogram of where the registers alive at that instruction were last written
"""
import re,sys
src=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(src) if l.startswith(name+':')][0]
end=[i for i,l in enumerate(src[start:]) if l.strip().startswith('.Lfunc_end')][0]+start
body=[l for l in src[start:end]]
ins=[]
for l in body:
    if l.startswith('\t') and not l.strip().startswith(('.',';')):
        ins.append(l.split(';')[0].strip())
def regs(tok):
    out=[]
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b',tok):
        if m.group(1): out+= [(m.group(1),i) for i in range(int(m.group(2)),int(m.group(3))+1)]
        else: out.append((m.group(4),int(m.group(5))))
    return out
nodef=('global_store','scratch_store','ds_write','s_waitcnt','s_nop','v_cmp','s_','buffer_store','v_writelane')  # first operand is not a def
parsed=[]
for t in ins:
    parts=t.split(None,1)
    op=parts[0]; ops=parts[1] if len(parts)>1 else ''
    opl=[o.strip() for o in ops.split(',')]
    if op.startswith(nodef) and not op.startswith('v_cmpx'):
        d=[]; u=sum((regs(o) for o in opl),[])
        if op.startswith('v_writelane'): d=regs(opl[0]); u=regs(opl[0])
    else:
        d=regs(opl[0]) if opl else []
        u=sum((regs(o) for o in opl[1:]),[])
        if op.startswith(('v_fmac','v_mac','v_accvgpr_write')) : u+=d if not op.startswith('v_accvgpr_write') else []
        if 'dpp' in op: u+=d   # old value
    parsed.append((op,set(d),set(u)))
live=set()
prof=[0]*len(parsed)
for it in range(2):
    for i in range(len(parsed)-1,-1,-1):
        op,d,u=parsed[i]
        live-=d; live|=u
        prof[i]=(sum(1 for r in live if r[0]=='v'),sum(1 for r in live if r[0]=='a'))
B=int(sys.argv[3]) if len(sys.argv)>3 else 300
for b in range(0,len(prof),B):
    seg=prof[b:b+B]
    print(b, 'max v',max(x[0] for x in seg),'max a',max(x[1] for x in seg),'max tot',max(x[0]+x[1] for x in seg))
if len(sys.argv)>4:
    lo,hi=int(sys.argv[4]),int(sys.argv[5])
    for i in range(lo,hi):
        print(i,prof[i],ins[i][:90])
if len(sys.argv)>6:
    p=int(sys.argv[6])
    # recompute live set at p
    live=set()
    for it in range(2):
        for i in range(len(parsed)-1,-1,-1):
            op,d,u=parsed[i]
            live-=d; live|=u
            if it==1 and i==p: snap=set(live)
    al=sorted(r[1] for r in snap if r[0]=='a')
    print('live agprs at',p,len(al))
    # last def before p
    import collections
    h=collections.Counter()
    for a in al:
        j=p
        while j>=0 and ('a',a) not in parsed[j][1]: j-=1
        h[(j//200)*200]+=1
    print(sorted(h.items()))
    vl=sorted(r[1] for r in snap if r[0]=='v')
    h=collections.Counter()
    for a in vl:
        j=p
        while j>=0 and ('v',a) not in parsed[j][1]: j-=1
        h[(j//200)*200]+=1
    print('live vgprs',len(vl),sorted(h.items()))
