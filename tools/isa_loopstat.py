"""hot-loop statistics of the persistent ADMM kernel in a -save-temps listing (tools/one_kernel.sh): instruction mix of the largest self-loop, cut at its back-edge.
usage: isa_loopstat.py listing.s ...  (writes listing.s.loop)"""
import re,sys,collections
def stat(path,kern="a1mpc_admm_kernel"):
    src=open(path).read()
    m=re.search(r"\n(_Z\w*%s\w*):"%kern,src); i=m.start(1); j=src.index(".end_amdhsa_kernel",i)
    blocks=re.split(r"\n(?=\.LBB\d+_\d+:)",src[i:j])
    best=None
    for b in blocks:
        name=b.split(":")[0]
        mm=re.search(r"s_cbranch\S*\s+%s\b"%re.escape(name),b)
        if mm:
            b=b[:mm.end()]
            if best is None or len(b)>len(best): best=b
    ops=[l.split()[0] for l in best.split("\n") if l.strip() and l.strip()[0] not in ";./" and not l.strip().split()[0].endswith(":")]
    c=collections.Counter(ops)
    f64=sum(v for k,v in c.items() if 'f64' in k)
    return best,dict(instrs=len(ops),f64=f64,ds=sum(v for k,v in c.items() if k.startswith('ds_')),scratch=sum(v for k,v in c.items() if k.startswith('scratch')),acc=sum(v for k,v in c.items() if 'accvgpr' in k),add_u32=c['v_add_u32_e32']+c['v_add_u32'],lshl_add=c['v_lshl_add_u32'],nop=c['s_nop'],waitcnt=c['s_waitcnt'],mov32=c['v_mov_b32_e32'],mov64=c['v_mov_b64']+c.get('v_mov_b64_e32',0),movdpp=c['v_mov_b32_dpp'],swap=c['v_permlane32_swap_b32_e32'],cnd=c['v_cndmask_b32_e64'])
if __name__=="__main__":
    for p in sys.argv[1:]:
        b,s=stat(p)
        open(p+".loop",'w').write(b)
        print(p,s)
