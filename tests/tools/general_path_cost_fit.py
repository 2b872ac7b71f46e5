"""(study) what predicts a general-path QP's cost from its inputs?  Reads general_path_cost_data.py's file.  -> profiles/r05_general_path_order.txt"""
import numpy as np, sys, os
h=int(sys.argv[1]) if len(sys.argv)>1 else 10
d=np.load(f"gpurun_out/study_gen_h{h}.npz")
it=d["it"]; nf=d["nf"]; cost=it+10*nf
x0=d["x0"]; xref=d["xref"].reshape(len(it),h,-1); c=d["contact"].reshape(len(it),h,4).astype(float)
ns=xref.shape[2]
print("ns",ns)
e=xref[:,0,:]-x0[:,:ns] if x0.shape[1]>=ns else None
# state order: euler(3) pos(3) angvel(3) linvel(3) g
evx,evy,evz=np.abs(e[:,9]),np.abs(e[:,10]),np.abs(e[:,11])
evz_s=e[:,11]
all0=c[:,0,:].all(1).astype(float)
hard=30*evz_s+27*np.hypot(evx,evy)+10*all0
def makespan(order_key, rows):
    # greedy list scheduling longest-first by key onto rows
    import heapq
    idx=np.argsort(-order_key,kind="stable")
    heap=[0.0]*rows; heapq.heapify(heap)
    for i in idx:
        t=heapq.heappop(heap); heapq.heappush(heap,t+cost[i])
    return max(heap)
rows=256*6
print("mean load",cost.sum()/rows,"max",cost.max())
print("makespan: perfect",makespan(cost.astype(float),rows)," current",makespan(hard,rows)," random",makespan(np.random.default_rng(0).random(len(it)),rows), "const", makespan(np.zeros(len(it)),rows))
print("corr current", np.corrcoef(hard,cost)[0,1])
nst=c.sum(2)            # stance legs per step
feat={
 "evz":evz_s,"evxy":np.hypot(evx,evy),"all0":all0,
 "nst_mean":nst.mean(1),"nst_min":nst.min(1),"nst0":nst[:,0],"nst_last":nst[:,-1],
 "allsteps":(nst==4).mean(1),"zero":(nst==0).mean(1),"one":(nst==1).mean(1),"two":(nst==2).mean(1),"three":(nst==3).mean(1),
 "switches":(np.abs(np.diff(c,axis=1)).sum((1,2))),
 "absevz":evz,
}
for k,v in feat.items(): print(k, round(np.corrcoef(v,cost)[0,1],3))
X=np.column_stack([feat[k] for k in feat]+[np.ones(len(it))])
w,*_=np.linalg.lstsq(X,cost,rcond=None)
pred=X@w
print("lstsq corr",np.corrcoef(pred,cost)[0,1]); print(dict(zip(list(feat)+["1"],np.round(w,2))))
print("makespan lstsq",makespan(pred,rows))
print("---- richer")
from sklearn.ensemble import GradientBoostingRegressor
from sklearn.model_selection import train_test_split
ee=e[:,:12]
F=np.column_stack([ee,np.abs(ee),nst,c.reshape(len(it),-1).reshape(len(it),h,4).sum(1), nst.min(1), (nst<=1).sum(1),(nst==0).sum(1)])
names=[f"e{i}" for i in range(12)]+[f"|e{i}|" for i in range(12)]+[f"nst{t}" for t in range(h)]+[f"leg{i}" for i in range(4)]+["nstmin","le1","eq0"]
Xtr,Xte,ytr,yte=train_test_split(F,cost,test_size=0.3,random_state=0)
m=GradientBoostingRegressor(n_estimators=300,max_depth=4).fit(Xtr,ytr)
p=m.predict(Xte); print("gbr test corr",np.corrcoef(p,yte)[0,1])
imp=sorted(zip(m.feature_importances_,names),reverse=True)[:12]; print(imp)
pall=m.predict(F); print("makespan gbr (in-sample mostly)",makespan(pall,rows))
