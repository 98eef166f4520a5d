import os, sys, time
import numpy as np
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/tests"]
from graphvite_amd import synthetic
from oracle_lib import Oracle, ReferenceSolver, link_prediction_auc, reference_train
model=sys.argv[1]; seed=int(sys.argv[2]); epochs=int(sys.argv[3]) if len(sys.argv)>3 else 50
chunk, ras = {"sequential": (0, False), "lock_step": (5120, False), "reads_at_start": (5120, True)}[model]
N,E,B=1000000,10000000,100000
oracle=Oracle()
edges=synthetic.power_law_edges(N,E,seed=1024)
train,(valid,test)=synthetic.link_prediction_split(edges,(100,1,1))
t0=time.time()
rs=ReferenceSolver(oracle, seed, train.astype(np.uint32), None, True, 1, 4, 1, 1, B, 0)
kw=dict(kernel_chunk=chunk, threads=int(os.environ.get("THREADS","3")), reads_at_start=ras) if chunk else {}
vertex,context,batch_id=reference_train(rs,"LINE",epochs,augmentation_step=1,**kw)
labels=rs.partition()[0]
name2id=np.full(int(labels.max())+1,-1,np.int64); name2id[labels]=np.arange(len(labels))
H,T,Y=(np.asarray(x) for x in test)
ok=(H<=labels.max())&(T<=labels.max()); H,T,Y=H[ok],T[ok],Y[ok]
ok=(name2id[H]>=0)&(name2id[T]>=0)
auc=link_prediction_auc(vertex,context,name2id[H[ok]],name2id[T[ok]],Y[ok])
print("C2 %s seed %d epochs %d: %d batches AUC %.6f %.0f s"%(model,seed,epochs,batch_id,auc,time.time()-t0),flush=True)
