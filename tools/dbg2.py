import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import fast_plaid_amd as fp, plaid_oracle as OC
R=fp.fast_plaid_rust
spec=fp.synth.SynthSpec(n_docs=1000, doc_len=300, n_centroids=8192, seed=42)
arr=fp.synth.host_index_arrays(spec)
q=fp.synth.make_queries(spec, arr["centroids"], 16, 50)
hip=R.construct_index(arr["nbits"], arr["centroids"], None, None, arr["bucket_weights"], arr["ivf"], arr["ivf_lengths"], arr["doc_codes"], arr["doc_residuals"], arr["doc_lengths"], "cuda:0")
orc=OC.OracleIndex(nbits=4, centroids=arr['centroids'], bucket_weights=arr['bucket_weights'], ivf=arr['ivf'], ivf_lengths=arr['ivf_lengths'], doc_codes=arr['doc_codes'], doc_residuals=arr['doc_residuals'], doc_lengths=arr['doc_lengths'])
params=R.SearchParameters(2000,4096,10,8)
for b in range(16):
    h=R.search_trace(hip,q[b],params); o=orc.search_trace(q[b],10,4096,8)
    a=set(h['cells'].tolist()); g=set(o['cells'].tolist())
    if a!=g:
        S=o['S'].astype(np.float32); Sh=h['S'].astype(np.float32)
        print("query",b,"only hip",sorted(a-g),"only ref",sorted(g-a), "S equal:", np.array_equal(h['S'].view(np.uint16), o['S'].view(np.uint16)))
        for c in sorted(a^g):
            col=int(np.argmax(Sh[c])); kth=np.sort(Sh[:,col])[::-1][:9]
            print("  cell",c,"best col",col,"val",Sh[c,col],"top9 of col",kth)
print("done")
# analyse: per column tau / count>=tau from hip's own S for the mismatching queries
for b in (12, 15):
    h=R.search_trace(hip,q[b],params); Sh=h['S'].astype(np.float32)
    cm=Sh.reshape(8,1024,50).max(1)              # [chunk][q]
    tau=np.sort(cm,axis=0)[::-1][7]              # 8th largest chunk max per column
    cnt=(Sh>=tau[None,:]).sum(0)
    print("query",b,"max count>=tau:",cnt.max(),"cols over 64:",int((cnt>64).sum()), "counts sample", cnt[:10])
    want=set()
    for col in range(50):
        idx=np.lexsort((np.arange(8192), -Sh[:,col]))[:8]; want|=set(idx.tolist())
    print("  expected cells from hip S:",len(want),"hip returned:",len(h['cells']), "missing:",sorted(want-set(h['cells'].tolist())))
