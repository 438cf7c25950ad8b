import sys, os, numpy as np
sys.path.insert(0,'/root/repo')
import fast_plaid_amd as fp
R=fp.fast_plaid_rust
spec=fp.synth.SynthSpec(n_docs=1000, doc_len=300, n_centroids=8192, seed=42)
arr=fp.synth.host_index_arrays(spec)
q=fp.synth.make_queries(spec, arr["centroids"], 16, 50)
hip=R.construct_index(arr["nbits"], arr["centroids"], None, None, arr["bucket_weights"], arr["ivf"], arr["ivf_lengths"], arr["doc_codes"], arr["doc_residuals"], arr["doc_lengths"], "cuda:0")
params=R.SearchParameters(2000,4096,10,8)
os.environ["FP_PROBE_DEBUG"]="1"
h=R.search_trace(hip,q[15],params); Sh=h['S']
def mono(x):
    u=x.view(np.uint16).astype(np.uint32); u=np.where((u&0x7FFF)==0,0,u)
    return np.where(u&0x8000, (~u)&0xFFFF, u|0x8000)
cm=Sh.reshape(8,1024,50).max(1)
for col in (17,21):
    print("numpy col",col,"cmax", [hex(int(v)) for v in cm[:,col].view(np.uint16)], "tau", hex(int(np.sort(mono(cm[:,col]))[::-1][7])), "cnt", int((Sh[:,col].astype(np.float32)>=np.sort(cm[:,col].astype(np.float32))[::-1][7]).sum()))
print("numpy chunkmax for all columns (hex):")
for col in range(50):
    print("np", col, " ".join("%04x"%int(v) for v in cm[:,col].view(np.uint16)))
