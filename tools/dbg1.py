import sys, ctypes, numpy as np
sys.path.insert(0,'/root/repo')
import fast_plaid_amd as fp
from fast_plaid_amd import _native
out=(ctypes.c_uint64*16)()
_native.check(_native.lib().fp_selftest_arith(0, ctypes.cast(out, ctypes.c_void_p)))
print("mismatch compensated",out[0],"add",out[1],"plain",out[2],"samples",out[3])
for i in range(min(12,out[3])):
    v=out[4+i]; eb=v&0xFFFF; nb=(v>>16)&0xFFFF; qf=(v>>32)&0xFFFF; qr=(v>>48)&0xFFFF
    e=np.array([eb],np.uint16).view(np.float16)[0]; n=np.array([nb],np.uint16).view(np.float16)[0]
    f=np.array([qf],np.uint16).view(np.float16)[0]; r=np.array([qr],np.uint16).view(np.float16)[0]
    true=np.float16(np.float32(e)/np.float32(n)); ex=float(e)/float(n)
    print(f"e={float(e)!r} n={float(n)!r} fast={float(f)!r} gpu_ref={float(r)!r} numpy_h(f32 div)={float(true)!r} exact={ex!r}")
# unique-code ratio at cfg2
R=fp.fast_plaid_rust
spec=fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, seed=42)
h=fp.synth.host_docs(spec, np.arange(0,2000))
o=0; u=[]
for l in h['doc_lengths']:
    u.append(len(set(h['doc_codes'][o:o+int(l)].tolist()))); o+=int(l)
print("mean unique codes per 128-token doc:", np.mean(u))
