import os, sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import fuzz_worker as FW
from test_hip_parity import _hip_index, _oracle, _random_arrays
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
seed, case = 11, 1763
rng = np.random.default_rng([seed, case])
shape = FW.draw_shape(rng)
n_docs, max_len, C, dim, nbits, B, Q, n_probe, n_full, top_k, subset, qkind = shape
print(shape)
arr = _random_arrays(rng, n_docs, max_len, C, dim, nbits, empty_frac=float(rng.choice([0.0, 0.1, 0.5])))
pick = rng.integers(0, C, (B, Q))
q = arr["centroids"][pick].astype(np.float32) + 0.3 * rng.standard_normal((B, Q, dim), dtype=np.float32) / np.sqrt(dim)
q /= np.linalg.norm(q, axis=2, keepdims=True)
q = q.astype(np.float16)
hip = _hip_index(fp, arr); orc = _oracle(arr)
params = R.SearchParameters(2000, n_full, top_k, n_probe)
for b in range(B):
    h = R.search_trace(hip, q[b], params, None)
    o = orc.search_trace(q[b], top_k, n_full, n_probe, None)
    hp, op = np.asarray(h["pids"]), np.asarray(o["pids"])
    if not np.array_equal(hp, op):
        bad = np.nonzero(hp != op)[0]
        print("query", b, "differs at", bad[:10], "n", len(hp))
        for i in bad[:6]:
            print(" pos", i, "hip", hp[i], h["scores"][i], "orc", op[i], o["scores"][i])
        om = dict(zip(op.tolist(), np.asarray(o["scores"]).tolist())); hm = dict(zip(hp.tolist(), np.asarray(h["scores"]).tolist()))
        for p in set(hp[bad].tolist()) | set(op[bad].tolist()):
            print("  doc", p, "oracle score", repr(om.get(p)), "hip score", repr(hm.get(p)), "len", arr["doc_lengths"][p])
        # exact fp32 recompute
        ex = np.asarray(o["exact"]); rr = np.asarray(o["rerank"])
        print("  rerank same:", np.array_equal(np.asarray(h["rerank"]), rr))
