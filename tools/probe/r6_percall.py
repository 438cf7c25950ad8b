"""per-call latency of the first calls of two shapes (plain, learnt capacity, capturing call, replays)   usage: r6_percall.py [warm]
warm: touch the legacy stream through the runtime first (hipStreamSynchronize(0))"""
import ctypes, sys, time, numpy as np
sys.path.insert(0, '.')
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
spec = fp.synth.SynthSpec(n_docs=100000, doc_len=128, n_centroids=32768, variable_len=False, seed=42)
idx = R.construct_synthetic_index(spec, "cuda:0")
cent = fp.synth.centroids(spec)
if len(sys.argv) > 1:
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(4096))
    host = (ctypes.c_char * 4096)()
    t = time.perf_counter()
    if sys.argv[1] == "warm":
        rc = hip.hipStreamSynchronize(ctypes.c_void_p(0))
    elif sys.argv[1] == "memcpy":     # synchronous copy on the legacy stream
        rc = hip.hipMemcpy(dev, host, ctypes.c_size_t(4096), ctypes.c_int(1))
    elif sys.argv[1] == "memset":
        rc = hip.hipMemset(dev, 0, ctypes.c_size_t(4096))
    elif sys.argv[1] == "blocking":   # a blocking stream with work on it
        st = ctypes.c_void_p(); rc = hip.hipStreamCreate(ctypes.byref(st)); hip.hipMemsetAsync(dev, 0, ctypes.c_size_t(4096), st); hip.hipStreamSynchronize(st)
    print(sys.argv[1], rc, round((time.perf_counter() - t) * 1e3, 3), "ms")
for B in (64, 16):
    params = R.SearchParameters(2000, 4096, 1000, 8)
    for i in range(5):
        q = fp.synth.make_queries(spec, cent, B, 32, seed=100 + i)
        t = time.perf_counter(); R.search_arrays(idx, q, params); dt = (time.perf_counter() - t) * 1e3
        print(B, i, round(dt, 3), R.last_search_counts()["s4_form"], R.graph_replay_count())
