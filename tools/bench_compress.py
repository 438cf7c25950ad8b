"""Throughput of fp_compress (exact nearest-centroid assignment + residual packing) at index-build sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fast_plaid_amd as fp
from fast_plaid_amd import create as CR

for C, T in ((8192, 1 << 18), (131072, 1 << 18)):
    rng = np.random.default_rng(0)
    cent = rng.standard_normal((C, 128), dtype=np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    emb = cent[rng.integers(0, C, T)] + 0.03 * rng.standard_normal((T, 128), dtype=np.float32)
    emb = (emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float16)
    cut = np.linspace(-0.05, 0.05, 15).astype(np.float16)
    CR.compress(cent, cut, emb[:4096], 4)
    t = time.perf_counter(); codes, res = CR.compress(cent, cut, emb, 4); dt = time.perf_counter() - t
    print("C=%d T=%d: %.3f s  %.2f Mtok/s  %.1f TFLOP/s useful (2 T C D / time; FP_TEST=assign_exact=1 forces the all-VALU kernel)" % (C, T, dt, T / dt / 1e6, 2.0 * T * C * 128 / dt / 1e12))
