"""GPU worker: randomized fuzz of the index build (fp_compress + the host codec training + the IVF; create.rs:148-184, :317-364,
:404-428, :528-559) against the op-for-op ATen restatement in oracle/plaid_oracle_torch.py.

Every case draws dim, nbits, the centroid count (duplicated centroids included: exact score ties go to the lowest index), the
number of documents and their lengths, how noisy / unnormalised the tokens are -- codes, packed residual bytes, bucket cutoffs and
weights, the IVF and its lengths must be IDENTICAL; the created arrays must search like the restatement's.

usage: create_fuzz_worker.py <n_cases> <seed> [<first_case>]
"""
import os
import sys
import traceback

import torch  # FIRST (torch wheels bundle their own HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
import plaid_oracle_torch as OT  # noqa: E402
from fast_plaid_amd import create as CR  # noqa: E402
from test_hip_parity import _hip_index  # noqa: E402

R = fp.fast_plaid_rust


def draw(rng):
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    dim = int(pick(128, 128, 64, 96, 48, 32, 256))
    nbits = int(pick(4, 4, 2, 2, 1, 8))
    C = int(pick(2, 5, 16, 37, 100, 257, 512, 1000, 4096))
    n_docs = int(pick(1, 2, 10, 60, 200, 500))
    max_len = int(pick(1, 3, 20, 50, 120))
    while n_docs * max_len * C > 3e7:   # (the restatement's half matmul runs on the host)
        n_docs = max(1, n_docs // 2)
    noise = float(pick(0.0, 0.1, 0.25, 0.6, 2.0))
    scale = float(pick(1.0, 1.0, 1.0, 0.5, 2.0))   # unnormalised tokens
    dups = int(pick(0, 0, 2, 12))
    return dim, nbits, C, n_docs, max_len, noise, scale, dups


def run_case(seed, case):
    rng = np.random.default_rng([seed, case, 9])
    dim, nbits, C, n_docs, max_len, noise, scale, dups = draw(rng)
    g = torch.Generator().manual_seed(int(rng.integers(1, 1 << 30)))
    cent = torch.nn.functional.normalize(torch.randn(C, dim, generator=g), dim=-1).to(torch.float16)
    src = int(rng.integers(0, C))
    for _ in range(min(dups, C - 1)):
        cent[int(rng.integers(0, C))] = cent[src]
    docs = []
    for _ in range(n_docs):
        n = int(torch.randint(1, max_len + 1, (1,), generator=g))
        pk = torch.randint(0, C, (n,), generator=g)
        if dups:
            pk[torch.rand(n, generator=g) < 0.2] = src
        d = cent[pk].float() + noise * torch.randn(n, dim, generator=g) / dim ** 0.5
        docs.append((torch.nn.functional.normalize(d, dim=-1) * scale).to(torch.float16))
    ref = OT.build_index_arrays(docs, cent, nbits, cast_cutoffs=False)
    T = int(ref["doc_lengths"].sum())
    got = CR.build_index_arrays([d.numpy() for d in docs], cent.numpy(), nbits, "cuda:0")
    assert np.array_equal(got["doc_codes"], ref["doc_codes"].numpy()[:T]), \
        f"nearest-centroid codes differ at {int((got['doc_codes'] != ref['doc_codes'].numpy()[:T]).sum())} of {T} tokens"
    for k in ("bucket_cutoffs", "bucket_weights"):
        assert np.array_equal(got[k].view(np.uint16), ref[k].numpy().view(np.uint16)), k
    assert np.array_equal(got["doc_residuals"], ref["doc_residuals"].numpy()[:T]), "packed residual bytes differ"
    assert np.array_equal(got["ivf"], ref["ivf"].numpy()) and np.array_equal(got["ivf_lengths"], ref["ivf_lengths"].numpy()), "IVF differs"
    q = torch.stack([torch.cat([docs[i % n_docs], docs[i % n_docs][:1].expand(8, -1)])[:8] for i in range(3)]).numpy()
    params = R.SearchParameters(2000, 64, 5, min(4, C))
    a = R.search_arrays(_hip_index(fp, got), q, params)
    ref_np = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in ref.items()}
    b = R.search_arrays(_hip_index(fp, ref_np), q, params)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)), "the created arrays do not search like the restatement's"


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = 0
    for case in range(first, first + n):
        try:
            run_case(seed, case)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED case={case} seed={seed} draw={draw(np.random.default_rng([seed, case, 9]))} {type(e).__name__}: {str(e)[:400]}",
                  flush=True)
            if os.environ.get("FP_FUZZ_TRACEBACK"):
                traceback.print_exc()
    print(f"FUZZ_FAIL {bad}/{n}" if bad else f"FUZZ_OK {n}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
