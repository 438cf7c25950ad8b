"""GPU worker for tests/test_hip_parity.py::test_probe_fallback_path_forced: runs with
FP_PROBE_FALLBACK=1 (read once per process by the library), so every probe goes through the
register top-k fallback instead of the threshold path; results must match the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
import plaid_oracle as OC  # noqa: E402
from parity import check_trace  # noqa: E402


def main():
    assert os.environ.get("FP_PROBE_FALLBACK") == "1"
    R = fp.fast_plaid_rust
    for n_probe, Q, C in ((8, 32, 4096), (1, 20, 1024), (32, 40, 2048)):
        spec = fp.synth.SynthSpec(n_docs=2500, doc_len=48, n_centroids=C, variable_len=True, seed=11 + n_probe)
        arr = fp.synth.host_index_arrays(spec)
        q = fp.synth.make_queries(spec, arr["centroids"], 5, Q)
        hip = R.construct_index(arr["nbits"], arr["centroids"], None, None, arr["bucket_weights"], arr["ivf"], arr["ivf_lengths"],
                                arr["doc_codes"], arr["doc_residuals"], arr["doc_lengths"], "cuda:0", False)
        orc = OC.OracleIndex(nbits=arr["nbits"], centroids=arr["centroids"], bucket_weights=arr["bucket_weights"], ivf=arr["ivf"],
                             ivf_lengths=arr["ivf_lengths"], doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"],
                             doc_lengths=arr["doc_lengths"])
        params = R.SearchParameters(2000, 256, 20, n_probe)
        for b in range(q.shape[0]):
            h = R.search_trace(hip, q[b], params)
            o = orc.search_trace(q[b], 20, 256, n_probe)
            check_trace(h, o, Q, n_probe, 256, 20)
    print("PROBE_FALLBACK_OK")


if __name__ == "__main__":
    main()
