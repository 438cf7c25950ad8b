"""GPU worker for tests/test_hip_parity.py::test_native_rccl_shard_search_one_rank: fp_shard_search (the three all-gathers issued
by the library on its own stream) with ONE rank over RCCL -- all a single-GPU box allows, since RCCL wants a GPU per rank; the
multi-rank protocol itself is covered through the staged entry points (shard_gpu_worker.py, shard_mp_worker.py, gloo tests).
One rank must reproduce fp_search exactly, batch after batch of the same shape, so from the second batch on the front half runs
on the learnt candidate capacity; with FP_TEST=spec_cap_pct=50 (set by the test) every such batch overflows, says so in the first
exchange and is run again."""
import os
import sys

import torch  # FIRST (torch wheels bundle their own HIP runtime)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fptest_env import test_opt  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
from fast_plaid_amd import sharded  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    R = fp.fast_plaid_rust
    spec = fp.synth.SynthSpec(n_docs=20000, doc_len=32, n_centroids=1024, variable_len=True, seed=11)
    host = fp.synth.host_index_arrays(spec)
    index = R.construct_synthetic_index(spec, "cuda:0", centroids=host["centroids"])
    comm = sharded.NativeComm.from_torch_dist(index.device_id, dist)
    if test_opt("shard_fail_at"):
        # a stage that fails locally must neither hang nor return results: every collective is still issued, the failure travels in
        # the status word of the next exchange, and the call raises after its final sync; the communicator stays usable
        params = R.SearchParameters(2000, 512, 50, 4)
        q = fp.synth.make_queries(spec, host["centroids"], 5, 32, seed=300)
        for rep in range(2):
            try:
                sharded.native_sharded_search(index, comm, q, params)
            except ValueError as e:
                assert "failed on this rank" in str(e) and "injected failure" in str(e), str(e)
            else:
                raise AssertionError("the injected failure did not surface")
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        print("NATIVE_SHARD_OK")
        return
    for n_probe, n_full, top_k in ((4, 512, 50), (16, 1024, 100)):
        params = R.SearchParameters(2000, n_full, top_k, n_probe)
        for rep in range(4):
            q = fp.synth.make_queries(spec, host["centroids"], 5, 32, seed=300 + rep)
            got = sharded.native_sharded_search(index, comm, q, params)
            want = R.search_arrays(index, q, params)
            for x, y in zip(got, want):
                assert np.array_equal(x, y), (n_probe, rep)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("NATIVE_SHARD_OK")


if __name__ == "__main__":
    main()
