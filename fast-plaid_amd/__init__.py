"""MI355X-native PLAID search engine: drop-in for the fast-plaid *search* hot path.

Layout:
  csrc/                 hand-written HIP kernels (gfx950) + host engine + C ABI (include/fastplaid.h)
  _native.py            ctypes loader of libfastplaid_hip.so (fails loudly when it is missing)
  fast_plaid_rust.py    mirror of the reference's PyO3 module `fast_plaid.fast_plaid_rust`
                        (rust/lib.rs:366-383): construct_index / pysearch / SearchParameters / QueryResult
  search/               mirror of `fast_plaid.search` (class FastPlaid)
  synth.py              synthetic compressed-domain corpora (numpy twin of csrc/fp_synth.hip)
  sharded.py            one-process-per-GPU document-sharded / replicated search over torch.distributed (RCCL)
  kmeans.py             k-means centroids (sampling protocol of the reference, Lloyd with device assignment)
  maintain.py           update (append with the existing codec) / delete, the native part of the reference
  create.py             index creation given centroids: codec training, device compression (fp_compress), IVF, directory
"""
from . import create, fast_plaid_rust, kmeans, maintain, search, synth  # noqa: F401

__all__ = ["create", "fast_plaid_rust", "kmeans", "maintain", "search", "synth"]
