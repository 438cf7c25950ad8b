cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 900 "$@" 2>&1 | grep -v amdgpu.ids | grep "FUZZ_" | tail -4 | cut -c1-400; }
run python tests/fuzz_worker.py 4000 11
for f in q8 l0 l0h; do FP_APPROX_IMPL=$f run python tests/fuzz_worker.py 1000 12; done
run python tests/fuzz_worker.py 300 13 0 big
run python tests/fuzz_worker.py 600 14 0 stateful
run python tests/fuzz_worker.py 600 15 0 stateful
run python tests/fuzz_worker.py 200 16 0 threads
run python tests/fuzz_worker.py 60 17 0 huge
run python tests/shard_fuzz_worker.py 2000 18
FP_TEST=shard_big=1 run python tests/shard_fuzz_worker.py 1000 19
run python tests/shard_fuzz_worker.py 300 20 0 native
run python tests/maintain_fuzz_worker.py 3000 21
run python tests/create_fuzz_worker.py 100 22
