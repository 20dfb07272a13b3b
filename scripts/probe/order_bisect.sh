#!/bin/bash
# which earlier test file changes the result of the chain test when run in the same process
for f in config2 denoiser determinism eigen_tridiag extra_features graph3d large_graphs modes projector reverse_step; do
  timeout -k 10 300 python -u -m pytest tests/test_gpu_$f.py "tests/test_gpu_sampling.py::test_chain_matches_cpu_oracle_bit_exactly" -q --timeout 180 --timeout-method thread -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/$f: /"
done
