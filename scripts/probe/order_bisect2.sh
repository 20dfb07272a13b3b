#!/bin/bash
CH="tests/test_gpu_sampling.py::test_chain_matches_cpu_oracle_bit_exactly"
for t in "test_gpu_reverse_step.py::test_philox_stream_matches_numpy_twin" "test_gpu_reverse_step.py::test_absorbing_properties_at_scale" "test_gpu_reverse_step.py::test_sampler_without_topk_matches_reference" "test_gpu_reverse_step.py::test_fused_reverse_step_matches_reference" "test_gpu_reverse_step.py::test_sampler_bit_exact_on_reference_probabilities" "test_gpu_extra_features.py::test_jacobi_eigensolver_properties" "test_gpu_extra_features.py::test_cycle_features_off_zeroes_the_cycle_channels" "test_gpu_extra_features.py::test_features_at_config2_scale_are_consistent"; do
  timeout -k 10 300 python -u -m pytest "tests/$t" "$CH" -q --timeout 180 --timeout-method thread -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/$t: /"
done
