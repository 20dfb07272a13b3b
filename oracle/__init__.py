"""CPU oracle for the discrete edge-diffusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the CPU number that is
reported beside the GPU number -- and by nothing else.  The shipped package
``semantically_consistent_graph_generation_amd`` never imports it and raises if
its HIP library is missing.

What it is: an independent eager-PyTorch (fp32, CPU) restatement of the
reference algorithm for the path SURVEY.md section 8 names (GraphTransformer denoiser,
absorbing-edge reverse step + symmetric categorical sampling, structural extra
features).  Every function cites the reference ``file:line`` it follows
(paths relative to /root/reference).

How it is pinned: ``oracle/gen_golden.py`` imports the *real* reference in the
build container (third-party modules stubbed), runs it on
seeded inputs and writes small fixtures to ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those fixtures.
The reference itself never travels to the GPU box.
"""
