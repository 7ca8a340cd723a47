"""CPU oracle for the ConfigNet GAN hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32/fp64 + an independent NumPy
restatement of the forward ops) of the arithmetic the reference performs on
the path named by BASELINE.json's north_star.  Every function cites the
reference file:line it follows (paths relative to /root/reference).

PARITY UNPINNED: the reference's arithmetic lives in tensorflow-gpu==2.1.0
(setup/requirements.txt:6), which is not vendored, not installed and not
installable here (no network), and the reference's own golden fixtures
(tests/test_assets/*.npz) need pretrained weights that are not in the tree
(SURVEY.md section 8c).  The oracle is therefore pinned only by
  (1) a double implementation (torch ops vs explicit NumPy loops, oracle/np_ops.py),
  (2) analytic known-answer tests (tests/test_oracle_kat.py),
  (3) the weight-free facts extractable from the shipped goldens
      (latent slice layout, Adam first-step magnitude).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package -- and only as the checker / timed CPU baseline, never as part of
the product path (confignet_amd/), which fails loudly without its HIP library.
"""
