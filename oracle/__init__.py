"""CPU oracle for the bayesgm BGM / CausalBGM hot path.

TEST INFRASTRUCTURE ONLY.  A NumPy restatement of the reference algorithm
(liuq-lab/bayesgm v1.0.2, TensorFlow 2.10 / TFP 0.18) used as the checker for
the HIP kernels.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``bayesgm_amd`` never does.

PARITY PIN STATUS
-----------------
* dataset generators / ``get_ADRF``: pinned by golden fixtures produced by
  importing the reference's own ``bayesgm.datasets`` / ``bayesgm.utils`` in the
  build container (``tests/golden/make_golden.py``).
* network / likelihood / MCMC / optimizer arithmetic: **parity unpinned** --
  TensorFlow and TFP are not installable here, the reference ships no golden
  vector or known-answer test for this path (``src/bayesgm/tests/test_models.py``
  asserts types only).  The restatement follows the cited reference lines and
  the published Keras/TFP semantics, is cross-checked against an independent
  PyTorch-CPU autograd implementation (tests/test_oracle_autograd.py), and
  against analytic truths (ADRF of the Hirano-Imbens generator).
"""
