"""CPU oracle for the PPG->mel->wav hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU fp32 / NumPy) of the reference's
algorithm for the hot path named in BASELINE.json.
Each function cites the reference file:line it follows.  It exists to CHECK the HIP
path; it is never the thing that is measured or shipped:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
    may import it;
  * nothing under ``fac-via-ppg_amd/`` imports it, and the product raises when the HIP
    library is missing instead of falling back to this code.

Pinning: the reference's own tests do not touch the hot path, so the
oracle is pinned against outputs of the reference itself, imported in the dev container
by ``tests/golden/make_golden.py`` (harness shims: SURVEY.md Appendix A) with seeded
synthetic weights, captured z noise and captured prenet-dropout masks; the vectors are
committed under ``tests/golden/`` and ``tests/test_oracle_golden.py`` replays them.

Parity UNPINNED at one boundary: ``librosa==0.6.2`` (environment.yml:46) is a third-party
dependency absent from /root/reference and from this image.  ``oracle.dsp.mel_filterbank``
restates its published algorithm (Slaney scale, area normalisation); it is checked only
against the librosa documentation example value, not against librosa itself.
"""
