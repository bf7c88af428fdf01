"""CPU oracle for the MI355X Shapley-attribution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The shipped path
(``group-attribution-for-diffusion-models_amd/gad``) never routes through it and
raises if the HIP extension is missing.

Parity status:
  * coalition samplers, Shapley/Banzhaf solvers, config registry:
      PINNED by golden vectors generated from the reference's own code
      (tests/golden/make_golden.py imports /root/reference in the build
      container).
  * U-Net / scheduler / EMA / LoRA numerics: the arithmetic lives in the
      un-vendored third-party dependency diffusers==0.24.0 (reference
      requirements.txt:5) which is not installable here and the reference has no
      tests for it -> "parity unpinned" by the reference; the restatement follows
      the published diffusers-0.24.0 algorithms and is pinned by closed-form
      known-answer tests (param count 35 746 307, beta/alpha tables, sinusoid
      embedding, DDIM step, EMA decay sequence).
"""
