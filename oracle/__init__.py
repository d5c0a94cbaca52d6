"""CPU oracle for the SANA training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product path
(``yat_amd``) never imports this package and fails loudly when the HIP
library is missing.

PARITY UNPINNED (floating-point model math).  The reference
(frutiemax92/YAT) ships no tests, fixtures or golden vectors,
and the arithmetic of its hot path lives in an *unpinned* third-party
dependency (``diffusers``, requirements.txt:7) that is absent from this
container, so the restatement of the model math below cannot be checked
against reference outputs.  What IS pinned:

* the config boundary: the host-side reader (``yat_amd/common/training_parameters_reader.py``)
  is checked in ``tests/test_params_golden.py`` against JSON produced by importing the
  reference's own ``common/training_parameters_reader.py``
  (``tests/golden/make_params_golden.py`` is the generating script);
* the RNG stream of the recipe (fresh ``torch.Generator()`` per step,
  trainer.py:325) and the AdamW / clip arithmetic: both are *stock torch*, the
  same dependency the reference calls, executed here on CPU;
* the flow-match sigma table known answers quoted in SURVEY.md §8(c).

``python -m oracle.pin_against_diffusers`` closes the gap on any machine that has the reference's dependencies: it
loads random diffusers models (SANA, PixArt-Sigma, SD3.5, the flow-match scheduler, EMAModel) into these restatements
with ``strict=True`` and compares outputs and gradients; here (no diffusers, no network) it reports "unpinned" and exits 2.
"""
