"""CPU oracle for the DiDi sampler hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain fp32 torch-CPU restatement of the reference algorithm on the path named by
BASELINE.json:north_star (``src/sampling.py`` loop driving ``DiDi.forward``).  It imports
neither ``transformers`` nor ``lightning`` nor anything under ``/root/reference``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  Nothing under ``didi_amd/`` imports it.

Pinning: the reference's own tests hold no vectors for this path (its only test is
``tests/test_example.py`` = ``assert sum([1,2,3]) == 6``).  The oracle is therefore pinned by
fixtures under ``tests/golden/`` that ``oracle/make_golden.py`` generated in the build container
by running the reference's real ``src.sampling.sample`` / ``src.diffusion.utils`` (imported from
/root/reference) on HuggingFace ``BertModel`` / ``BlenderbotEncoder`` modules built from config,
with the noise stream captured.  ``tests/test_oracle_golden.py`` checks the oracle against them.

The floating-point path has no C restatement: the reference is Python/torch, so the oracle is
Python/torch (fp32, eager, CPU).

Reference files restated (file:line relative to /root/reference, ``HF:`` = transformers):
  src/diffusion/schedules.py:4-21      -> oracle/diffusion_math.py
  src/diffusion/utils.py:6-57          -> oracle/diffusion_math.py
  src/sampling.py:6-117                -> oracle/sampling.py
  src/diffusion/model.py:104-127,165-198 -> oracle/model.py
  HF:models/bert/modeling_bert.py:53-448,594-718        -> oracle/decoder.py
  HF:models/blenderbot/modeling_blenderbot.py:65-300,424-512 -> oracle/encoder.py
"""
