"""CPU oracle for the Bind-Your-Avatar denoise-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker (never as the thing measured or
shipped).  The product path (``bind_your_avatar_implementation_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * reference-OWNED arithmetic (``models/transformer.py`` forward control flow,
    ``models/router.py``, ``models/audio_model.py``) is pinned against the
    reference itself, imported in the build container by
    ``tests/golden/make_golden.py``; fixtures live in ``tests/golden``.
  * arithmetic that lives in the un-vendored third-party dependency
    ``diffusers==0.34.0.dev0`` (``requirements.txt:23``: Attention + processors,
    FeedForward, CogVideoXLayerNormZero, AdaLayerNorm, CogVideoXPatchEmbed,
    Timesteps, TimestepEmbedding, apply_rotary_emb) is restated in
    ``oracle/layers.py`` from the library's published algorithm.  The reference
    repo holds no tests / golden vectors for that boundary, so for those layers
    the oracle is **parity unpinned**.
"""
