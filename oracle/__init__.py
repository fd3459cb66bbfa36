"""CPU oracle for the MuseDiffusion hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / torch-CPU fp32 restatement of the reference's
algorithm (GaussianDiffusion sampling + training losses driving the BERT-style
denoiser).  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``musediffusion_amd/`` imports it, and the product path raises when the HIP
library is missing instead of falling back here.

Parity pinning: the reference ships no tests / golden vectors,
so the oracle is pinned against outputs of the reference's own Python imported
in the build container (``tools/make_golden.py`` -> ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` replays every fixture through this package.

The Transformer layers of the reference live in the third-party dependency
``transformers`` (pinned 4.22.2 in the reference's requirements.txt:15, source
not under /root/reference).  ``oracle.denoiser`` restates the published
BertEncoder algorithm (post-LN, exact-erf GELU, eps 1e-12, scores / sqrt(dh))
and is pinned on the fixtures generated through the reference's call sites
(models/network.py:74, :151) with the transformers 5.15.0 eager BertEncoder
installed in the build container.
"""
from . import schedule, denoiser, sampling, losses  # noqa: F401
