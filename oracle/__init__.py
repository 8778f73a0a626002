"""CPU oracle for the sequential-recommender training hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline. The product path (``xfmr_rec_amd``) never imports this package and
raises when the HIP library is missing.

The oracle is a plain fp32 PyTorch restatement (CPU) of the reference
algorithm -- every function cites the reference ``file:line`` it follows
(paths relative to the reference checkout; ``TF:`` = the ``transformers``
package, which holds the BERT arithmetic the reference instantiates at
``xfmr_rec/models.py:93-102``).

Pinning: the reference's own tests hold no numeric vectors
(``tests/test_recommender.py:28-63`` only asserts key presence and
``loss >= 0``), so the oracle is pinned against outputs of the reference run
in the build container: ``oracle/make_golden.py`` imports the reference's
``xfmr_rec/losses.py`` unmodified and builds the encoder exactly as
``xfmr_rec/models.py:93-102`` does (``BertModel(BertConfig(...))`` from a
local config), and writes the vectors under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks this restatement against them.
"""
