"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the mDT hot path.

Nothing under ``oracle/`` is part of the shipped product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and there only as the checker.  The product package
(``multimodaldiscussiontransformer_amd``) never imports this package and fails loudly if
its HIP extension is missing.

Parity status: the float restatement (``mdt_ref_cpu``) and the integer restatement
(``structure``) are pinned against golden vectors produced by importing the *real*
reference modules in the build container (``oracle/gen_golden.py`` →
``tests/golden/*.npz``).  Third-party arithmetic the reference delegates to
(HF ``transformers`` BertLayer/ViTLayer, fairseq LayerNorm/softmax/gelu) is not pinned
by any reference-side test ("parity unpinned" at that boundary); there
the oracle follows the installed transformers 5.15 eager math, whose outputs are part
of the golden vectors.
"""
