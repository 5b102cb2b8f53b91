"""CPU oracle for the occlusions-4d hot path (encode + cross-attention decode).

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product path
(``occlusions-4d_amd/``) never imports it and has no CPU fallback.

What it is: a plain PyTorch-CPU restatement (own code) of the reference's
algorithm for the path SURVEY.md §8(a) names, one function per reference
function, each citing the reference ``file:line`` it follows.

How it is pinned: ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` in the build container (with the absent third-party modules
stubbed), runs it on seeded inputs and commits inputs + outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors.  One boundary stays **parity unpinned**: the reference
calls ``torch_cluster.fps`` / ``torch_cluster.knn`` (rusty1s/pytorch_cluster,
version unpinned, source not under /root/reference).  ``oracle/cluster.py``
restates their documented semantics; the golden vectors were produced with that
restatement standing in for ``torch_cluster``.
"""
