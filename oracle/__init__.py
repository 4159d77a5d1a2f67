"""CPU oracle for the GE2E loss hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path
(``speaker_embedding_ge2e_loss_amd``) never imports, calls or falls back to
anything in here; it fails loudly when the HIP library is missing.

Parity status: PINNED for the softmax loss (eq. 6) -- both restatements in
``ge2e_oracle`` are checked against golden vectors produced by importing the
reference's own ``embedding_model_GE2E/s3_loss_function_GE2E.py`` on CPU
(``tests/golden/make_golden.py``).  UNPINNED for the contrast loss (eq. 7):
the reference does not implement it, so it is defined from arXiv:1710.10467
and pinned only by the agreement of the two independent restatements here.
"""
from .ge2e_oracle import (  # noqa: F401
    expand_form_cos_sim,
    expand_form_loss,
    expand_form_loss_and_grads,
    closed_form,
    centroids,
)
