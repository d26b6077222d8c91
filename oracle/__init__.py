"""CPU oracle for the SlotFormer hot path (SAVi/STEVE slot extraction + rollout).

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 (optionally fp64)
PyTorch-CPU restatement of the reference algorithm.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``slotformer_amd/`` imports
it, and the product path raises if the HIP extension is missing.

Parity status: PINNED.  Every function here is checked against outputs of the
reference's own classes (imported from /root/reference in the authoring
container by ``tools/gen_golden.py``); the resulting vectors are committed
under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
The one exception is the CNN layer convention that lives in the un-vendored
third-party package ``nerv`` v0.1.0 (conv bias on/off, Sequential nesting):
"parity unpinned" for that detail only -- see DESIGN.md.
"""
from .slotformer_oracle import *  # noqa: F401,F403
