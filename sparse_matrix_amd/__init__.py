"""sparse_matrix_amd — MI355X (gfx950) back end for the spam_csr SpMV / SpGEMM hot path.

The product is ``libspam_hip.so`` (hand-written HIP, C ABI in ``include/spam_hip.h``); this package is
the thin host-side mirror of the reference's ``CsrMatrix`` API used by the tests and the bench.
Importing it does not load the library; the first device call does, and raises if it is missing.
"""
from ._lib import SpamError, SPAM_F32, SPAM_F64  # noqa: F401
from .csr import CsrMatrix, DeviceCsr  # noqa: F401
