"""pailliercryptolib_python_amd — the reference's Python API (``ipcl_python``) on an MI355X-native engine.

Public names mirror ``src/ipcl_python/__init__.py:4-11`` of the reference:
``PaillierKeypair, PaillierPublicKey, PaillierPrivateKey, PaillierEncryptedNumber`` plus the inert
``context / hybridControl / hybridMode`` QAT shims.  Everything below the API is new: a C-ABI shared
library (``lib/libpaillier_hip.so``, built from ``csrc/`` with hipcc for gfx950) called through ctypes.
"""
from .bindings import (  # noqa: F401
    context,
    hybridControl,
    hybridMode,
    ipclBigNumber,
    ipclCipherText,
    ipclKeypair,
    ipclPlainText,
    ipclPrivateKey,
    ipclPublicKey,
)
from .fixedpoint import FixedPointNumber  # noqa: F401
from .paillier import (  # noqa: F401
    BNUtils,
    PaillierEncryptedNumber,
    PaillierKeypair,
    PaillierPrivateKey,
    PaillierPublicKey,
)

__all__ = [
    "PaillierKeypair", "PaillierPublicKey", "PaillierPrivateKey", "PaillierEncryptedNumber", "BNUtils",
    "FixedPointNumber", "context", "hybridControl", "hybridMode",
]
__version__ = "0.1.0"
