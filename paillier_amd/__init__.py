"""paillier_amd -- MI355X-native batched Paillier engine (host-side Python mirror of the reference API).

The product is the C-ABI shared library `libpaillier_hip.so` (include/paillier_hip.h), built from
`paillier_amd/csrc/` by `__graft_entry__.build()`.  This package is a thin ctypes binding whose class and
method names follow sachaservan/paillier (PublicKey / SecretKey / EncryptWithR / Decrypt / Add / ConstMult),
with slice-in / slice-out batch variants.  There is no CPU path: every batch call runs HIP kernels, and
loading fails loudly when the library has not been built.
"""
from .api import (  # noqa: F401
    Context,
    Modulus,
    PaillierHipError,
    PublicKey,
    SecretKey,
    ThresholdPublicKey,
    DECRYPT_DEFAULT,
    DECRYPT_NO_CRT,
    LANE_NONUNIT,
    ENC_LEVEL_ONE,
    ENC_LEVEL_TWO,
    MEM_DEVICE,
    MEM_HOST,
    load_library,
    library_path,
)
