"""MI355X-native BN254 proving hot path (G1/G2 Pippenger MSM + Fr NTT + Groth16 prove) behind the gnark-crypto
call sites that lambdaclass/noir_backend_using_gnark reaches through groth16.Prove / plonk.Prove.

Layout
  csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/zkmi.h) -> libzkmi.so
  _lib.py      ctypes loader (fails loudly when the HIP library or a GPU is missing -- there is no CPU fallback)
  bn254.py     host-side mirror of the gnark-crypto interface for this path (MultiExp, fft.Domain)
  groth16.py   host-side mirror of gnark's groth16 prove for this path (ProvingKey, prove, compute_h)
  parallel.py  one-process-per-GPU sharding of MSMs / proofs with torch.distributed (RCCL)
"""
from . import _lib  # noqa: F401
from .bn254 import (DIF, DIT, Domain, MultiExpConfig, bit_reverse, g1_multi_exp, g2_multi_exp)  # noqa: F401
from .groth16 import R1CS, ProvingKey, compute_h, prove, prove_r1cs, setup  # noqa: F401
from .wire import deserialize_felts, serialize_felts  # noqa: F401

__all__ = ["DIF", "DIT", "Domain", "MultiExpConfig", "bit_reverse", "g1_multi_exp", "g2_multi_exp", "ProvingKey", "compute_h", "prove"]
