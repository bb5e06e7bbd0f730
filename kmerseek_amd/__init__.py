"""kmerseek_amd — MI355X-native protein k-mer sketch-and-search (the hot path of seanome/kmerseek).

All compute lives in libkmerseek_amd.so (hand-written HIP for gfx950) behind the C ABI of
include/kmerseek_amd.h.  There is no CPU fallback: importing is cheap, but any call needs the
built library and a GPU.
"""
from ._lib import KmerseekLibraryError, SO_PATH  # noqa: F401
from .engine import (Context, Hits, Index, InvalidAminoAcid, KmerseekError, Sketches, SEED,  # noqa: F401
                     make_params, max_hash, moltype_id, pack, validate_and_resolve)

__all__ = ["Context", "Sketches", "Index", "Hits", "KmerseekError", "InvalidAminoAcid", "KmerseekLibraryError",
           "SEED", "make_params", "max_hash", "moltype_id", "pack", "validate_and_resolve", "SO_PATH"]
