"""jackalope_amd -- MI355X-native HTS read generation behind jackalope's illumina()/pacbio() API.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
include/jackalope_hip.h) and the host-side mirror of the reference's R-level interface.
"""
from ._abi import JackalopeHipError, lib  # noqa: F401
from .bgzf import bgzf_bound, bgzf_deflate  # noqa: F401
from .genome import DeviceGenome, FlatHapSet, HapBuilder, HapSet, RefGenome, create_genome, read_fasta, synthetic_genome  # noqa: F401
from .illumina import illumina, IlluminaSession  # noqa: F401
from .pacbio import pacbio  # noqa: F401
from .profiles import read_profile, Profile  # noqa: F401
from .rng import seed_words  # noqa: F401


def arena_stats(device=-1):
    """{parked bytes, hits, misses} of the library's device arena (large buffers kept from closed sessions for the next one)."""
    import ctypes as C
    b, h, m = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib().jk_device_arena_stats(C.c_int(device), C.byref(b), C.byref(h), C.byref(m))
    return {"bytes": int(b.value), "hits": int(h.value), "misses": int(m.value)}


def arena_trim(device=-1):
    """Hand the arena's parked buffers back to the driver."""
    import ctypes as C
    lib().jk_device_arena_trim(C.c_int(device))

__all__ = ["illumina", "pacbio", "IlluminaSession", "RefGenome", "synthetic_genome", "read_profile", "Profile", "seed_words",
           "JackalopeHipError", "lib", "HapBuilder", "HapSet", "bgzf_bound", "bgzf_deflate", "create_genome", "DeviceGenome", "read_fasta"]
