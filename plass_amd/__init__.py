"""plass_amd — MI355X-native hot path of Plass/PenguiN (kmermatcher -> rescorediagonal -> assembleresults).

Thin ctypes binding of the plasship C-ABI (include/plasship.h).  The Python names mirror the reference's
module names and flag names (mm/commons/Parameters.cpp:423-439,872-892; src/commons/LocalParameters.h:96-102)
so the parity tests read like invocations of the reference modules.  There is no CPU fallback: without the
in-tree HIP library (plass_amd/libplasship.so) or without a GPU every entry point raises.
"""
from ._lib import (  # noqa: F401
    PlasshipError, Context, SeqDB, Candidates, Alignments,
    KmermatchParams, RescoreParams, AssembleParams, OrfParams, OrfHeaders, SynthParams, lib_path, load_library,
)

__all__ = ["PlasshipError", "Context", "SeqDB", "Candidates", "Alignments", "KmermatchParams",
           "RescoreParams", "AssembleParams", "OrfParams", "OrfHeaders", "SynthParams", "lib_path", "load_library"]
