"""prmf_amd: MI355X-native implementation of PRMF's multiplicative-update inner loop.

Hand-written HIP kernels (gfx950) behind a C-ABI (include/prmf_hip.h), driven by
Python host code that mirrors the reference's solver functions and CLI
(script/prmf_runner.py of gitter-lab/prmf).  No CPU fallback.
"""
from .pathways import PathwayLibrary  # noqa: F401

__all__ = ["PathwayLibrary", "PrmfEngine", "nmf_manifold_vec_update", "nmf_manifold_vec_update_tradeoff",
           "nmf_pathway"]


def __getattr__(name):
    if name in ("PrmfEngine", "nmf_manifold_vec_update", "nmf_manifold_vec_update_tradeoff", "nmf_pathway"):
        from . import solver
        return getattr(solver, name)
    raise AttributeError(name)
