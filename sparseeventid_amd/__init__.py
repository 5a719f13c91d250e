"""MI355X-native submanifold sparse-convolution engine for the SparseEventID ResNet hot path.

``sparseeventid_amd.scn`` mirrors the ``sparseconvnet`` module surface the reference uses;
``sparseeventid_amd.networks`` restates the reference topology on top of it.
"""
from . import config, scn  # noqa: F401
from .networks import Encoder, build_networks  # noqa: F401

__version__ = "0.1.0"
