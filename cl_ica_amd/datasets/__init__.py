"""Latent-lookup half of the reference's 3DIdent data pipeline (datasets/threedident_dataset.py) on the GPU."""
from .threedident_dataset import IndexFlatL2, ThreeDIdentLatentPairs

__all__ = ["IndexFlatL2", "ThreeDIdentLatentPairs"]
