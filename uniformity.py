"""Module-level alias so that `from uniformity import torch_uniformity1, ...` (reference sparsify_clip.py:27) keeps working."""
from sparsify_clip_amd.uniformity import (numpy_uniformity, torch_uniformity, torch_uniformity1, torch_uniformity_equivalent,  # noqa: F401
                                          uniformity10)
