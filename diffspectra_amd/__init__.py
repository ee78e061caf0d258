"""diffspectra_amd — MI355X-native DMT + SpecFormer denoising path behind the reference's
model-factory / score_fn / sampler API.  Compute lives in the HIP library under
``csrc/``; importing the package does not require a GPU, running the model does."""
from .config import Config, qm9s_config  # noqa: F401

__all__ = ["Config", "qm9s_config"]
