"""Drop-in for the reference's source/models/frequency_nerf.py (see renderer.py here)."""
from sparf_amd.frequency_nerf import FrequencyEmbedder, NeRF  # noqa: F401
