"""Drop-in for the reference's source/models/renderer.py: put `dropin/` (and the repo root, plus `compat/` for easydict)
ahead of the reference tree on PYTHONPATH, or replace the reference file with this one.
Trainers do `from source.models.renderer import Graph` and subclass it
(joint_pose_nerf_trainer.py:710-749); nothing else changes."""
from sparf_amd.renderer import Graph  # noqa: F401
from sparf_amd.frequency_nerf import FrequencyEmbedder, NeRF  # noqa: F401
