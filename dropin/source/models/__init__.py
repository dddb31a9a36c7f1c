"""Overlay of `source.models`: `renderer` / `frequency_nerf` come from this directory, the rest
(`poses_models`, `flow_net`) from the reference's `source/models/` (see ../__init__.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
