"""Overlay of the reference's `source` package: only `source.models.renderer` and
`source.models.frequency_nerf` are replaced (files in this directory tree); every other
submodule -- `source.utils.camera`, `source.models.poses_models.*`, `source.training.*`,
`source.datasets.*` -- still resolves to the reference tree further down `sys.path`, because the
package search path is extended with every other `source/` directory found there."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
