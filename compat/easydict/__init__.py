"""`easydict` import shim: `from easydict import EasyDict as edict` (reference
renderer.py:20) resolves here when `compat/` is on PYTHONPATH and the real
package is absent."""
from sparf_amd.edict import EasyDict  # noqa: F401
