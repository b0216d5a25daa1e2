"""``magma`` -- import-path alias of ``magma_amd`` so that code written against the
reference package runs unchanged (reference magma/__init__.py:1-20, example_inference.py:1-2):

    from magma import Magma
    from magma.image_input import ImageInput

Every ``magma.<submodule>`` IS the ``magma_amd.<submodule>`` module object (one copy of
each class: ``isinstance`` checks inside the package keep working), not a re-import."""
import importlib
import os
import pkgutil
import sys

import magma_amd as _impl

_self = sys.modules[__name__]
for _m in pkgutil.iter_modules(_impl.__path__):
    if _m.ispkg or _m.name.startswith("_") or not os.path.exists(os.path.join(_impl.__path__[0], _m.name + ".py")):
        continue                      # python sources only (libmagma_hip.so sits in the same directory)
    _mod = importlib.import_module(f"magma_amd.{_m.name}")
    sys.modules[f"{__name__}.{_m.name}"] = _mod
    setattr(_self, _m.name, _mod)

from magma_amd import *  # noqa: E402,F401,F403
from magma_amd import __all__ as _all  # noqa: E402
from magma_amd.datasets import collate_fn  # noqa: E402,F401  (reference magma/__init__.py:20)
from magma_amd.train_loop import eval_step, inference_step, train_step  # noqa: E402,F401  (reference magma/__init__.py:19)

__all__ = list(_all) + ["collate_fn", "eval_step", "inference_step", "train_step"]
