"""Overlay package: the modules of this directory shadow the reference's `pipeline/` modules of the same name, every other
`pipeline.*` import (e.g. `pipeline.train_val_utils`, `pipeline.distributed_utils`: train_SROIE.py:13-21) falls through to the
reference checkout's `pipeline/` directory, which has no `__init__.py` (a namespace portion).  A regular package wins over namespace
portions wherever it sits on sys.path, so the overlay also holds when the reference's own directory comes first
(`python train_SROIE.py` puts the script's directory at sys.path[0]); `extend_path` then appends every other `pipeline/`
directory found on sys.path BEHIND this one.  tests/test_overlay.py pins the resolution."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
