"""Import-name aliasing for the reference's layout.

The reference resolves its hot path as TOP-LEVEL packages -- `from aligner.sg_aligner import *`, `from aligner.losses
import *` with `src/` on sys.path (src/trainers/trainval_sgaligner.py:6-12), `from utils import alignment` with the
repository root on it (src/inference/sgaligner/inference_align_reg.py:14-19).  A maintainer switches to this
implementation by putting `<root>/sgaligner_amd` ahead of those directories.  The packages in here
are then imported under the reference's names (`aligner`, `utils`, `datasets`); their `__init__` calls `alias()`,
which imports the canonical `sgaligner_amd.<pkg>` package instead and registers it -- and the listed submodules --
under the top-level names, so that

  * there is ONE module object per file (`aligner.sg_aligner is sgaligner_amd.aligner.sg_aligner`), whichever
    name it is imported by, and the relative imports inside the package keep working;
  * other portions of the same (namespace) package later on sys.path stay importable: `aligner.eva`,
    `utils.torch_util`, `datasets.loaders` ... still come from the reference tree.
"""
import importlib
import os
import pkgutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))     # the directory that holds sgaligner_amd/


def alias(top: str, submodules, ours_first: bool = True):
    """Make `top` (and `top.<sub>` for every listed submodule) names of the canonical sgaligner_amd.<top> modules."""
    if ROOT not in sys.path:
        sys.path.append(ROOT)
    real = importlib.import_module('sgaligner_amd.' + top)
    for sub in submodules:
        sys.modules[f'{top}.{sub}'] = importlib.import_module(f'sgaligner_amd.{top}.{sub}')
    own = list(real.__path__)
    own_real = {os.path.realpath(p) for p in own}
    others = [p for p in pkgutil.extend_path([], top) if os.path.realpath(p) not in own_real]
    real.__path__ = own + others if ours_first else others + own
    sys.modules[top] = real          # the import system returns sys.modules[top] once the importing __init__ finishes
    return real
