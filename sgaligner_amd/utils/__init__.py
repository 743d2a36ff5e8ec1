"""Drop-in for the reference's `utils.alignment` (utils/alignment.py) + the FPS step of `utils.point_cloud`.
With `<root>/sgaligner_amd` on sys.path this package is found as top-level `utils`; only `utils.alignment` is taken
over then -- every other `utils.*` module (torch_util, common, scan3r, the full point_cloud ...) keeps resolving to
the reference tree further down sys.path (see sgaligner_amd/_dropin.py)."""
if __name__ == 'utils':
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from sgaligner_amd._dropin import alias as _alias
    _alias('utils', ['alignment'], ours_first=False)
