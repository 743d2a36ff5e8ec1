"""Drop-in for the reference's `datasets.scan3r` (src/datasets/scan3r.py).  As top-level `datasets` (reference layout,
`<root>/sgaligner_amd` on sys.path) only `datasets.scan3r` is taken over; `datasets.loaders` etc. keep resolving to
the reference tree (sgaligner_amd/_dropin.py)."""
if __name__ == 'datasets':
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from sgaligner_amd._dropin import alias as _alias
    _alias('datasets', ['scan3r'], ours_first=False)
else:
    from .scan3r import Scan3RDataset, DeviceBatch, DevicePrefetcher  # noqa: F401
