"""Drop-in for the reference's `aligner` package (src/aligner/): sg_aligner, losses, networks.{base,gat,pct,pointnet}.
Importable as `sgaligner_amd.aligner` or -- with `<root>/sgaligner_amd` on sys.path, the reference's own layout
(src/trainers/trainval_sgaligner.py:6-12) -- as top-level `aligner`; both names give the same module objects."""
if __name__ == 'aligner':                     # imported the reference's way: become an alias of the canonical package
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from sgaligner_amd._dropin import alias as _alias
    _alias('aligner', ['networks', 'networks.base', 'networks.gat', 'networks.pct', 'networks.pointnet', 'sg_aligner', 'losses'])
