"""CPU (cross-compile only): the kernels of the headline path must not spill.  A register spill inside a tile loop is a scratch reload
whose vmcnt wait also stalls the hand-placed LDS DMAs (that is what held the loss sweeps at 0.75 of the MFMA peak), and it
appears or disappears with unrelated source changes -- so it is asserted, not assumed.  tools/kernel_resources.py is the full report."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


def test_headline_kernels_do_not_spill():
    import kernel_resources as kr
    from sgaligner_amd import _build
    if not os.path.exists(_build.HIPCC):
        pytest.skip('hipcc not available')
    hot = {
        'contrastive.hip': ['14sweep16_kernelILi3ELb1ELb0E', '14sweep16_kernelILi3ELb0ELb0E', '14sweep16_kernelILi2ELb1ELb0E',
                            '16sweep16x2_kernelILb1E', '16sweep16x2_kernelILb0E',
                            '25anchor_multi_bwd16_kernelILi3ELb1ELi32ELb1E', '25anchor_multi_bwd16_kernelILi3ELb1ELi32ELb0E',
                            '25anchor_multi_bwd16_kernelILi4ELb1ELi16ELb0E', '19anchor_multi_kernelILi3ELb0E'],
        'pointnet.hip': ['19pointnet_fwd_kernelILi256ELb1ELb0E', '25pointnet_bwd_fused_kernel'],
        'sweepb.hip': ['13sweepb_kernelILi3ELb1E', '13sweepb_kernelILi3ELb0E'],          # opt-in bf16x3 sweeps (39 spills until round 3)
    }
    for base, res in (kr.analyse(os.path.join(_build.CSRC, f)) for f in hot):
        for tag in hot[base]:
            ks = [k for k in res if tag in k]
            assert ks, (base, tag, 'kernel not found (renamed? update this list)')
            for k in ks:
                v = res[k]
                assert v['scratch'] == 0 and v['vspill'] == 0 and v['sspill'] == 0, (k, v)
                assert not v.get('loop_scratch') and not v.get('loop_readlane'), (k, v)


def test_symmetric_walk_blocks_cover_the_anchors_within_the_stash_bound():
    """ops._sym_chunks: contiguous blocks on 32-row boundaries (except the end) whose two stashes -- [A - lo, ns] + [A - hi, ns] floats per
    table -- stay inside STASH_BYTES, growing as the walk moves right."""
    from sgaligner_amd import ops
    keep = ops.STASH_BYTES
    try:
        for A, M, stash in ((155648, 3, 16 << 30), (19456, 3, 1 << 28), (1000, 2, 1 << 20), (31, 3, 1 << 30), (4099, 4, 1 << 24)):
            ops.STASH_BYTES = stash
            ch = ops._sym_chunks(A, M)
            assert ch[0][0] == 0 and ch[-1][1] == A and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
            assert all(lo % 32 == 0 and (hi % 32 == 0 or hi == A) and hi > lo for lo, hi in ch)
            for lo, hi in ch:
                floats = (2 * A - lo - hi) * (hi - lo)
                assert 4 * M * floats <= stash or hi - lo == 32 or hi == A and hi - lo < 32, (A, M, lo, hi)
            heights = [hi - lo for lo, hi in ch[:-1]]
            assert heights == sorted(heights)
    finally:
        ops.STASH_BYTES = keep
