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
        # (the three-plane forward of the training step -- arg-max, BN sums, whole objects per workgroup -- and the three-plane backward: the defaults)
        'pointnet.hip': ['19pointnet_fwd_kernelILi256ELb1ELb0E', '25pointnet_bwd_fused_kernel',
                         '22pointnet_fwd_p3_kernelILi256ELb1ELb1ELb0ELb1E', '22pointnet_bwd_p3_kernel'],
        # the DEFAULT sweeps (three exact bf16 planes): gradient (one wave per SIMD), sums (two), the stash products
        # (M = 4: all four tables' owner gradients in one launch, the small-product accumulators shared -- 495 of 512 registers)
        'sweep3.hip': ['13sweep3_kernelILi3ELb1ELi4E', '13sweep3_kernelILi3ELb0ELi8E', '13sweep3_kernelILi2ELb1ELi4E', '13sweep3_kernelILi2ELb0ELi8E',
                       '13sweep3_kernelILi4ELb1ELi4ELi4E', '13stash3_kernel'],
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


def test_symmetric_walk_over_ranks_visits_every_ordered_pair_once():
    """ops._sym_jobs: over all ranks of an anchor-sharded job every ORDERED anchor pair (i, j) is produced exactly once -- directly (columns
    left of `mir`: the block's own square), or as the direct / mirrored element of a tile right of it -- the ranks visit (nearly) the same
    number of pairs, and every launch's two stashes stay inside STASH_BYTES."""
    import numpy as np
    from sgaligner_amd import ops
    keep = ops.STASH_BYTES
    try:
        for R, per_rank, stash in ((1, 40, 1 << 16), (2, 12, 1 << 15), (3, 8, 1 << 14), (4, 6, 1 << 30), (5, 4, 1 << 13), (8, 3, 1 << 14), (8, 19, 1 << 17)):
            ops.STASH_BYTES = stash
            nb = R * per_rank                              # anchors in units of 32 rows
            cuts = [32 * per_rank * r for r in range(R + 1)]
            cover = np.zeros((nb, nb), dtype=np.int32)
            work = []
            for rank in range(R):
                w = 0
                for lo, hi, jl, jh, mir in ops._sym_jobs(cuts, rank, 3):
                    assert lo % 32 == 0 and hi % 32 == 0 and jl % 32 == 0 and jh % 32 == 0 and mir % 32 == 0
                    assert cuts[rank] <= lo < hi <= cuts[rank + 1]
                    assert 4 * 3 * ((jh - jl) + max(0, jh - mir)) * (hi - lo) <= stash or hi - lo == 32
                    bl, bh, cl, ch, bm = lo // 32, hi // 32, jl // 32, jh // 32, mir // 32
                    for j in range(cl, ch):
                        cover[bl:bh, j] += 1                       # (i, j) direct
                        if j >= bm:
                            cover[j, bl:bh] += 1                   # (j, i) mirrored
                        else:
                            assert bl <= j < bh                    # ordered columns only inside the block's own square
                        w += (bh - bl)
                work.append(w)
            assert (cover == 1).all(), (R, per_rank, np.argwhere(cover != 1)[:5])
            assert max(work) <= 1.35 * min(work) + 2 * per_rank * per_rank, (R, work)
    finally:
        ops.STASH_BYTES = keep
