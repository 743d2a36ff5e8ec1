"""Row S: the per-pair E E^T similarity blocks on the matrix cores with rank / top-K fused behind them, and the
per-pair Hits@K / MRR / SGAR aggregation on the device -- against the oracle's restatement of
src/inference/sgaligner/inference_align_reg.py:125-143 + utils/alignment.py:3-70 (computed in fp64 on the host so that only
exact ties could differ).  Embedding widths 100 (one modality), 300 / 400 (joint tables) and 1024 (BASELINE.json configs[4],
both operands streamed); pairs of up to 512 objects; the fp16-input MFMA mode of configs[4] at its relaxed tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, N, D, seed, ragged=True):
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(B, N, 1, seed=seed, ragged=ragged, anchors='val')
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(T, D, generator=g, dtype=torch.float64)
    # matched objects are noisy copies (like trained embeddings), so ranks are non-trivial but mostly small
    e1i, e2i = np.asarray(dd['e1i']), np.asarray(dd['e2i'])
    emb[e2i] = emb[e1i] + 0.8 * torch.randn(len(e1i), D, generator=g, dtype=torch.float64)
    emb *= (0.5 + torch.rand(T, 1, generator=g, dtype=torch.float64))          # un-normalised rows: the kernel normalises
    return dd, emb


@pytest.mark.parametrize('B,N,D', [(5, 20, 100), (4, 64, 300), (3, 33, 400), (2, 256, 300), (3, 40, 1024), (2, 130, 104), (6, 9, 7)])
def test_metrics_equal_oracle(B, N, D):
    from oracle import sga_oracle as O
    from sgaligner_amd.utils import alignment
    dd, emb = _setup(B, N, D, seed=B * 100 + N + D)
    ref = O.evaluate_batch(emb, dd)
    got = alignment.evaluate_batch(emb.float().cuda(), dd, reg_k=3)
    assert [got[k]['correct'] for k in (1, 2, 3, 4, 5)] == [ref['hits'][k][0] for k in (1, 2, 3, 4, 5)]
    assert [got[k]['total'] for k in (1, 2, 3, 4, 5)] == [ref['hits'][k][1] for k in (1, 2, 3, 4, 5)]
    assert np.allclose(got['mrr'], ref['mrr'])
    for mode in ('2', '50', '100'):
        assert got['sgar'][mode] == ref['sgar'][mode], mode
    # node correspondences (alignment.py:59-70) per pair
    o = 0
    for b in range(B):
        n = int(dd['tot_obj_count'][b])
        ns = int(dd['graph_per_obj_count'][b][0])
        sim, _ = O.pair_similarity(emb[o:o + n])
        assert got['node_corrs'][b] == O.node_corrs(sim, ns, 3), b
        o += n


def test_topk_distances_and_ranks_raw():
    """The raw kernel outputs: every object of every pair as a query, K = 5 nearest others with their distances."""
    from sgaligner_amd import ops
    dd, emb = _setup(3, 50, 300, seed=5)
    counts = np.asarray(dd['tot_obj_count'])
    T = int(counts.sum())
    qi = np.arange(T, dtype=np.int32)
    tgt = np.concatenate([np.roll(np.arange(o, o + n), 1) for o, n in zip(np.concatenate([[0], np.cumsum(counts)[:-1]]), counts)]).astype(np.int32)
    rank, tki, tks, _ = ops.simrank(emb.float().cuda(), counts, qi, tgt, 5)
    rank, tki, tks = rank.cpu().numpy(), tki.cpu().numpy(), tks.cpu().numpy()
    o = 0
    for n in counts:
        e = emb[o:o + n]
        e = e / e.norm(dim=1)[:, None]
        sim = (1 - e @ e.t()).numpy()
        for i in range(n):
            order = [j for j in np.argsort(sim[i], kind='stable') if j != i]
            assert tki[o + i].tolist() == order[:5], (o, i)
            assert np.abs(tks[o + i] - sim[i][order[:5]]).max() < 2e-6
            assert rank[o + i] == order.index(int(tgt[o + i] - o)) + 1
        o += n


@pytest.mark.parametrize('D', [300, 1024])
def test_f16_similarity_mode(D):
    """configs[4]: fp16-input MFMA similarity.  Distances within 1e-2 of fp64 (observed ~1e-3); Hits@K identical wherever the
    decisive distance gap exceeds the tolerance."""
    from sgaligner_amd import ops
    dd, emb = _setup(3, 100, D, seed=D)
    counts = np.asarray(dd['tot_obj_count'])
    e1i, e2i = np.asarray(dd['e1i']), np.asarray(dd['e2i'])
    r32, k32, s32, _ = ops.simrank(emb.float().cuda(), counts, e1i, e2i, 3, f16=False)
    r16, k16, s16, _ = ops.simrank(emb.float().cuda(), counts, e1i, e2i, 3, f16=True)
    assert (s32 - s16).abs().max().item() < 1e-2
    assert (s32 - s16).abs().max().item() > 0                                   # it really is the other arithmetic
    gap = (s32[:, 1] - s32[:, 0]).cpu().numpy()                                 # top-1 vs top-2 distance gap
    same = (k32[:, 0] == k16[:, 0]).cpu().numpy()
    assert same[gap > 2e-2].all()
    assert (r32 == r16).float().mean().item() > 0.95


def test_queries_only_in_some_row_blocks_and_empty_batch():
    from sgaligner_amd import ops
    dd, emb = _setup(2, 150, 100, seed=9, ragged=False)
    counts = np.asarray(dd['tot_obj_count'])
    qi = np.asarray([3, 299, 300 + 17, 300 + 280], dtype=np.int32)             # 4 queries in 4 different 16-row blocks
    tgt = np.asarray([200, 5, 300 + 100, 300 + 2], dtype=np.int32)
    rank, tki, _, _ = ops.simrank(emb.float().cuda(), counts, qi, tgt, 1)
    e = emb / emb.norm(dim=1)[:, None]
    for q, (a, t) in enumerate(zip(qi, tgt)):
        o = 0 if a < 300 else 300
        sim = (1 - e[o:o + 300] @ e[a]).numpy()
        order = [j for j in np.argsort(sim, kind='stable') if j != a - o]
        assert int(rank[q]) == order.index(int(t - o)) + 1 and int(tki[q, 0]) == order[0]
    r, k, s, _ = ops.simrank(emb.float().cuda(), counts, np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), 2)
    assert r.numel() == 0 and k.shape == (0, 2)
