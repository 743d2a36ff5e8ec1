"""Dataset/collate restatement against vectors produced by the reference's Scan3RDataset on the
same synthetic on-disk dataset (oracle/make_golden.py gen_dataset; the writer is deterministic in its seed).  Every
key of the collated data_dict: same dtype, same shape, same values, bit for bit."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope='module')
def dataset_root(tmp_path_factory):
    from sgaligner_amd.datasets import synthetic_scan3r as S
    root = str(tmp_path_factory.mktemp('scan3r'))
    S.write_dataset(root, n_pairs=6, seed=5)
    return root


@pytest.mark.parametrize('split,kw,tag', [('val', {}, 'val'), ('train', {}, 'train'),
                                          ('val', {'overlap_low': 0.3, 'overlap_high': 0.8}, 'val_overlap')])
def test_collated_batch_equals_reference(dataset_root, split, kw, tag):
    from sgaligner_amd.datasets import Scan3RDataset, synthetic_scan3r as S
    g = load_golden('scan3r_collate_' + tag)
    ds = Scan3RDataset(S.make_cfg(dataset_root, pc_res=64, **kw), split)
    assert len(ds) == int(g['n_items'])
    np.random.seed(123)                       # train: one np.random.rand(1) per item picks the centring scan
    batch = ds.collate_fn([ds[i] for i in range(len(ds))])
    keys = set(batch) | {'n_items'}
    assert keys == set(g.keys()), keys ^ set(g.keys())
    for k, v in batch.items():
        ref = g[k]
        if isinstance(v, torch.Tensor):
            assert str(v.dtype).replace('torch.', '') == str(ref.dtype), (k, v.dtype, ref.dtype)
            v = v.numpy()
        elif k == 'batch_size':
            assert v == int(ref)
            continue
        elif k == 'scene_ids':
            assert np.array_equal(np.asarray(v).astype('U16'), ref)
            continue
        else:
            assert np.asarray(v).dtype == ref.dtype, (k, np.asarray(v).dtype, ref.dtype)
        assert np.asarray(v).shape == ref.shape, (k, np.asarray(v).shape, ref.shape)
        assert np.array_equal(np.asarray(v), ref), k


def test_dataloader_and_hot_path_contract(dataset_root):
    """The collated batch satisfies the hot path's input contract (the same checks synthetic.make_batch obeys)."""
    from sgaligner_amd.datasets import Scan3RDataset, synthetic_scan3r as S
    ds = Scan3RDataset(S.make_cfg(dataset_root, pc_res=32), 'val')
    dl = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=ds.collate_fn)
    seen = 0
    for b in dl:
        T = int(b['tot_obj_count'].sum())
        assert b['tot_obj_pts'].shape == (T, 32, 3) and b['tot_obj_pts'].dtype == torch.float32
        assert b['edges'].shape[0] == int(b['graph_per_edge_count'].sum())
        assert b['graph_per_obj_count'].sum() == T
        for k in ('e1i', 'e2i', 'e1j', 'e2j'):
            assert b[k].dtype == np.int32 and (b[k] >= 0).all() and (b[k] < T).all()
            assert len(b[k]) == int(b[k + '_count'].sum())
        assert len(b['e1i']) == len(b['e2i'])
        # every object is an anchor or a negative of its own scene, exactly once
        assert sorted(np.concatenate([b['e1i'], b['e1j'], b['e2i'], b['e2j']]).tolist()) == list(range(T))
        seen += b['batch_size']
    assert seen == len(ds)


def test_scan_cache_is_bounded_and_value_neutral(dataset_root):
    """The per-process scan cache is an LRU bounded in bytes (every DataLoader worker holds one): whatever the bound -- off, smaller
    than one scan, two scans, unbounded -- the items are identical, and the bytes held never exceed the bound."""
    from sgaligner_amd.datasets import Scan3RDataset, synthetic_scan3r as S
    cfg = S.make_cfg(dataset_root, pc_res=64)
    ref = Scan3RDataset(cfg, 'val', cache=False)
    ref_items = [ref[i] for i in range(len(ref))]
    assert ref.cache_bytes == 0 and len(ref._pkls) == 0
    probe = Scan3RDataset(cfg, 'val')
    probe[0]
    one_scan = max(nb for _, nb in probe._pkls.values())
    for bound in (1, int(2.5 * one_scan), 1 << 30):
        ds = Scan3RDataset(cfg, 'val', cache_bytes=bound)
        for rep in range(2):                                           # second sweep reads through the cache where it holds
            for i in range(len(ds)):
                it = ds[i]
                for k, v in ref_items[i].items():
                    if isinstance(v, torch.Tensor):
                        assert torch.equal(v, it[k]), k
                    elif isinstance(v, np.ndarray):
                        assert np.array_equal(v, it[k]), k
                    else:
                        assert v == it[k], k
                assert ds._pkl_bytes <= bound and ds._pkl_bytes == sum(nb for _, nb in ds._pkls.values())
        assert (len(ds._pkls) == 0) == (bound < one_scan)
        # only the fields __getitem__ reads are kept, and only this dataset's point resolution
        for d, _ in ds._pkls.values():
            assert list(d['obj_points']) == [64]


def test_missing_files_fail_loudly(tmp_path):
    from sgaligner_amd.datasets import Scan3RDataset, synthetic_scan3r as S
    with pytest.raises(FileNotFoundError):
        Scan3RDataset(S.make_cfg(str(tmp_path)), 'val')
