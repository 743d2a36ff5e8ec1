"""Seeded synthetic subscan-pair batches with the reference's `data_dict` contract.

Schema follows Scan3RDataset.collate_fn (reference src/datasets/scan3r.py:179-209) and the index-set
construction of __getitem__/_collate_entity_idxs (:102-107, :142-173): per pair the src objects come
first, then the ref objects; `e1i/e2i` are anchor (matching) objects, `e1j/e2j` all other objects of the
respective scene; e2* are offset by N_src and everything by the running object count of the batch.
The generator itself is SURVEY.md 8(d)'s: half of each scene's objects are shared shapes.
"""
from __future__ import annotations

import numpy as np
import torch


def make_batch(n_pairs: int, n_obj, n_pts: int, seed: int = 42, device='cpu', rel_dim: int = 41,
               attr_dim: int = 164, ragged: bool = False, anchors: str = 'train', gen_device=None) -> dict:
    """n_obj: int N (both scenes) or (N_src, N_ref).  ragged=True varies the per-pair object counts.
    anchors='train' -> A = int(0.3*N) per pair (min 2, capped at the common objects); 'val' -> all common objects."""
    gen_device = gen_device or 'cpu'
    g = torch.Generator(device=gen_device).manual_seed(seed)
    rng = np.random.default_rng(seed)
    ns0, nr0 = (n_obj, n_obj) if isinstance(n_obj, int) else n_obj
    counts = []
    for b in range(n_pairs):
        if ragged:
            counts.append((max(3, ns0 - int(rng.integers(0, 3))), max(3, nr0 - int(rng.integers(0, 3)))))
        else:
            counts.append((ns0, nr0))
    T = sum(a + b for a, b in counts)

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=gen_device, dtype=torch.float32)

    def rand(*shape):
        return torch.rand(*shape, generator=g, device=gen_device, dtype=torch.float32)

    pts = torch.empty((T, n_pts, 3), dtype=torch.float32, device=gen_device)
    pose = torch.empty((T, 3), dtype=torch.float64, device=gen_device)
    e1i, e2i, e1j, e2j = [], [], [], []
    e1c, e2c, j1c, j2c = [], [], [], []
    edges, ecounts = [], []
    off = 0
    for (ns, nr) in counts:
        ncom = min(ns, nr) // 2
        c_src = rand(ns, 3) * 6 - 3
        sc_src = 0.3 * (0.5 + rand(ns, 3))
        c_ref = rand(nr, 3) * 6 - 3
        sc_ref = 0.3 * (0.5 + rand(nr, 3))
        c_ref[:ncom] = c_src[:ncom]
        sc_ref[:ncom] = sc_src[:ncom]
        p_src = c_src[:, None, :] + sc_src[:, None, :] * randn(ns, n_pts, 3)
        p_ref = c_ref[:, None, :] + sc_ref[:, None, :] * randn(nr, n_pts, 3)
        p_ref[:ncom] += 0.01 * randn(ncom, n_pts, 3)
        center = p_src.reshape(-1, 3).mean(0)
        pts[off:off + ns] = p_src - center
        pts[off + ns:off + ns + nr] = p_ref - center
        pose[off:off + ns] = (c_src[0:1] - c_src).double()
        pose[off + ns:off + ns + nr] = (c_ref[0:1] - c_ref).double()
        # A = int(0.3*N) anchors per pair (configs[1]: 19, configs[2]: 38), the first A of the
        # common objects (train-time truncation of scan3r.py:89-91, min 2); 'val' = every common object
        a = ncom if anchors == 'val' else max(2, int(0.3 * min(ns, nr)))
        a = min(a, ncom)
        e1i += list(range(off, off + a))
        e2i += list(range(off + ns, off + ns + a))
        e1j += list(range(off + a, off + ns))
        e2j += list(range(off + ns + a, off + ns + nr))
        e1c.append(a); e2c.append(a); j1c.append(ns - a); j2c.append(nr - a)
        for n in (ns, nr):
            ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
            m = ii != jj
            edges.append(np.stack([ii[m], jj[m]], 1))
            ecounts.append(int(m.sum()))
        off += ns + nr
    # bag-of-words features: rel rows sum to N-1 (mostly the 'none' column 0), attr ~ Bernoulli(0.03)
    rel = torch.zeros((T, rel_dim), dtype=torch.float64, device=gen_device)
    o = 0
    for (ns, nr) in counts:
        for n in (ns, nr):
            k = torch.floor(rand(n, 3) * (rel_dim - 1)).long() + 1
            r = torch.zeros((n, rel_dim), dtype=torch.float64, device=gen_device)
            r.scatter_add_(1, k, torch.ones((n, 3), dtype=torch.float64, device=gen_device))
            r[:, 0] = (n - 1) - r[:, 1:].sum(1)
            rel[o:o + n] = r.clamp_min(0)
            o += n
    attr = (rand(T, attr_dim) < 0.03).double()
    f = lambda x: np.asarray(x, dtype=np.int32)
    dd = {
        'tot_obj_pts': pts.to(device),
        'tot_bow_vec_object_attr_feats': attr.to(device),
        'tot_bow_vec_object_edge_feats': rel.to(device),
        'tot_rel_pose': pose.to(device),
        'edges': torch.from_numpy(np.concatenate(edges).astype(np.int64)).to(device),
        'e1i': f(e1i), 'e2i': f(e2i), 'e1j': f(e1j), 'e2j': f(e2j),
        'e1i_count': np.asarray(e1c), 'e2i_count': np.asarray(e2c),
        'e1j_count': np.asarray(j1c), 'e2j_count': np.asarray(j2c),
        'tot_obj_count': np.asarray([a + b for a, b in counts]),
        'graph_per_obj_count': np.asarray(counts),
        'graph_per_edge_count': np.asarray(ecounts).reshape(-1, 2),
        'batch_size': n_pairs,
    }
    return dd


def to_device(data_dict: dict, device) -> dict:
    """torch tensors move, numpy index arrays stay on host (reference utils/torch_util.py:26-36)."""
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in data_dict.items()}


def make_batch_fast(n_pairs: int, n_obj: int, n_pts: int, seed: int = 42, device='cuda', rel_dim: int = 41,
                    attr_dim: int = 164, anchors: str = 'train', pair_chunk: int = 256) -> dict:
    """Same distribution and data_dict schema as make_batch for UNIFORM batches (every scene has n_obj objects), generated
    with batched device ops -- configs[2] (4096 pairs x 128 objects x 512 points: 1 048 576 objects, 133 M edges) takes
    seconds instead of minutes.  The random stream differs from make_batch's (which draws pair by pair), so the two
    generators give different -- equally distributed -- batches for the same seed."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    N, P, B = int(n_obj), int(n_pts), int(n_pairs)
    T = 2 * B * N
    ncom = N // 2
    a = ncom if anchors == 'val' else min(max(2, int(0.3 * N)), ncom)

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=dev, dtype=torch.float32)

    def rand(*shape):
        return torch.rand(*shape, generator=g, device=dev, dtype=torch.float32)

    pts = torch.empty((T, P, 3), dtype=torch.float32, device=dev)
    pose = torch.empty((T, 3), dtype=torch.float64, device=dev)
    rel = torch.empty((T, rel_dim), dtype=torch.float64, device=dev)
    for b0 in range(0, B, pair_chunk):
        nb = min(pair_chunk, B - b0)
        c = rand(nb, 2, N, 3) * 6 - 3                         # [pair, scene, object, xyz]
        sc = 0.3 * (0.5 + rand(nb, 2, N, 3))
        c[:, 1, :ncom] = c[:, 0, :ncom]                      # the first N/2 objects are shared shapes
        sc[:, 1, :ncom] = sc[:, 0, :ncom]
        p = c[:, :, :, None, :] + sc[:, :, :, None, :] * randn(nb, 2, N, P, 3)
        p[:, 1, :ncom] += 0.01 * randn(nb, ncom, P, 3)
        center = p[:, 0].reshape(nb, -1, 3).mean(1)          # both scenes centred by the src mean (scan3r.py:76,96-97)
        p -= center[:, None, None, None, :]
        pts[2 * N * b0:2 * N * (b0 + nb)] = p.reshape(-1, P, 3)
        pose[2 * N * b0:2 * N * (b0 + nb)] = (c[:, :, 0:1] - c).reshape(-1, 3).double()
        k = torch.floor(rand(nb * 2 * N, 3) * (rel_dim - 1)).long() + 1
        r = torch.zeros((nb * 2 * N, rel_dim), dtype=torch.float64, device=dev)
        r.scatter_add_(1, k, torch.ones((nb * 2 * N, 3), dtype=torch.float64, device=dev))
        r[:, 0] = (N - 1) - r[:, 1:].sum(1)
        rel[2 * N * b0:2 * N * (b0 + nb)] = r.clamp_min(0)
        del p, c, sc, r, k
    attr = torch.empty((T, attr_dim), dtype=torch.float64, device=dev)
    for t0 in range(0, T, 1 << 18):
        t1 = min(T, t0 + (1 << 18))
        attr[t0:t1] = (rand(t1 - t0, attr_dim) < 0.03).double()
    # all ordered pairs i != j per graph, graph-local ids (preprocess.py:184-193), the same template for every graph
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing='ij')
    m = ii != jj
    tmpl = torch.from_numpy(np.stack([ii[m], jj[m]], 1).astype(np.int64)).to(dev)
    edges = tmpl.repeat(2 * B, 1)
    off = (2 * N) * np.arange(B, dtype=np.int64)[:, None]
    f = lambda x: np.ascontiguousarray(x.reshape(-1).astype(np.int32))
    dd = {
        'tot_obj_pts': pts, 'tot_bow_vec_object_attr_feats': attr, 'tot_bow_vec_object_edge_feats': rel,
        'tot_rel_pose': pose, 'edges': edges,
        'e1i': f(off + np.arange(a)[None, :]), 'e2i': f(off + N + np.arange(a)[None, :]),
        'e1j': f(off + np.arange(a, N)[None, :]), 'e2j': f(off + N + np.arange(a, N)[None, :]),
        'e1i_count': np.full(B, a), 'e2i_count': np.full(B, a),
        'e1j_count': np.full(B, N - a), 'e2j_count': np.full(B, N - a),
        'tot_obj_count': np.full(B, 2 * N), 'graph_per_obj_count': np.full((B, 2), N),
        'graph_per_edge_count': np.full((B, 2), N * (N - 1)), 'batch_size': B,
    }
    return dd
