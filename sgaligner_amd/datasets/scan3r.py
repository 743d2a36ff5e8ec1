"""Scan3R sub-scan pair dataset and collate: the data format in front of the hot path.

Drop-in for the reference's `Scan3RDataset` (src/datasets/scan3r.py:11-209): same constructor (`cfg`, `split`), same
on-disk layout, same per-item and collated `data_dict` keys, dtypes and values (tests compare bit for bit against
vectors produced by the reference class on the same files).

On-disk layout read here (written by the reference's preprocessing, preprocessing/scan3r/preprocess.py:195-211, and by
`synthetic_scan3r.write_dataset` for tests):
    <scans_dir>/scans/<scan_id>/data.npy             structured vertex array, fields x,y,z,... (utils/scan3r.py:143-144)
    <scans_dir>/files/<mode>/data/<scan_id>.pkl      per-scan graph: objects_id, objects_cat, edges [E,2], obj_points
                                                     {resolution: [N,res,3]}, object_id2idx, rel_trans [N,3],
                                                     bow_vec_object_edge_feats [N,41], bow_vec_object_attr_feats [N,164]
    <scans_dir>/files/<mode>/anchors<type>_<split>.json   list of {src, ref, overlap, anchorIds}

Beyond the reference: `DeviceBatch` moves a collated batch to the GPU once (tensors -> device, the four numpy index
sets stay on the host exactly as `utils/torch_util.to_cuda` leaves them, plus ready-made int32 device copies), so the
training step does no per-iteration host->device index traffic.
"""
from __future__ import annotations

import json
import os
import os.path as osp
import pickle
from collections import OrderedDict

import numpy as np
import torch
import torch.utils.data as data


def _load_points(npy_path):
    """All vertices of a scan as [n,3] (utils/scan3r.py:98-100: stack of the x,y,z fields)."""
    v = np.load(npy_path)
    return np.stack([v['x'], v['y'], v['z']]).transpose((1, 0))


def _load_pkl(path):
    with open(path, 'rb') as fh:
        return pickle.load(fh)


class Scan3RDataset(data.Dataset):
    def __init__(self, cfg, split, cache=True, cache_bytes=None):
        """cache (beyond the reference, which re-reads both files of both scans for every item): keep each scan's centroid (all that
        is used of data.npy, 24 bytes) and the fields of its graph dict that `__getitem__` reads -- the obj_points array of THIS
        dataset's resolution only, not every preprocessed resolution -- after the first read: same values, same np.random draws;
        17 of the 43 ms per iteration of the end-to-end loop at 16 pairs were these reads.
        The graph cache is an LRU bounded by `cache_bytes` (default 1 GiB, env SGA_DATASET_CACHE_MB; 0 disables it): every
        DataLoader worker process holds its own copy (and `persistent_workers` keeps it for the whole run), so the bound is per
        worker -- budget num_workers x cache_bytes of host memory per loader."""
        self.split = split
        self.cache = bool(cache)
        if cache_bytes is None:
            cache_bytes = int(float(os.environ.get('SGA_DATASET_CACHE_MB', '1024')) * (1 << 20))
        self.cache_bytes = int(cache_bytes) if self.cache else 0
        self._centroids, self._pkls, self._pkl_bytes = {}, OrderedDict(), 0
        self.pc_resolution = cfg.val.pc_res if split == 'val' else cfg.train.pc_res
        self.anchor_type_name = cfg.preprocess.anchor_type_name
        self.model_name = cfg.model_name
        self.scan_type = cfg.scan_type
        self.data_root_dir = cfg.data.root_dir
        self.scans_dir = cfg.data.root_dir if self.scan_type == 'scan' else cfg.data.subscan_dir
        self.scans_scenes_dir = osp.join(self.scans_dir, 'scans')
        self.scans_files_dir = osp.join(self.scans_dir, 'files')
        self.mode = 'orig' if self.split == 'train' else cfg.val.data_mode
        self.anchor_data_filename = osp.join(self.scans_files_dir,
                                             '{}/anchors{}_{}.json'.format(self.mode, self.anchor_type_name, split))
        with open(self.anchor_data_filename) as fh:
            self.anchor_data = json.load(fh)
        if split == 'val' and cfg.val.overlap_low != cfg.val.overlap_high:      # scan3r.py:35-41
            self.anchor_data = [a for a in self.anchor_data
                                if cfg.val.overlap_low <= a['overlap'] < cfg.val.overlap_high]
        self.is_training = self.split == 'train'
        self.do_augmentation = False if self.split == 'val' else cfg.train.use_augmentation
        self.rot_factor = cfg.train.rot_factor
        self.augment_noise = cfg.train.augmentation_noise

    def __len__(self):
        return len(self.anchor_data)

    def __getitem__(self, idx):
        pair = self.anchor_data[idx]
        src_id, ref_id = pair['src'], pair['ref']
        overlap = pair['overlap'] if 'overlap' in pair else -1.0

        # centring (scan3r.py:66-77): train draws which scan's centroid to use -- one np.random.rand(1) per item, after
        # BOTH point clouds were loaded (the RNG stream position is part of the contract)
        src_c, ref_c = self._centroid(src_id), self._centroid(ref_id)
        if self.split == 'train':
            center = src_c if np.random.rand(1)[0] > 0.5 else ref_c
        else:
            center = src_c

        src, ref = self._graph(src_id), self._graph(ref_id)
        src_ids, ref_ids = src['objects_id'], ref['objects_id']

        # anchors (scan3r.py:85-93): listed anchors that are non-zero and present on both sides, in list order;
        # at train time only the first max(2*, int(0.3 n)) of them   (* 2 when int(0.3 n) < 1)
        anchors = pair['anchorIds'] if 'anchorIds' in pair else src_ids
        in_src, in_ref = set(np.asarray(src_ids).tolist()), set(np.asarray(ref_ids).tolist())
        anchors = [a for a in anchors if a != 0 and a in in_src and a in in_ref]
        if self.split == 'train':
            keep = int(0.3 * len(anchors))
            anchors = anchors[:2 if keep < 1 else keep]
        anchor_set = set(anchors)

        src_obj_pts = src['obj_points'][self.pc_resolution] - center
        ref_obj_pts = ref['obj_points'][self.pc_resolution] - center
        n_src = src_obj_pts.shape[0]
        s_map, r_map = src['object_id2idx'], ref['object_id2idx']
        e1i = np.array([s_map[a] for a in anchors])
        e1j = np.array([s_map[o] for o in src_ids if o not in anchor_set])
        e2i = np.array([r_map[a] for a in anchors]) + n_src
        e2j = np.array([r_map[o] for o in ref_ids if o not in anchor_set]) + n_src

        cat = lambda k: torch.cat([torch.from_numpy(src[k]), torch.from_numpy(ref[k])])
        tot_pts = torch.cat([torch.from_numpy(src_obj_pts), torch.from_numpy(ref_obj_pts)]).type(torch.FloatTensor)
        return {
            'obj_ids': np.concatenate([src_ids, ref_ids]),
            'tot_obj_pts': tot_pts,
            'graph_per_obj_count': np.array([n_src, ref_obj_pts.shape[0]]),
            'graph_per_edge_count': np.array([src['edges'].shape[0], ref['edges'].shape[0]]),
            'e1i': e1i, 'e1i_count': e1i.shape[0], 'e2i': e2i, 'e2i_count': e2i.shape[0],
            'e1j': e1j, 'e1j_count': e1j.shape[0], 'e2j': e2j, 'e2j_count': e2j.shape[0],
            'tot_obj_count': tot_pts.shape[0],
            'tot_bow_vec_object_attr_feats': cat('bow_vec_object_attr_feats'),
            'tot_bow_vec_object_edge_feats': cat('bow_vec_object_edge_feats'),
            'tot_rel_pose': cat('rel_trans'),
            'edges': cat('edges'),
            'global_obj_ids': np.concatenate((src['objects_cat'], ref['objects_cat'])),
            'scene_ids': [src_id, ref_id],
            'pcl_center': center,
            'overlap': overlap,
        }

    def _centroid(self, scan_id):
        """np.mean over all vertices of the scan (scan3r.py:66-77 uses nothing else of data.npy)."""
        c = self._centroids.get(scan_id) if self.cache else None
        if c is None:
            c = np.mean(_load_points(osp.join(self.scans_scenes_dir, '{}/data.npy'.format(scan_id))), axis=0)
            if self.cache:
                self._centroids[scan_id] = c
        return c

    _GRAPH_KEYS = ('objects_id', 'objects_cat', 'edges', 'object_id2idx', 'rel_trans', 'bow_vec_object_edge_feats',
                   'bow_vec_object_attr_feats')

    def _graph(self, scan_id):
        if self.cache_bytes <= 0:
            return _load_pkl(osp.join(self.scans_files_dir, '{}/data/{}.pkl'.format(self.mode, scan_id)))
        d = self._pkls.get(scan_id)
        if d is not None:
            self._pkls.move_to_end(scan_id)
            return d[0]                                               # (fields, bytes)
        full = _load_pkl(osp.join(self.scans_files_dir, '{}/data/{}.pkl'.format(self.mode, scan_id)))
        d = {k: full[k] for k in self._GRAPH_KEYS if k in full}
        d['obj_points'] = {self.pc_resolution: full['obj_points'][self.pc_resolution]}
        nbytes = sum(v.nbytes for v in d.values() if isinstance(v, np.ndarray)) + d['obj_points'][self.pc_resolution].nbytes \
            + 64 * len(d.get('object_id2idx', ()))
        if nbytes <= self.cache_bytes:
            self._pkls[scan_id] = (d, nbytes)
            self._pkl_bytes += nbytes
            while self._pkl_bytes > self.cache_bytes:                 # least recently used scans leave first
                _, (_, nb) = self._pkls.popitem(last=False)
                self._pkl_bytes -= nb
        return d

    @staticmethod
    def _collate_entity_idxs(batch):
        """Batch-global index sets (scan3r.py:142-173): every pair's four sets shifted by the objects before it."""
        shift = np.concatenate([[0], np.cumsum([b['tot_obj_count'] for b in batch])[:-1]]).astype(np.int64)
        out = []
        for key in ('e1i', 'e2i', 'e1j', 'e2j'):
            parts = [np.asarray(b[key]) + s for b, s in zip(batch, shift)]
            out.append(np.concatenate(parts).astype(np.int32))
        return tuple(out)

    def collate_fn(self, batch):
        tcat = lambda k: torch.cat([b[k] for b in batch])
        stack = lambda k: np.stack([b[k] for b in batch])
        d = {'tot_obj_pts': tcat('tot_obj_pts')}
        d['e1i'], d['e2i'], d['e1j'], d['e2j'] = self._collate_entity_idxs(batch)
        for k in ('e1i_count', 'e2i_count', 'e1j_count', 'e2j_count', 'tot_obj_count'):
            d[k] = stack(k)
        d['global_obj_ids'] = np.concatenate([b['global_obj_ids'] for b in batch])
        d['tot_bow_vec_object_attr_feats'] = tcat('tot_bow_vec_object_attr_feats').double()
        d['tot_bow_vec_object_edge_feats'] = tcat('tot_bow_vec_object_edge_feats').double()
        d['tot_rel_pose'] = tcat('tot_rel_pose').double()
        d['graph_per_obj_count'] = stack('graph_per_obj_count')
        d['graph_per_edge_count'] = stack('graph_per_edge_count')
        d['edges'] = tcat('edges')
        d['scene_ids'] = stack('scene_ids')
        d['obj_ids'] = np.concatenate([b['obj_ids'] for b in batch])
        d['pcl_center'] = stack('pcl_center')
        d['overlap'] = stack('overlap')
        d['batch_size'] = d['overlap'].shape[0]
        return d


class DeviceBatch(dict):
    """A collated batch resident on one GPU: tensors moved once (non_blocking from pinned memory when available), numpy
    entries left on the host exactly as the reference's `to_cuda` leaves them (utils/torch_util.py:26-36)."""

    def __init__(self, data_dict, device='cuda', pin=True, staging=None):
        """staging: a dict of reusable pinned host buffers (DevicePrefetcher owns two such sets): `tensor.pin_memory()` registers
        fresh memory for every tensor of every batch (1.1 ms each: 8 of the 43 ms per iteration of the end-to-end loop); copying into
        a pinned buffer that already exists is a memcpy."""
        super().__init__()
        for k, v in data_dict.items():
            if isinstance(v, torch.Tensor):
                if pin and v.device.type == 'cpu' and torch.cuda.is_available() and not v.is_pinned():
                    if staging is not None:
                        buf = staging.get(k)
                        if buf is None or buf.dtype != v.dtype or buf.numel() < v.numel():
                            buf = torch.empty((max(v.numel(), 1) * 5 // 4,), dtype=v.dtype).pin_memory()     # head-room: batches vary in size
                            staging[k] = buf
                        pv = buf[:v.numel()].view(v.shape)
                        np.copyto(pv.numpy(), v.contiguous().numpy())        # plain memcpy (Tensor.copy_ into pinned memory goes through the runtime: ~2 ms)
                        v = pv
                    else:
                        v = v.pin_memory()
                self[k] = v.to(device, non_blocking=True)
            else:
                self[k] = v


class DevicePrefetcher:
    """Iterate a loader of collated host batches ONE BATCH AHEAD on the device: while the step of batch i runs on the
    compute stream, batch i+1 is uploaded on a second HIP stream (pinned host memory, non-blocking copies), so the PCIe
    transfer (403 MB of points per step at BASELINE configs[1]: ~7 ms at Gen5 x16) is off the step's critical path.  The
    reference moves each batch synchronously inside the loop (epoch_based_trainer.py:86, utils/torch_util.py:26-36).
    `prepare` is applied to the host batch first (the per-rank pair shard in the multi-GPU trainer).  On a CPU device it
    degrades to a plain map."""

    def __init__(self, loader, device='cuda', prepare=None):
        self.loader, self.device, self.prepare = loader, torch.device(device), prepare
        self.stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None
        self._staging = [{}, {}]                 # two sets of reusable pinned buffers, used alternately
        self._staging_ev = [None, None]          # the upload event that last read each set
        self._n = 0

    def __len__(self):
        return len(self.loader)

    def _upload(self, data_dict):
        if data_dict is None:
            return None
        if self.prepare is not None:
            data_dict = self.prepare(data_dict)
        if self.stream is None:
            return DeviceBatch(data_dict, self.device, pin=False), None
        slot = self._n & 1
        self._n += 1
        if self._staging_ev[slot] is not None:
            self._staging_ev[slot].synchronize()             # the copy that read this staging set two batches ago (long done)
        with torch.cuda.stream(self.stream):
            batch = DeviceBatch(data_dict, self.device, staging=self._staging[slot])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._staging_ev[slot] = ev
        return batch, ev

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._upload(next(it, None))
        while nxt is not None:
            batch, ev = nxt
            nxt = self._upload(next(it, None))          # issued before batch i is handed out: overlaps with its step
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for v in batch.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(cur)            # allocated on the copy stream, consumed on the compute stream
            yield batch
