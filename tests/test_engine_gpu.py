"""Epoch loop on a synthetic on-disk dataset: training runs and learns, snapshots have the
reference's layout, resume continues the same trajectory, a reference-style snapshot (no loss layers) loads, and one
optimiser step equals oracle gradients + torch Adam on the CPU."""
import os.path as osp

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(tr):
    return torch.cat([p.detach().flatten().cpu() for p in tr.steps.params])


def test_train_resume_and_snapshot_format(tmp_path):
    from sgaligner_amd.datasets import synthetic_scan3r as S
    from sgaligner_amd.epoch_trainer import Trainer
    root = str(tmp_path / 'data')
    S.write_dataset(root, n_pairs=8, seed=3, resolutions=(64,))
    cfgA = S.make_cfg(root, max_epoch=3, output_dir=str(tmp_path / 'runA'))
    np.random.seed(0)
    a = Trainer(cfgA)
    hist = a.run()
    assert len(hist) == 3 and all(np.isfinite(h['train']['loss']) and np.isfinite(h['val']['loss']) for h in hist)
    assert hist[-1]['train']['loss'] < hist[0]['train']['loss']            # it learns
    snap = torch.load(osp.join(a.snapshot_dir, 'snapshot.pth.tar'), map_location='cpu', weights_only=False)
    assert {'epoch', 'iteration', 'model', 'optimizer'} <= set(snap)        # base_trainer.py:80-101
    ep = torch.load(osp.join(a.snapshot_dir, 'epoch-3.pth.tar'), map_location='cpu', weights_only=False)
    assert ep['epoch'] == 3 and ep['iteration'] == 3 * len(a.train_loader) and 'optimizer' not in ep
    assert set(ep['model']) == set(a.model.state_dict())
    assert osp.exists(osp.join(a.snapshot_dir, 'best_snapshot.pth.tar'))

    # two epochs, then a fresh process-like object resumes for the third: same parameters as the straight run
    cfgB = S.make_cfg(root, max_epoch=2, output_dir=str(tmp_path / 'runB'))
    np.random.seed(0)
    b = Trainer(cfgB)
    b.run()
    cfgB.optim.max_epoch = 3
    b2 = Trainer(cfgB)
    b2.run(resume=True)
    assert b2.epoch == 3 and b2.iteration == a.iteration
    pa, pb = _params(a), _params(b2)
    # Adam turns fp32 summation-order noise on near-zero gradients into +-lr steps, so a handful of parameters may sit a
    # few lr apart; the trajectories agree if the bulk is identical and the distance is tiny in norm
    diff = (pa - pb).abs()
    assert diff.median() < 1e-7 and (diff > 1e-5).float().mean() < 0.02, (diff.median(), (diff > 1e-5).float().mean())
    assert diff.norm() < 2e-3 * pa.norm() and diff.max() < 10 * cfgA.optim.lr, (diff.norm() / pa.norm(), diff.max())

    # a reference-style snapshot: only epoch / iteration / model
    ref_style = {'epoch': 7, 'iteration': 70, 'model': {('module.' + k): v for k, v in a.model.state_dict().items()}}
    path = str(tmp_path / 'ref_style.pth.tar')
    torch.save(ref_style, path)
    c = Trainer(S.make_cfg(root, max_epoch=1, output_dir=str(tmp_path / 'runC')))
    rep = c.load_snapshot(path)
    assert rep == {'missing': [], 'unexpected': []} and c.epoch == 7 and c.iteration == 70
    for k, v in a.model.state_dict().items():
        assert torch.equal(c.model.state_dict()[k].cpu(), v.cpu()), k


def test_one_optimizer_step_equals_oracle_plus_adam(tmp_path):
    from oracle import sga_oracle as O
    from sgaligner_amd.datasets import Scan3RDataset, synthetic_scan3r as S
    from sgaligner_amd.epoch_trainer import Trainer
    root = str(tmp_path / 'data')
    S.write_dataset(root, n_pairs=4, seed=6, resolutions=(64,))
    cfg = S.make_cfg(root, max_epoch=1, batch_size=4, lr=1e-2, output_dir=str(tmp_path / 'run'))
    tr = Trainer(cfg)
    names = [n for n, _ in tr.model.named_parameters()]
    before = {n: p.detach().cpu().clone() for n, p in tr.model.named_parameters()}
    state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items() if 'num_batches' not in k}
    # the single batch of the epoch, exactly as the loader will produce it (drop_last, shuffle seed 42)
    ds = Scan3RDataset(cfg, 'train')
    # the epoch's shuffle: generator seeded with seed + epoch; the DataLoader iterator draws its base seed first
    g = torch.Generator().manual_seed(42 + 1)
    torch.empty((), dtype=torch.int64).random_(generator=g)
    order = torch.randperm(len(ds), generator=g).tolist()
    np.random.seed(5)
    dd = ds.collate_fn([ds[i] for i in order])
    _, loss_o, grads_o = O.train_step(state, dd, cfg.modules)
    cpu_params = [before[n].clone().requires_grad_(True) for n in names]
    opt = torch.optim.Adam(cpu_params, lr=1e-2)
    for p, n in zip(cpu_params, names):
        p.grad = grads_o[n].float() if n in grads_o else torch.zeros_like(p)
    opt.step()
    np.random.seed(5)
    tr.set_train_mode()
    tr.epoch = 1
    summary = tr.train_epoch()
    lo = float(loss_o['loss'].detach())
    assert abs(summary['loss'] - lo) < 1e-3 * max(1.0, abs(lo))
    for p_ref, (n, p) in zip(cpu_params, tr.model.named_parameters()):
        if n not in grads_o:
            continue
        # Adam normalises the step to ~lr: compare the step, not just the value
        step_ref, step = p_ref.detach() - before[n], p.detach().cpu() - before[n]
        big = grads_o[n].abs() > 1e-4 * grads_o[n].abs().max()
        assert (step - step_ref)[big].abs().max() < 2e-2 * 1e-2 + 1e-6, n


def test_ground_truth_config_modules_train_end_to_end(tmp_path):
    """configs/scan3r/scan3r_ground_truth.yaml:5 -- modules ['pct', 'gat', 'rel', 'attr']: dataset -> epoch loop -> PCT
    object encoder (train-mode BatchNorm + Dropout) -> 4-table fused loss; two epochs run, stay finite and learn."""
    from sgaligner_amd.datasets import synthetic_scan3r as S
    from sgaligner_amd.epoch_trainer import Trainer
    root = str(tmp_path / 'data')
    S.write_dataset(root, n_pairs=8, seed=12, resolutions=(64,))
    cfg = S.make_cfg(root, modules=('pct', 'gat', 'rel', 'attr'), max_epoch=3, batch_size=4, output_dir=str(tmp_path / 'run'))
    np.random.seed(1)
    torch.manual_seed(1)
    tr = Trainer(cfg)
    hist = tr.run()
    assert len(hist) == 3
    assert all(np.isfinite(h['train']['loss']) and np.isfinite(h['val']['loss']) for h in hist)
    assert hist[-1]['train']['loss'] < hist[0]['train']['loss']
    sd = torch.load(osp.join(tr.snapshot_dir, 'epoch-3.pth.tar'), map_location='cpu', weights_only=False)['model']
    assert any(k.startswith('object_encoder.sa1.') for k in sd) and 'object_encoder.embedding.bn1.running_mean' in sd
