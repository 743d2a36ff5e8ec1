"""The reference's import path: `sys.path` carries `<root>/sgaligner_amd` ahead of the reference's `src/` and
repository root, and the trainer / tester files do `from aligner.sg_aligner import *`, `from aligner.losses import *`,
`from utils import alignment` (src/trainers/trainval_sgaligner.py:6-12, src/inference/sgaligner/inference_align_reg.py:14-19),
then build the model and the loss exactly as trainval_sgaligner.py:41-66 does.  Runs in a fresh interpreter against a
stand-in reference tree (the real one does not travel), so the aliasing of sgaligner_amd/_dropin.py is what is tested."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.append('.')                                      # trainval_sgaligner.py:6-7 (cwd = <ref>/src)
sys.path.insert(0, SGA_DIR)                               # INTEGRATION.md 1: the one edit a maintainer makes
from aligner.sg_aligner import *                          # trainval_sgaligner.py:11
from aligner.losses import *                              # :12
from utils import torch_util, scan3r                      # inference_align_reg.py:16 -- must still be the reference's
from utils import alignment, common, point_cloud          # :19 -- alignment is ours, the rest the reference's
import aligner.eva                                        # trainval_eva.py: other portions of the namespace stay importable
from aligner.networks.pointnet import PointNetfeat as PN
from aligner.networks.gat import MultiGAT as MG
from aligner.networks.base import BaseNetwork
import sgaligner_amd.aligner.sg_aligner as canon
import sgaligner_amd.aligner.losses as canon_l
import sgaligner_amd.utils.alignment as canon_a
import aligner.sg_aligner, aligner.losses

assert aligner.sg_aligner is canon and aligner.losses is canon_l and alignment is canon_a
assert MultiModalEncoder is canon.MultiModalEncoder and OverallLoss is canon_l.OverallLoss
assert PN is canon.PointNetfeat and MG is canon.MultiGAT
assert torch_util.MARK == 'ref' and scan3r.MARK == 'ref' and common.MARK == 'ref' and point_cloud.MARK == 'ref'
assert aligner.eva.MultiGAT is MG
for fn in ('compute_mean_reciprocal_rank', 'compute_hits_k', 'compute_sgar', 'compute_node_corrs',
           'get_node_corrs_objects_ids', 'compute_alignment_score'):
    assert callable(getattr(alignment, fn)), fn
torch.set_grad_enabled(True)                              # trainval_sgaligner.py:93 uses `torch` from the star import
assert F is torch.nn.functional and nn is torch.nn

# ---- trainval_sgaligner.py:41-66 ------------------------------------------------------------------------------
class Cfg: pass
modules = ['point', 'gat', 'rel', 'attr']
device = torch.device('cpu')                              # construction only: forward needs the MI355X
model = MultiModalEncoder(modules=modules, rel_dim=41, attr_dim=164).to(device)
multi_loss_layer_icl = CustomMultiLossLayer(loss_num=len(modules), device=device)
multi_loss_layer_ial = CustomMultiLossLayer(loss_num=len(modules), device=device)
meta = {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': modules}
loss_func = OverallLoss(ial_loss_layer=multi_loss_layer_ial, icl_loss_layer=multi_loss_layer_icl, device=device, metadata=meta)
params = [{'params': list(model.parameters()) + list(loss_func.align_multi_loss_layer.parameters())
           + list(loss_func.contrastive_multi_loss_layer.parameters())}]
opt = torch.optim.Adam(params, lr=1e-3, weight_decay=0.0)
keys = set(model.state_dict().keys())
for k in ('object_encoder.conv1.weight', 'object_encoder.bn3.running_var', 'object_embedding.weight',
          'structure_encoder.layer_stack.0.att_src', 'structure_encoder.layer_stack.1.lin_src.weight',
          'structure_embedding.bias', 'meta_embedding_rel.weight', 'meta_embedding_attr.weight', 'fusion.weight'):
    assert k in keys, k
try:                                                      # no silent CPU path behind the reference's names either
    model({'tot_obj_pts': torch.zeros(2, 8, 3)})
    raise SystemExit('CPU tensors must raise')
except RuntimeError as e:
    assert 'no CPU path' in str(e)
print('DROPIN-OK', len(keys))
'''


def test_reference_import_recipe(tmp_path):
    ref = tmp_path / 'ref'
    (ref / 'src' / 'aligner').mkdir(parents=True)
    (ref / 'utils').mkdir()
    (ref / 'src' / 'aligner' / 'eva.py').write_text('from aligner.networks.gat import MultiGAT\n')
    for name in ('torch_util', 'scan3r', 'common', 'point_cloud', 'alignment'):
        (ref / 'utils' / f'{name}.py').write_text("MARK = 'ref'\n")
    script = tmp_path / 'run.py'
    script.write_text(f'SGA_DIR = {os.path.join(ROOT, "sgaligner_amd")!r}\n' + textwrap.dedent(SCRIPT))
    env = dict(os.environ)
    env['PYTHONPATH'] = str(ref)                           # the reference is run with its root on PYTHONPATH
    r = subprocess.run([sys.executable, str(script)], cwd=str(ref / 'src'), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DROPIN-OK' in r.stdout, r.stdout + r.stderr


def test_canonical_names_unaffected():
    """Importing the package the normal way never installs top-level aliases."""
    r = subprocess.run([sys.executable, '-c',
                        'import sys; sys.path.insert(0, %r); import sgaligner_amd.aligner.sg_aligner, sgaligner_amd.utils.alignment, '
                        'sgaligner_amd.datasets; assert "aligner" not in sys.modules and "utils" not in sys.modules; print("ok")' % ROOT],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr
