"""GPU: run-to-run drift of the training step at BASELINE configs[1] size.  The library is built with -munsafe-fp-atomics and the
loss / weight-gradient kernels accumulate through fp32 atomics, so sums are order-dependent: two runs of the SAME step on the SAME
batch and weights may differ in the last bits.  This bounds that noise: the loss must repeat to 1e-6 relative and every parameter
gradient to 2e-4 of the model's largest gradient entry (an order of magnitude inside the 1e-3 parity bar), over three repetitions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_c2_step_repeats_within_bound():
    from sgaligner_amd.synthetic import make_batch_fast
    from sgaligner_amd.trainer import AlignerSteps
    steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda:0', seed=42)
    dd = make_batch_fast(512, 64, 512, seed=43, device='cuda:0')
    runs = []
    for _ in range(3):
        _, ld = steps.forward_backward(dd)
        torch.cuda.synchronize()
        runs.append((float(ld['loss'].item()), {n: p.grad.detach().clone() for n, p in steps.model.named_parameters() if p.grad is not None}))
    l0, g0 = runs[0]
    gmax = max(float(g.abs().max()) for g in g0.values())
    worst = 0.0
    for l, g in runs[1:]:
        assert abs(l - l0) <= 1e-6 * abs(l0), (l, l0)
        for n in g0:
            worst = max(worst, float((g[n] - g0[n]).abs().max()) / gmax)
    assert worst <= 2e-4, worst
    # per parameter, relative to that parameter's own largest entry: well inside the parity tolerance for every weight matrix
    # (biases whose gradient is a small difference of large sums are the noisiest: bounded at 5e-3 of their own maximum)
    for l, g in runs[1:]:
        for n in g0:
            own = float((g[n] - g0[n]).abs().max()) / max(1e-30, float(g0[n].abs().max()))
            assert own <= (5e-3 if n.endswith('bias') else 1e-3), (n, own)


def test_c3_step_repeats_within_bound():
    """The same bound at BASELINE configs[2] (4096 pairs x 128 objects x 512 points on one GPU, the headline): loss to 1e-6 relative, every
    gradient to 2e-4 of the model's largest entry; per parameter 1e-3 of its own maximum (5e-3 for biases) -- except meta_embedding_rel.*,
    whose gradient is a 1e-4-sized remainder of 10^6-term sums (its exact-fp32 value is itself ~1.5e-2 of its maximum away from fp64,
    profiles/r04_c3_gradient_vs_fp64.json): bounded at 1e-2 of its own maximum (measured 2.6e-3 .. 3.6e-3)."""
    from sgaligner_amd.synthetic import make_batch_fast
    from sgaligner_amd.trainer import AlignerSteps
    torch.cuda.empty_cache()
    steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda:0', seed=42)
    dd = make_batch_fast(4096, 128, 512, seed=43, device='cuda:0')
    runs = []
    for _ in range(2):
        _, ld = steps.forward_backward(dd)
        torch.cuda.synchronize()
        runs.append((float(ld['loss'].item()), {n: p.grad.detach().clone() for n, p in steps.model.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = runs
    gmax = max(float(g.abs().max()) for g in g0.values())
    assert abs(l1 - l0) <= 1e-6 * abs(l0), (l1, l0)
    for n in g0:
        d = float((g1[n] - g0[n]).abs().max())
        assert d / gmax <= 2e-4, (n, d / gmax)
        own = d / max(1e-30, float(g0[n].abs().max()))
        bound = 1e-2 if n.startswith('meta_embedding_rel') else (5e-3 if n.endswith('bias') else 1e-3)
        assert own <= bound, (n, own)
