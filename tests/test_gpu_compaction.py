"""Fused row compaction against torch boolean indexing, and pruning a model through it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p_keep", [(1, 1.0), (1000, 0.5), (4097, 0.03), (100000, 0.9), (5000, 0.0)])
def test_compact_rows_equals_boolean_indexing(gpu_device, n, p_keep):
    from gaussmart_amd.compaction import compact_rows
    g = torch.Generator().manual_seed(n)
    keep = (torch.rand(n, generator=g) < p_keep).to(gpu_device)
    shapes = [(n, 3), (n, 1, 3), (n, 15, 3), (n, 1), (n, 2), (n, 4), (n,)]
    tensors = [torch.randn(*s, generator=g).to(gpu_device) for s in shapes]
    tensors.append(torch.randint(0, 1 << 40, (n,), generator=g).to(gpu_device))          # int64 segments
    tensors += [torch.randn(n, 7, generator=g).to(gpu_device) for _ in range(20)]          # > 24 tensors: two launches
    got = compact_rows(tensors, keep)
    for t, o in zip(tensors, got):
        assert o.dtype == t.dtype and torch.equal(o, t[keep])


def test_prune_points_on_device_equals_host(gpu_device):
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.params import OptimizationParams
    from gaussmart_amd.synthetic import make_scene
    params, _ = make_scene(3000, 64, 64, seed=1)
    mask = torch.rand(3000, generator=torch.Generator().manual_seed(2)) < 0.3
    models = []
    for dev in (gpu_device, torch.device("cpu")):
        m = GaussianModel(3, device=dev)
        m.use_fused_adam = False if dev.type == "cpu" else m.use_fused_adam
        m.create_from_params(params)
        m.training_setup(OptimizationParams())
        for p in m.parameters():                         # give Adam some state to carry along
            p.grad = torch.ones_like(p) * 0.01
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        m.max_radii2D += 3.0
        m.prune_points(mask.to(dev))
        models.append(m)
    a, b = models
    assert a.get_xyz.shape[0] == b.get_xyz.shape[0] == int((~mask).sum())
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa.detach().cpu(), pb.detach(), rtol=1e-6, atol=1e-7)
        sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
        torch.testing.assert_close(sa["exp_avg"].cpu(), sb["exp_avg"], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(sa["exp_avg_sq"].cpu(), sb["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    assert torch.equal(a.max_radii2D.cpu(), b.max_radii2D) and torch.equal(a._segments.cpu(), b._segments)


def test_fused_densification_stats_equal_reference_ops(gpu_device):
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.synthetic import make_scene
    params, _ = make_scene(5000, 64, 64, seed=3)
    g = torch.Generator().manual_seed(4)
    radii = torch.randint(-1, 30, (5000,), generator=g, dtype=torch.int32)
    grad = torch.randn(5000, 3, generator=g)
    outs = []
    for dev in (gpu_device, torch.device("cpu")):
        m = GaussianModel(3, device=dev)
        m.create_from_params(params)
        m.xyz_gradient_accum = torch.rand(5000, 1, generator=torch.Generator().manual_seed(5)).to(dev)
        m.denom = torch.ones(5000, 1, device=dev)
        m.max_radii2D = torch.full((5000,), 7.0, device=dev)
        pts = torch.zeros(5000, 3, device=dev, requires_grad=True)
        pts.grad = grad.to(dev)
        m.update_densification_stats(pts, radii.to(dev))
        outs.append((m.max_radii2D.cpu(), m.xyz_gradient_accum.cpu(), m.denom.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-6, atol=1e-7)
