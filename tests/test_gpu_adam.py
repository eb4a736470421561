"""FusedAdam (one HIP launch) vs torch.optim.Adam configured like the reference
(scene/gaussian_model.py:282-295: six groups, eps 1e-15)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 2), (1001, 4)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    return ps, [{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))]


def test_matches_torch_adam(gpu_device):
    from gaussmart_amd.fused_adam import FusedAdam
    pa, ga = _groups(gpu_device, 0)
    pb, gb = _groups(gpu_device, 0)
    oa = FusedAdam(ga, lr=0.0, eps=1e-15)
    ob = torch.optim.Adam(gb, lr=0.0, eps=1e-15, foreach=False, fused=False)
    gen = torch.Generator().manual_seed(1)
    for it in range(25):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=gen).to(gpu_device) * (10.0 ** (it % 5 - 3))
            if it % 7 == 3:
                gr[::2] = 0           # invisible Gaussians: zero gradient, moments still decay
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 10:
            ga[0]["lr"] = gb[0]["lr"] = 3e-5           # schedule changes the xyz lr every iteration
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 3e-5
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
        # moments: rounding noise is relative to the gradient scale (cancellation in g - m)
        ea, eb = oa.state[a]["exp_avg"], ob.state[b]["exp_avg"]
        torch.testing.assert_close(ea, eb, rtol=2e-6, atol=1e-6 * float(eb.abs().max()))
        va, vb = oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]
        torch.testing.assert_close(va, vb, rtol=2e-6, atol=1e-6 * float(vb.abs().max()))
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == 25


def test_survives_densification_surgery(gpu_device):
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.fused_adam import FusedAdam
    from gaussmart_amd.params import OptimizationParams
    from gaussmart_amd.synthetic import make_scene
    params, _ = make_scene(3000, 64, 64)
    m = GaussianModel(3, device=gpu_device)
    m.create_from_params(params)
    m.training_setup(OptimizationParams())
    assert isinstance(m.optimizer, FusedAdam)
    for p in m.parameters():
        p.grad = torch.randn_like(p)
    m.optimizer.step()
    m.xyz_gradient_accum += 1e-3; m.denom += 1
    m.densify_and_prune(0.0002, 0.05, 5.0, 20)
    m.reset_opacity()
    n = m.get_xyz.shape[0]
    for p in m.parameters():
        assert p.shape[0] == n
        p.grad = torch.randn_like(p)
    m.optimizer.step()
    assert m.optimizer.state[m._xyz]["exp_avg"].shape[0] == n
    sd = m.optimizer.state_dict()
    assert len(sd["state"]) == 6
