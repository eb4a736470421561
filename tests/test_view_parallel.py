"""World-size-2 gloo test of the view-parallel path on CPU: one replica per rank, different
views, one flat all-reduce of the per-Gaussian gradients, replicated densification."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussmart_amd import gaussian_renderer
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.synthetic import make_scene, jittered_cameras
    from gaussmart_amd.trainer import training_step, densification_step
    from gaussmart_amd.view_parallel import ViewParallel
    from oracle import surfel_ref as O
    gaussian_renderer.GaussianRasterizer = O.OracleRasterizer      # test-only stand-in for the HIP operator

    params, _ = make_scene(150, 48, 48, seed=0)
    cams = jittered_cameras(world, 48, 48, seed=0, device="cpu")
    gt = torch.rand(3, 48, 48, generator=torch.Generator().manual_seed(5))
    opt, pipe, bg = OptimizationParams(), PipelineParams(), torch.zeros(3)

    m = GaussianModel(3, device="cpu"); m.use_fused_adam = False
    m.create_from_params(params); m.training_setup(opt)
    vp = ViewParallel(m)
    assert vp.world_size == world and vp.rank == rank
    shard = vp.shard_views(list(range(10)), epoch_seed=3)
    pkg, _ = training_step(m, cams[rank], gt, opt, pipe, bg, 601, view_parallel=vp, step_optimizer=False)
    grads = [p.grad.clone() for p in m.parameters()]
    # densification with synced statistics and a replicated generator
    for it in (601, 700):
        if it != 601:
            pkg, _ = training_step(m, cams[rank], gt, opt, pipe, bg, it, view_parallel=vp, step_optimizer=False)
        densification_step(m, pkg, opt, it, cameras_extent=5.0, view_parallel=vp)
        m.optimizer.step(); m.optimizer.zero_grad(set_to_none=True)
    torch.save({"grads": grads, "shard": shard, "xyz": m.get_xyz.detach().clone(), "n": m.get_xyz.shape[0]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_view_parallel_world2(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(world)]
    # identical averaged gradients and identical replicas after a synced densify step
    for a, b in zip(r[0]["grads"], r[1]["grads"]):
        assert torch.equal(a, b)
    assert r[0]["n"] == r[1]["n"] and torch.equal(r[0]["xyz"], r[1]["xyz"])
    assert sorted(r[0]["shard"] + r[1]["shard"]) == list(range(10)) and not set(r[0]["shard"]) & set(r[1]["shard"])

    # the all-reduced gradient is the mean of the two single-view gradients
    from gaussmart_amd import gaussian_renderer
    from gaussmart_amd.gaussian_model import GaussianModel
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    from gaussmart_amd.synthetic import make_scene, jittered_cameras
    from gaussmart_amd.trainer import training_step
    from oracle import surfel_ref as O
    old = gaussian_renderer.GaussianRasterizer
    gaussian_renderer.GaussianRasterizer = O.OracleRasterizer
    try:
        params, _ = make_scene(150, 48, 48, seed=0)
        cams = jittered_cameras(world, 48, 48, seed=0, device="cpu")
        gt = torch.rand(3, 48, 48, generator=torch.Generator().manual_seed(5))
        single = []
        for cam in cams:
            m = GaussianModel(3, device="cpu"); m.use_fused_adam = False
            m.create_from_params(params); m.training_setup(OptimizationParams())
            training_step(m, cam, gt, OptimizationParams(), PipelineParams(), torch.zeros(3), 601, step_optimizer=False)
            single.append([p.grad.clone() for p in m.parameters()])
    finally:
        gaussian_renderer.GaussianRasterizer = old
    for k in range(6):
        torch.testing.assert_close(r[0]["grads"][k], 0.5 * (single[0][k] + single[1][k]), rtol=1e-5, atol=1e-9)


def _flat_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussmart_amd.view_parallel import ViewParallel

    class Toy:   # six parameters whose gradients are adjacent views of ONE buffer, as the fused backward makes them
        def __init__(self):
            g = torch.Generator().manual_seed(0)
            self.ps = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in ((7, 3), (7, 1, 3), (7, 15, 3), (7, 1), (7, 2), (7, 4))]
            self._features_rest = self.ps[2]
        def parameters(self):
            return self.ps

    toy = Toy()
    sizes = [toy.ps[i].numel() for i in (0, 1, 3, 4, 5, 2)]          # buffer order: f_rest last
    flat = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1)
    for i, part in zip((0, 1, 3, 4, 5, 2), torch.split(flat, sizes)):
        toy.ps[i].grad = part.view_as(toy.ps[i])
    vp = ViewParallel(toy)
    base = vp.flat_gradient([p.grad for p in toy.ps])
    assert base is not None and base.numel() == flat.numel() and base.data_ptr() == flat.data_ptr()
    vp.allreduce_gradients()
    # not adjacent -> no flat view; the per-tensor path still averages
    loose = [torch.full_like(p, float(rank + 1)) for p in toy.ps]
    assert vp.flat_gradient(loose) is None
    torch.save({"flat": flat.clone()}, os.path.join(out_dir, f"flat{rank}.pt"))
    dist.destroy_process_group()


def test_flat_gradient_buffer_is_reduced_in_one_piece(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_flat_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(os.path.join(tmp_path, f"flat{i}.pt"))["flat"] for i in range(world)]
    want = torch.arange(r[0].numel(), dtype=torch.float32) * 1.5      # mean of x*1 and x*2
    assert torch.equal(r[0], r[1]) and torch.allclose(r[0], want)


def _factored_worker(rank, world, port, out_dir):
    """The factored exchange of the view-parallel step (view_parallel.ViewParallel.exchange_factored) over gloo:
    geometry gradients all-reduced in place, colour-gradient records all-gathered in rank order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussmart_amd.rasterizer import ColorGradRecord
    from gaussmart_amd.view_parallel import ViewParallel
    from gaussmart_amd.sh import eval_sh

    n, deg = 37, 3
    g0 = torch.Generator().manual_seed(0)                      # replicated model
    xyz = torch.randn(n, 3, generator=g0) * 2
    sh = (torch.randn(n, 16, 3, generator=g0) * 0.3).requires_grad_(True)
    gr = torch.Generator().manual_seed(10 + rank)              # this rank's view
    campos = torch.randn(3, generator=gr) * 0.2 + torch.tensor([0.0, 0.0, -5.0])
    upstream = torch.randn(n, 3, generator=gr)
    upstream[rank::3] = 0.0                                    # Gaussians this view does not see
    # explicit single-view SH gradient: autograd through the reference-parity colour function
    d = xyz - campos
    rgb = eval_sh(deg, sh.transpose(1, 2), d / d.norm(dim=1, keepdim=True)) + 0.5
    mask = (rgb > 0).float()                                   # clamp_min(0) passes no gradient where it clamps
    (torch.clamp_min(rgb, 0.0) * upstream).sum().backward()
    # what the factored backward leaves: [geometry gradients | masked colour gradient [N,3] + camera position]
    n_head = 10 * n + 2                                        # some padding in the head, as alignment may add
    flat = torch.zeros(n_head + 3 * n + 4)
    flat[:n_head] = torch.arange(n_head, dtype=torch.float32) * (rank + 1)
    flat[n_head:n_head + 3 * n] = (upstream * mask).reshape(-1)
    flat[n_head + 3 * n:n_head + 3 * n + 3] = campos
    rec = ColorGradRecord(flat, flat[:n_head], flat[n_head:], n, 16, deg, xyz)

    class Model:
        def parameters(self):
            return [xyz]
    vp = ViewParallel(Model())
    vp.exchange_factored(rec)
    assert rec.exchanged and rec.n_views == world and rec.grad_scale == 1.0 / world
    torch.save({"head": rec.head.clone(), "gathered": rec.gathered.clone(), "explicit": sh.grad.clone(),
                "record": rec.record.clone(), "xyz": xyz}, os.path.join(out_dir, f"fact{rank}.pt"))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])      # 8 = the rank count of the driver's scaling run, rehearsed on gloo
def test_factored_exchange(tmp_path, world):
    from gaussmart_amd.sh import sh_basis
    port = _free_port()
    mp.start_processes(_factored_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(os.path.join(tmp_path, f"fact{i}.pt")) for i in range(world)]
    n = r[0]["xyz"].shape[0]
    stride = 3 * n + 4
    # every rank holds the records of all ranks, in rank order, and the averaged geometry gradients
    for k in range(1, world):
        assert torch.equal(r[0]["gathered"], r[k]["gathered"]) and torch.equal(r[0]["head"], r[k]["head"])
    for k in range(world):
        assert torch.equal(r[0]["gathered"][k * stride:(k + 1) * stride], r[k]["record"])
    want_head = torch.arange(r[0]["head"].numel(), dtype=torch.float32) * ((world + 1) / 2.0)   # mean of x * (rank + 1)
    assert torch.allclose(r[0]["head"], want_head)
    # the gradient the SH optimiser step rebuilds -- mean over views of basis(dir_v) x g_v -- is the mean of the explicit
    # single-view gradients (what an all-reduce of the [N,16,3] tensors would have delivered)
    rebuilt = torch.zeros(n, 16, 3, dtype=torch.float64)
    for k in range(world):
        rec = r[0]["gathered"][k * stride:(k + 1) * stride].double()
        g, campos = rec[:3 * n].view(n, 3), rec[3 * n:3 * n + 3]
        d = r[0]["xyz"].double() - campos
        rebuilt += sh_basis(3, d / d.norm(dim=1, keepdim=True))[:, :, None] * g[:, None, :] / world
    explicit = sum(x["explicit"].double() for x in r) / world
    assert (rebuilt - explicit).abs().max().item() <= 1e-6 * explicit.abs().max().item()
    assert explicit.abs().max().item() > 0
