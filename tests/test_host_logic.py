"""Host-side logic that needs no GPU: the per-model hand-over slots of the operator, the row-scan job's life cycle, the
oracle farm (a case end to end in this process, the committed checksums, the bar arithmetic), the trainer's decision which
allmap channels an iteration needs, and the schedule script's presets."""
import math
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_raster_state_slots_are_per_object():
    from gaussmart_amd.rasterizer import RasterState
    a, b = RasterState(), RasterState()
    a.set_pending_param_event("event-a", "stream-a")
    a.color_grad = "record-a"
    assert b.pending is None and b.color_grad is None            # nothing is shared between two models' states
    assert a.pop_pending() == ("event-a", "stream-a") and a.pop_pending() is None
    assert a.take_color_grad() == "record-a" and a.take_color_grad() is None
    from gaussmart_amd.gaussian_model import GaussianModel
    m1, m2 = GaussianModel(3, device="cpu"), GaussianModel(3, device="cpu")
    assert isinstance(m1.raster_state, RasterState) and m1.raster_state is not m2.raster_state
    import gaussmart_amd.rasterizer as R
    for name in ("_PENDING_PARAM_EVENT", "_ROW_SCAN_JOB", "_COLOR_GRAD"):      # round 3's module-level dicts are gone
        assert not hasattr(R, name)


def test_row_scan_job_is_found_through_the_image_and_retired(monkeypatch):
    """take_row_scan_job hands a job out once, only for the image of ITS forward, only on the stream it was ordered on, and
    never after the rasterizer's backward retired it."""
    import gaussmart_amd.rasterizer as R
    stream = types.SimpleNamespace(cuda_stream=1234)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: stream)

    def image_with(job):
        return types.SimpleNamespace(grad_fn=types.SimpleNamespace(row_scan_job=job), device="cuda:0")

    job = types.SimpleNamespace(_taken=False, _dead=False, _stream=1234)
    assert R.take_row_scan_job(image_with(job)) is job and job._taken
    assert R.take_row_scan_job(image_with(job)) is None                          # handed out once
    assert R.take_row_scan_job(types.SimpleNamespace(grad_fn=None, device="cuda:0")) is None      # a derived / foreign tensor
    other = types.SimpleNamespace(_taken=False, _dead=False, _stream=99)
    assert R.take_row_scan_job(image_with(other)) is None and not other._taken   # ordered on another stream
    dead = types.SimpleNamespace(_taken=False, _dead=True, _stream=1234)
    assert R.take_row_scan_job(image_with(dead)) is None
    assert R.row_scan_job_alive(job) and not R.row_scan_job_alive(dead) and not R.row_scan_job_alive(None)


def test_factored_gradient_needs_a_state_and_cpu_tensors_are_refused():
    import gaussmart_amd.rasterizer as R
    with pytest.raises(ValueError):
        R.rasterize_gaussians_raw(torch.zeros(1, 3), None, None, None, None, None, None, None, factored_sh_grad=True)


def test_oracle_farm_case_in_process_and_bar_arithmetic():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_farm as F
    sp = F.spec("facing", 120, 48, 32, 3, sens_tols=(1e-3,))
    a, cam, bg, wc, wa = F.build_inputs(sp)
    b = F.build_inputs(sp)[0]
    assert all(torch.equal(a[k], b[k]) for k in a)                # seeded: the HIP side and a worker see the same bits
    res = F.run_case(sp)
    assert set(res["grads"]) == {"means3D", "opacities", "shs", "scales", "rotations", "means2D"}
    assert res["grads"]["shs"].shape == (120, 16, 3) and res["color"].shape == (3, 32, 48)
    assert set(res["d32"]) == set(res["grads"]) and 1e-3 in res["sens"] and res["sens"][1e-3][0].shape == (120,)
    go = {k: torch.from_numpy(v) for k, v in res["grads"].items()}
    # the fp32 oracle is within its own bars, a corrupted gradient is not
    n = 120
    stable = torch.ones(n, dtype=torch.bool)
    s32 = {k: F.summarize(None, go[k], n, rows=stable, d=torch.from_numpy(res["d32"][k]), trim=F.TRIM(n)) for k in go}
    same = {k: F.summarize(go[k], go[k], n, rows=stable, trim=F.TRIM(n)) for k in go}
    assert F.check_gradient_bars("unit", same, s32) == 0.0
    bad = {k: F.summarize(go[k] * 1.01, go[k], n, rows=stable, trim=F.TRIM(n)) for k in go}
    with pytest.raises(AssertionError):
        F.check_gradient_bars("unit-bad", bad, s32)
    del F.REPORT[:]
    # trimmed maximum: one wild row is left out of the norm-wise figure and reported on its own
    g = torch.ones(1000, 3)
    h = g.clone(); h[7] += 0.5
    st = F.summarize(h, g, 1000, trim=1)
    assert st["normwise"] == 0.0 and math.isclose(st["trimmed_max"], 0.5, rel_tol=1e-6)


def test_committed_oracle_checksums_cover_every_registered_case():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import oracle_farm as F
    import test_gpu_rasterizer, test_gpu_deep_lists, test_gpu_wide_payload      # noqa: F401  (register the cases)
    want = json.load(open(F.CHECKSUMS))
    assert set(F.FARM.specs) == set(want), sorted(set(F.FARM.specs) ^ set(want))
    for key, sums in want.items():
        for k, (s, sa) in sums.items():
            assert math.isfinite(s) and math.isfinite(sa) and sa >= 0.0, (key, k)   # (precomputed T + colours: dL/dmeans3D = 0)


def test_trainer_asks_the_forward_for_the_channels_the_objective_reads(monkeypatch):
    """training_step's choice of color_only / no_dist_median follows train.py:132-133 (lambda_normal from iteration 7,000,
    lambda_dist from 3,000) and the reference's defaults (lambda_dist = 0, depth_ratio = 0)."""
    import gaussmart_amd.trainer as T
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    seen = []

    class Stop(Exception):
        pass

    def fake_render(cam, g, pipe, bg, **kw):
        seen.append((kw.get("color_only"), kw.get("no_dist_median")))
        raise Stop

    monkeypatch.setattr(T, "render", fake_render)
    monkeypatch.setattr(T, "_use_factored_sh_grad", lambda *a: True)
    g = types.SimpleNamespace(update_learning_rate=lambda it: None, get_xyz=torch.zeros(1, 3), optimizer=None, active_sh_degree=3)
    for it, lam_n, lam_d, ratio in ((100, 0.05, 0.0, 0.0), (7001, 0.05, 0.0, 0.0), (7001, 0.0, 0.0, 0.0), (3001, 0.05, 100.0, 0.0),
                                    (2999, 0.05, 100.0, 0.0), (9000, 0.05, 0.0, 1.0)):
        opt, pipe = OptimizationParams(), PipelineParams()
        opt.lambda_normal, opt.lambda_dist, pipe.depth_ratio = lam_n, lam_d, ratio
        with pytest.raises(Stop):
            T.training_step(g, None, None, opt, pipe, None, it, render_fn=fake_render)
    assert seen == [(True, True), (False, True), (True, True), (False, False), (True, True), (False, False)]


def test_schedule_script_presets_match_bench():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench
    import full_schedule_train as FS
    for name in ("scan24", "bicycle", "headline"):
        p = bench.PRESETS[name]
        assert FS.PRESETS[name] == (p["gaussians"], p["width"], p["height"], p["radius_px"])


def test_loss_utils_refuses_what_it_does_not_implement():
    """gaussmart_amd.loss_utils (the reference's l1_loss / ssim signatures on the fused kernel): argument checks run before any
    device work, and a CPU tensor is refused -- there is no CPU path behind these names."""
    from gaussmart_amd import _lib, loss_utils as H
    a, b = torch.rand(3, 16, 16, requires_grad=True), torch.rand(3, 16, 16)
    with pytest.raises(ValueError):
        H.ssim(a, b, window_size=7)
    with pytest.raises(ValueError):
        H.ssim(a, b, size_average=False)
    with pytest.raises(ValueError):
        H.l1_loss(a, b.clone().requires_grad_(True))
    with pytest.raises(ValueError):
        H.l1_loss(a, b[:, :8])
    with pytest.raises(ValueError):
        H.ssim(torch.rand(2, 3, 16, 16), torch.rand(2, 3, 16, 16))      # a batch: the kernel takes one image
    with pytest.raises(_lib.GsrError):
        H.l1_loss(a, b)
    with pytest.raises(_lib.GsrError):
        H.ssim(a[None], b[None])


def test_reference_objective_switch_keeps_the_torch_maps(monkeypatch):
    """PipelineParams.reference_objective: training_step asks render() for the torch-derived maps (surface_maps=True), i.e. the
    reference's own objective formulation, instead of the fused node on allmap."""
    import gaussmart_amd.trainer as T
    from gaussmart_amd.params import OptimizationParams, PipelineParams
    seen = []

    class Stop(Exception):
        pass

    def fake_render(cam, g, pipe, bg, **kw):
        seen.append(kw.get("surface_maps"))
        raise Stop

    g = types.SimpleNamespace(update_learning_rate=lambda it: None, get_xyz=types.SimpleNamespace(is_cuda=True), optimizer=None,
                              active_sh_degree=3)
    for flag in (False, True):
        opt, pipe = OptimizationParams(), PipelineParams()
        pipe.reference_objective = flag
        pipe.factored_sh_grad = False
        with pytest.raises(Stop):
            T.training_step(g, None, None, opt, pipe, None, 8000, render_fn=fake_render)
    assert seen == [False, True]
    # ... and it wins over the factored path (whose forward builds no maps for torch to read)
    pipe = PipelineParams()
    pipe.reference_objective = True
    g.raster_state, g.optimizer = object(), T.FusedAdam.__new__(T.FusedAdam)
    assert not T._use_factored_sh_grad(g, pipe, T.render, True)
    pipe.reference_objective = False
    assert T._use_factored_sh_grad(g, pipe, T.render, True)
