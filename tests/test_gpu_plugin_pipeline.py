"""``ns-train neurad-hip`` end to end at test size: the method's own pipeline, data manager and optimizers under the reference's
``Trainer.train_iteration`` (engine/trainer.py:535-579), on a synthetic drive.

A dataparser written here (3 cameras with rolling-shutter metadata, PNG images on disk, 2 lidar sweeps, 3 actors) feeds the
method's ``ADHipPipelineConfig`` exactly as PandaSet would: ``config.pipeline.setup(device=...)`` builds
``ADHipDataManager`` (images + scans cached in HBM, batches drawn by csrc/raygen.hip through data/pixel_samplers.py and
cameras/raygen.py), ``NeuRADHipModel``, and the reference's ``Optimizers`` build ``HashGridAdam`` for the tables from the
method's optimizer table.  Checked: the device batch against the reference's own samplers / ray generators fed the SAME draws
(bit-exact indices and ground truth, directions to a few ulps); K iterations of the reference's loop run without a
device->host read inside ``get_train_loss_dict``'s model calls and the loss goes down."""
import os
import sys
import types
from collections import defaultdict
from copy import deepcopy
from dataclasses import dataclass, field
from pathlib import Path
from typing import Type

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
import synth  # noqa: E402
import test_gpu_reference_plugin as t  # noqa: E402
from test_gpu_reference_plugin import ref  # noqa: E402,F401

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_import.reference_available(), reason="no reference (oracle/_ref ships with the lease)")]

H, W, N_CAM, N_LIDAR, PTS = 48, 72, 3, 2, 700


def _poses(n, seed):
    out = []
    for i in range(n):
        yaw = 0.2 * i + 0.05 * float(synth.normal((1,), seed + i)[0])
        c, s = np.cos(yaw), np.sin(yaw)
        # camera looking along +x of the world: columns = (right, up, back)
        rot = np.array([[s, 0.0, -c], [-c, 0.0, -s], [0.0, 1.0, 0.0]], np.float32)
        out.append(np.concatenate([rot, np.array([[2.0 * i], [0.3 * i], [1.6]], np.float32)], 1))
    return np.stack(out).astype(np.float32)


def make_parser_classes(root: Path):
    from nerfstudio.cameras.cameras import Cameras, CameraType
    from nerfstudio.cameras.lidars import Lidars, LidarType
    from nerfstudio.data.dataparsers.base_dataparser import DataParser, DataParserConfig, DataparserOutputs
    from nerfstudio.data.scene_box import SceneBox
    from PIL import Image

    files = []
    for i in range(N_CAM):
        img = (synth.uniform((H, W, 3), 0, 1, 300 + i) * 255).astype(np.uint8)
        f = root / f"cam{i}.png"
        Image.fromarray(img).save(f)
        files.append(f)

    @dataclass
    class SynthParserConfig(DataParserConfig):
        _target: Type = field(default_factory=lambda: SynthParser)
        data: Path = root
        add_missing_points: bool = True  # (read by ADPipeline.__init__, pipelines/ad_pipeline.py:71-73)

    class SynthParser(DataParser):
        includes_time = True

        def _generate_dataparser_outputs(self, split="train", **kwargs):
            T = t.T
            md = {"rolling_shutter_time": T(synth.uniform((N_CAM, 1), 0.01, 0.03, 6)),
                  "time_to_center_pixel": T(synth.uniform((N_CAM, 1), -0.01, 0.01, 7)),
                  "velocities": T(synth.normal((N_CAM, 3), 8) * 3), "sensor_idxs": torch.arange(N_CAM)[:, None] % 2}
            cams = Cameras(camera_to_worlds=T(_poses(N_CAM, 10)), fx=T(synth.uniform((N_CAM, 1), 60, 70, 1)),
                           fy=T(synth.uniform((N_CAM, 1), 60, 70, 2)), cx=float(W / 2), cy=float(H / 2), width=W, height=H,
                           camera_type=CameraType.PERSPECTIVE, times=T(synth.uniform((N_CAM, 1), 0.3, 3.5, 5)), metadata=md)
            l2w = _poses(N_LIDAR, 40)
            lid = Lidars(lidar_to_worlds=T(l2w), lidar_type=LidarType.VELODYNE64E, assume_ego_compensated=True,
                         times=T(synth.uniform((N_LIDAR, 1), 0.3, 3.5, 41)),
                         metadata={"velocities": T(synth.normal((N_LIDAR, 3), 42) * 3),
                                   "sensor_idxs": torch.full((N_LIDAR, 1), 2)}, valid_lidar_distance_threshold=1000.0)
            clouds = []
            for i in range(N_LIDAR):
                p = np.concatenate([synth.normal((PTS, 3), 44 + i) * np.array([15.0, 15.0, 1.0], np.float32),
                                    synth.uniform((PTS, 1), 0, 1, 46 + i), synth.uniform((PTS, 1), -0.05, 0.05, 48 + i)], -1)
                p[:40, :3] *= 200.0  # beams without a return
                clouds.append(T(p))
            return DataparserOutputs(
                image_filenames=list(files), cameras=cams,
                scene_box=SceneBox(aabb=torch.tensor([[-100.0] * 3, [100.0] * 3])),
                metadata={"lidars": lid, "point_clouds": clouds, "trajectories": t._trajectories(), "duration": 5.0,
                          "sensor_idx_to_name": {0: "cam0", 1: "cam1", 2: "lidar"}})

    return SynthParserConfig


def _method_config(tmp_path):
    """the ``neurad-hip`` method's trainer config, pointed at the synthetic drive and shrunk to test size"""
    methods = dict(__import__("nerfstudio.configs.method_configs", fromlist=["all_methods"]).all_methods)
    if "neurad-hip" not in methods:
        from nerfstudio.plugins.registry import discover_methods

        methods.update(discover_methods()[0])
    cfg = deepcopy(methods["neurad-hip"])
    pc = cfg.pipeline
    pc.ray_patch_size = (4, 4)
    pc.datamanager.dataparser = make_parser_classes(tmp_path)()
    pc.datamanager.train_num_rays_per_batch = 5 * 16
    pc.datamanager.train_num_lidar_rays_per_batch = 48
    pc.datamanager.eval_num_rays_per_batch = 16
    pc.datamanager.eval_num_lidar_rays_per_batch = 16
    pc.datamanager.pixel_sampler.patch_size, pc.datamanager.pixel_sampler.patch_scale = 4, pc.model.rgb_upsample_factor
    t._shrink(pc.model)
    pc.__post_init__()
    return cfg


@pytest.fixture()
def pipeline(ref, tmp_path):
    from neurad_studio_amd.integration.pipeline import ADHipDataManager, ADHipPipeline

    cfg = _method_config(tmp_path)
    pc = cfg.pipeline
    torch.manual_seed(0)
    pipe = pc.setup(device="cuda:0", test_mode="val", world_size=1, local_rank=0, grad_scaler=None)
    assert isinstance(pipe, ADHipPipeline) and isinstance(pipe.datamanager, ADHipDataManager)
    t._fill(pipe.model)
    return cfg, pipe


def test_device_batches_equal_the_references_samplers_on_the_same_draws(pipeline):
    """next_train on the device vs ScaledPatchSampler / LidarPointSampler / RayGenerator / LidarRayGenerator of the reference
    (data/pixel_samplers.py:474-765, model_components/ray_generators.py:27-90) on the CPU, fed the SAME random draws"""
    from nerfstudio.data.datamanagers.image_lidar_datamanager import _merge_img_lidar
    from nerfstudio.data.pixel_samplers import LidarPointSamplerConfig, ScaledPatchSamplerConfig
    from nerfstudio.model_components.ray_generators import LidarRayGenerator, RayGenerator

    cfg, pipe = pipeline
    dm = pipe.datamanager
    assert not dm.data_procs and dm._train_images["image"].is_cuda  # no worker processes; the images live in HBM
    draws = {}
    real_rand, real_randperm = torch.rand, torch.randperm

    def rec_rand(*a, **k):
        out = real_rand(*a, **k)
        draws.setdefault("rand", []).append(out.detach().cpu())
        return out

    def rec_randperm(*a, **k):
        out = real_randperm(*a, **k)
        draws["perm"] = out.detach().cpu()
        return out

    torch.rand, torch.randperm = rec_rand, rec_randperm
    try:
        rb, batch = dm.next_train(0)
    finally:
        torch.rand, torch.randperm = real_rand, real_randperm
    n_cam, n_lid = 5 * 16, 48
    assert len(rb) == n_cam + n_lid and batch["image"].shape == (5, 12, 12, 3) and batch["lidar"].shape[0] == n_lid
    assert bool(rb.metadata["is_lidar"][n_cam:].all()) and not bool(rb.metadata["is_lidar"][:n_cam].any())
    # the reference's samplers on the CPU, their torch.rand / randperm replaced by the recorded draws, in call order
    queue = list(draws["rand"])

    def play_rand(*a, **k):
        return queue.pop(0).to(k.get("dtype", torch.float32))

    ps = ScaledPatchSamplerConfig(patch_size=4, patch_scale=3).setup(num_rays_per_batch=n_cam)
    pts = LidarPointSamplerConfig().setup(num_rays_per_batch=n_lid)
    images = {"image": dm._train_images["image"].cpu(), "image_idx": dm._train_images["image_idx"].cpu()}
    points = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in dm._train_points.items()}
    torch.rand, torch.randperm = play_rand, lambda *a, **k: draws["perm"]
    try:
        ib = ps.sample(images)
        lb = pts.sample(points)
    finally:
        torch.rand, torch.randperm = real_rand, real_randperm
    irb = RayGenerator(dm.train_dataset.cameras)(ib["indices"])
    lrb = LidarRayGenerator(dm.train_lidar_dataset.lidars)(lb.pop("indices"), points=lb["lidar"])
    wrb, wbatch = _merge_img_lidar(irb, ib, lrb, lb, len(dm.train_dataset))
    assert torch.equal(batch["img_indices"].cpu(), wbatch["img_indices"])
    assert torch.equal(batch["image"].cpu(), wbatch["image"]) and torch.equal(batch["lidar"].cpu(), wbatch["lidar"])
    assert torch.equal(rb.camera_indices.cpu(), wrb.camera_indices)
    assert float((rb.directions.cpu() - wrb.directions).abs().max()) < 5e-7
    assert float((rb.origins.cpu() - wrb.origins).abs().max()) < 5e-6
    assert t.rel_l2(t.N(rb.pixel_area), t.N(wrb.pixel_area)) < 2e-4 and float((rb.times.cpu() - wrb.times).abs().max()) < 1e-6
    for k in ("is_lidar", "did_return"):
        assert torch.equal(rb.metadata[k].cpu(), wrb.metadata[k]) and torch.equal(batch[k].cpu(), wbatch[k])
    assert t.rel_l2(t.N(batch["distance"]), t.N(wbatch["distance"])) < 1e-6
    assert torch.equal(rb.metadata["sensor_idxs"].cpu(), wrb.metadata["sensor_idxs"])


def test_ns_train_neurad_hip_iterations_at_test_size(pipeline):
    """Trainer.train_iteration (the reference's function) over the method's own pipeline + optimizers, mixed precision on"""
    from nerfstudio.engine.optimizers import Optimizers
    from nerfstudio.engine.trainer import Trainer
    from torch.cuda.amp.grad_scaler import GradScaler

    from neurad_studio_amd.optim import HashGridAdam

    cfg, pipe = pipeline
    assert cfg.mixed_precision  # the neurad default (configs/method_configs.py:401)
    pipe.train()
    loop = types.SimpleNamespace(config=types.SimpleNamespace(log_gradients=False), device="cuda:0", mixed_precision=True,
                                 grad_scaler=GradScaler(enabled=True), gradient_accumulation_steps=defaultdict(lambda: 1),
                                 pipeline=pipe)
    groups = {k: v for k, v in pipe.get_param_groups().items() if len(v)}
    table = deepcopy({k: cfg.optimizers[k] for k in groups})
    for v in table.values():
        v["scheduler"].warmup_steps = 0
    loop.optimizers = Optimizers(table, groups)  # Trainer.setup_optimizers (engine/trainer.py:264-275)
    assert isinstance(loop.optimizers.optimizers["hashgrids"], HashGridAdam)
    assert pipe.model.fused_training_possible()
    losses = []
    torch.manual_seed(3)
    for step in range(12):
        loss, loss_dict, metrics = Trainer.train_iteration(loop, step)
        pipe.model.sampler.step_cb(step)
        losses.append(float(loss))
        assert set(loss_dict) >= {"rgb_loss", "interlevel_loss", "depth_loss", "carving_loss", "ray_drop_loss"}
        assert {"psnr", "depth_median_l2", "depth_mean_rel_l2", "intensity_rmse", "ray_drop_accuracy", "sdf_to_density",
                "traj_opt_translation"} <= set(metrics)
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-4:]) < np.mean(losses[:4]), losses  # it trains
    st = loop.optimizers.optimizers["hashgrids"].state[pipe.model.field.hashgrid.static_grid.hash_table]
    assert st["step"].is_cuda and float(st["step"]) >= 8  # (a GradScaler back-off may skip a few)


def test_fused_metrics_equal_the_references_get_metrics_dict(pipeline):
    """the sync-free training metrics (integration/neurad_hip.py:_fused_metrics_dict) against the reference's own
    get_metrics_dict (models/neurad.py:461-529) on the same outputs: same keys, same values"""
    cfg, pipe = pipeline
    pipe.train()
    t._deterministic(pipe.model, True)
    torch.manual_seed(5)
    rb, batch = pipe.datamanager.next_train(0)
    m = pipe.model
    out = m(rb, patch_size=(4, 4))
    got = m.get_metrics_dict(out, batch)
    m.config.fused_losses = False
    try:
        want = m.get_metrics_dict(out, batch)
    finally:
        m.config.fused_losses = True
    assert set(got) == set(want), set(got) ^ set(want)
    for k, w in want.items():
        a, c = float(got[k]), float(w)
        assert abs(a - c) <= 2e-5 * abs(c) + 1e-7, (k, a, c)


def test_ns_train_neurad_hip_from_config_setup_to_checkpoint(ref, tmp_path):
    """What ``ns-train neurad-hip`` does after parsing its arguments (scripts/train.py:96-107,246-261): ``config.setup()`` ->
    ``trainer.setup()`` -> ``trainer.train()`` -- here with the method's own config objects all the way: HipTrainer built by
    ``TrainerConfig.setup``, its TableGradScaler, ``ADHipPipeline`` / ``ADHipDataManager`` built by ``trainer.setup()``, the
    reference's ``Optimizers`` with HashGridAdam, the model's training callbacks, the reference's train loop with logging, and
    checkpoints written by ``save_checkpoint`` (engine/trainer.py:499-533) mid-run and at the end.  The checkpoint must hold what
    the reference's would: pipeline state, every optimizer's ``state_dict`` (HashGridAdam's in torch.optim.Adam's layout),
    scheduler counts that match the optimizer steps taken, the scaler; a second trainer resumes from it."""
    from nerfstudio.engine.optimizers import Optimizers

    from neurad_studio_amd.integration.pipeline import ADHipDataManager, ADHipPipeline
    from neurad_studio_amd.integration.trainer import HipTrainer
    from neurad_studio_amd.optim import HashGridAdam, TableGradScaler

    n_iter = 8
    cfg = _method_config(tmp_path)
    cfg.output_dir, cfg.experiment_name, cfg.timestamp = tmp_path / "outputs", "synthetic-drive", "run"
    cfg.vis = "none"  # no viewer, no event writer (neither viser nor tensorboard is installed here); the local writer stays
    cfg.max_num_iterations, cfg.steps_per_save = n_iter, 4
    cfg.steps_per_eval_batch = cfg.steps_per_eval_image = cfg.steps_per_eval_all_images = 10 ** 9
    cfg.logging.steps_per_log = 2
    cfg.viewer.quit_on_train_completion = True
    for group in cfg.optimizers.values():
        group["scheduler"].warmup_steps = 0
    cfg.get_base_dir().mkdir(parents=True)
    torch.manual_seed(0)
    trainer = cfg.setup(local_rank=0, world_size=1)
    assert type(trainer) is HipTrainer and isinstance(trainer.grad_scaler, TableGradScaler) and trainer.mixed_precision
    trainer.setup()
    assert isinstance(trainer.pipeline, ADHipPipeline) and isinstance(trainer.pipeline.datamanager, ADHipDataManager)
    assert isinstance(trainer.optimizers.optimizers["hashgrids"], HashGridAdam)
    assert len(trainer.callbacks) >= 1  # the model's own (sampler anneal / step callbacks, models/neurad.py:291-300)
    t._fill(trainer.pipeline.model)
    table = trainer.pipeline.model.field.hashgrid.static_grid.hash_table
    before = table.detach().clone()
    trainer.train()
    assert getattr(trainer, "_pending_scheduler_step", None) is None  # the last save_checkpoint settled it
    ckpt = trainer.checkpoint_dir / f"step-{n_iter - 1:09d}.ckpt"
    assert sorted(p.name for p in trainer.checkpoint_dir.glob("*.ckpt")) == [ckpt.name]  # save_only_latest_checkpoint
    state = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert state["step"] == n_iter - 1 and set(state) >= {"pipeline", "optimizers", "schedulers", "scalers"}
    hg = state["optimizers"]["hashgrids"]
    idx = [i for i, p in enumerate(trainer.optimizers.optimizers["hashgrids"].param_groups[0]["params"]) if p is table][0]
    steps = float(hg["state"][idx]["step"])
    scale = state["scalers"]["scale"]
    skipped = n_iter - steps
    assert 1 <= steps <= n_iter and scale == 65536.0 * 0.5 ** skipped
    for name, sched in state["schedulers"].items():
        assert sched["last_epoch"] == steps, (name, sched["last_epoch"], steps)  # one scheduler step per optimizer step taken
    assert not torch.equal(table.detach(), before)
    assert torch.equal(state["pipeline"]["_model.field.hashgrid.static_grid.hash_table"], table.detach().cpu())

    # resume (ns-train --load-dir): pipeline + scaler through Trainer._load_checkpoint, optimizers / schedulers through the
    # reference's own loaders (its setup() calls _load_checkpoint before the optimizers exist: load_optimizer off there)
    cfg2 = deepcopy(cfg)
    cfg2.load_dir, cfg2.load_optimizer, cfg2.load_scheduler, cfg2.timestamp = trainer.checkpoint_dir, False, False, "resumed"
    cfg2.get_base_dir().mkdir(parents=True)
    resumed = cfg2.setup(local_rank=0, world_size=1)
    # (the reference's schedulers compute their rates with numpy, so `_last_lr` in a checkpoint is a numpy scalar; its
    #  `torch.load(path, map_location="cpu")` predates torch 2.6's weights_only default: allow-list what the file holds)
    with torch.serialization.safe_globals([np.core.multiarray.scalar, np.dtype, type(np.dtype(np.float64)), type(np.dtype(np.int64))]):
        resumed.setup()
    assert resumed._start_step == n_iter and resumed.grad_scaler.get_scale() == scale
    assert torch.equal(resumed.pipeline.model.field.hashgrid.static_grid.hash_table.detach(), table.detach())
    assert isinstance(resumed.optimizers, Optimizers)
    resumed.optimizers.load_optimizers(state["optimizers"])
    resumed.optimizers.load_schedulers(state["schedulers"])
    a, b = trainer.optimizers.optimizers["hashgrids"], resumed.optimizers.optimizers["hashgrids"]
    for pa, pb in zip(a.param_groups[0]["params"], b.param_groups[0]["params"]):
        if pa in a.state:
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(a.state[pa][k], b.state[pb][k].to(a.state[pa][k].device)), k
            assert float(a.state[pa]["step"]) == float(b.state[pb]["step"])
    loss, _, _ = resumed.train_iteration(n_iter)  # and it keeps training from there
    assert torch.isfinite(loss)
