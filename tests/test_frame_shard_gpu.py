"""Frame-sharded single-clip mode (BASELINE.json configs[2]; camc2v_amd/parallel.py): the 16 frames of every clip split over the
ranks of a process group.  Two (and four) worker processes share the one GPU of the test box and talk over gloo (RCCL refuses two
ranks on one device; on a node the same code runs on backend "nccl" = RCCL over xGMI); every rank compares the sharded forward
with the unsharded one it computes itself on the well-conditioned medium fixture (model_channels 128, 16x16 latents, native
patch-ordered epipolar masks).  Stated tolerances: (1) layer level -- a camera-conditioned temporal transformer (clip-wide
GroupNorm, temporal attention, Pluecker rows, epipolar attention on local mask rows), a temporal conv block (halos) and a spatial
transformer with per-frame image tokens (frame offset) on random rows: rel-L2 <= 1e-3 against this rank's rows of the unsharded
layer (measured 1e-5 .. 1e-4); (2) whole UNet: as close to the REFERENCE's fp32 output as the unsharded path is (fixture bound
2.5e-2 / 5e-2; measured 2.16e-2 against 2.11e-2), and within 3e-2 of the unsharded HIP result -- with half the rows per rank most
kernels take other tile / split-K / single-launch variants, i.e. another rounding realisation of the same bf16 arithmetic, and on
this seeded-weight fixture two such realisations differ by about as much as each differs from the truth."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, golden_dir, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from camc2v_amd import camera
    from oracle.golden_inputs import MEDIUM_CFG, SEED, medium_inputs
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(golden_dir, "unet_medium.npz"))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": MEDIUM_CFG})
    unet.enable_camera_conditioning(dict(origin_h=128, origin_w=128, is_3d_full_attn=False, num_register_tokens=4,
                                         attention_resolution=[8, 4, 2, 1], compression_factor=1))
    unet.epipolar_origin_h = 128
    unet.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    unet = unet.to(dev).eval()
    inp = medium_inputs()
    g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    packed = camera.epipolar_masks_packed(torch.from_numpy(fx["F128"]).to(dev), 16, 128, 128)      # patch-ordered native masks
    cam = dict(pluker_embedding_features=[f.to(dev) for f in inp["feats"]], sample_locs_dict=None, sample_locs_packed=packed,
               add_type="add_to_main_branch")
    res = {}

    def rel(a, b):
        a, b = a.float(), b.float()
        return ((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()

    cases = {
        "camera, repeated context": dict(context=g["ctx_rep"], camera_condition=cam),
        "camera, per-frame context": dict(context=g["ctx_pf"], camera_condition=cam),
        "no camera, per-frame context": dict(context=g["ctx_pf"]),
        "CFG pair, shared input": dict(context=[g["ctx_rep"], g["ctx_pf"]], camera_condition=cam, cfg_shared_input=True),
    }
    from camc2v_amd import ops, parallel

    for name, kw in cases.items():
        want = unet(g["x"], g["t"], fs=g["fs"], **kw)
        unet.enable_frame_sharding()
        c0 = unet.frame_shard.collectives
        got = unet(g["x"], g["t"], fs=g["fs"], **kw)
        res.setdefault("collectives per forward", []).append(unet.frame_shard.collectives - c0)
        unet.disable_frame_sharding()
        assert got.shape == want.shape and torch.isfinite(got).all()
        res[name] = rel(got, want)
        if name == "camera, repeated context":
            res["vs reference fixture, camera"] = rel(got.cpu(), torch.from_numpy(fx["y_cam_rep"]))
            res["vs reference fixture, camera (unsharded, for comparison)"] = rel(want.cpu(), torch.from_numpy(fx["y_cam_rep"]))
        if name == "no camera, per-frame context":
            res["vs reference fixture, no camera"] = rel(got.cpu(), torch.from_numpy(fx["y_nocam_pf"]))
        parts = [torch.empty_like(got) for _ in range(world)]          # every rank ends up with the same full tensor
        dist.all_gather(parts, got)
        res["ranks agree"] = max(res.get("ranks agree", 0.0), float(max((p - parts[0]).abs().max().item() for p in parts)))
    # ---- layer level, tight: one camera-conditioned temporal transformer (GroupNorm over the clip, temporal attention, Pluecker
    # rows, epipolar attention on this rank's mask rows against all keys), one temporal conv block (halos), one spatial transformer
    # with per-frame image tokens (frame offset) -- this rank's rows of the unsharded result against the sharded layer
    from camc2v_amd.unet import Geom, SpatialTransformer, TemporalConvBlock, TemporalTransformer
    shard = parallel.FrameShard()
    b, T, H, W = g["ctx_pf"].shape[0], 16, 16, 16          # as many clips as the fixture's contexts / camera features describe
    hw = H * W
    gen = torch.Generator().manual_seed(5)
    tt = next(m for m in unet.modules() if isinstance(m, TemporalTransformer) and hasattr(m.transformer_blocks[0], "epipolar") and m.ds == 1)
    tc = next(m for m in unet.modules() if isinstance(m, TemporalConvBlock) and m.in_channels == tt.in_channels)
    st = next(m for m in unet.modules() if isinstance(m, SpatialTransformer) and m.in_channels == tt.in_channels)
    x = torch.randn(b * T * hw, tt.in_channels, generator=gen).to(dev)
    full = unet._camera_inputs(cam, b, T, H, W)
    cam_full = dict(rows=full["rows"][0], mask=full["masks"][8], add_type="add_to_main_branch")
    all_blocks = [blk for m in unet.modules() if isinstance(m, SpatialTransformer) for blk in m.transformer_blocks]
    per_block = list(unet._context_groups(g["ctx_pf"], T)[1])          # per-frame image tokens (77 + 16 T)
    groups = [per_block[all_blocks.index(blk)] for blk in st.transformer_blocks]
    assert all(grp[0][5] for grp in groups)
    wants = dict(tt=tt.forward_rows(x, Geom(b, T, H, W), cam_full), tc=tc.forward_rows(x, Geom(b, T, H, W)),
                 st=st.forward_rows(x, Geom(b, T, H, W), groups))
    with parallel.FrameCtx(shard, T) as fc:
        xl = fc.local_frames(x, b, hw)
        loc = unet._local_camera_inputs(full, fc, H, W)
        gl = Geom(b, fc.t_loc, H, W)
        res["layer: temporal transformer + epipolar"] = rel(tt.forward_rows(xl, gl, dict(rows=loc["rows"][0], mask=loc["masks"][8], add_type="add_to_main_branch")),
                                                            fc.local_frames(wants["tt"], b, hw))
        res["layer: temporal conv block"] = rel(tc.forward_rows(xl, gl), fc.local_frames(wants["tc"], b, hw))
        res["layer: spatial transformer, per-frame tokens"] = rel(st.forward_rows(xl, gl, groups), fc.local_frames(wants["st"], b, hw))
    # ---- the sampler on top: the 25-step CFG 7.5 trajectory of tests/test_trajectory_gpu.py (fixture produced by RUNNING THE
    # REFERENCE's DDIMSampler + UNet, oracle/gen_golden_traj.py) with the frames sharded; eager (collectives are not captured);
    # every rank holds whole clips after each step
    from utils.utils import instantiate_from_config as inst
    tr = np.load(os.path.join(golden_dir, "traj_medium.npz"))
    model = inst({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(MEDIUM_CFG)}, linear_start=0.00085, linear_end=0.012,
        conditioning_key="hybrid", channels=4, image_size=[16, 16], temporal_length=16, add_type="add_to_main_branch",
        pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": {}},
        epipolar_config=dict(origin_h=128, origin_w=128, is_3d_full_attn=False, num_register_tokens=4, attention_resolution=[8, 4, 2, 1],
                             compression_factor=1))})
    model.model.diffusion_model.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    model = model.to(dev).eval()
    torch.manual_seed(int(tr["noise_seed_cam"]))
    zs = [torch.randn(1, 4, 16, 16, 16) for _ in range(25)]
    assert np.allclose([float(z.double().sum()) for z in zs], tr["cam_noise_checksum"], rtol=0, atol=1e-6)
    cam2 = dict(cam, cond_frame_index=torch.zeros(1, dtype=torch.long, device=dev))
    cond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_rep"]], camera_condition=cam2)
    uncond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_pf"]])
    model.model.diffusion_model.enable_frame_sharding()
    samples, inter = model.sample_log(cond, 1, True, 25, unconditional_conditioning=uncond, log_every_t=1, eta=1.0, x_T=inp["x_T"],
                                      unconditional_guidance_scale=7.5, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                      fs=g["fs"], enable_camera_condition=True, injected_noise=zs)
    xs = inter["x_inter"][1:]
    worst = 0.0
    for i, ref in zip(tr["keep_steps"], tr["cam_x_steps"]):
        got_i = xs[int(i)].float().cpu()
        worst = max(worst, ((got_i - torch.from_numpy(ref)).norm() / torch.from_numpy(ref).norm()).item())
    res["25-step CFG trajectory vs REFERENCE"] = (worst, 0.0)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2] + ([4] if os.environ.get("CCV_TEST_SHARD4") == "1" else []))   # 4 ranks pass too (measured); one spawn keeps the suite short
def test_frame_sharded_forward_equals_unsharded(world, golden_dir, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    mp.spawn(_worker, args=(world, _free_port(), golden_dir, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = json.load(open(tmp_path / f"rank{r}.json"))
        for name, v in res.items():
            print(f"[parity] frame shard {world} ranks, rank {r}, {name}: {v}")
        assert res["ranks agree"] == 0.0
        # one collective per temporal convolution (22 ResBlocks x 4: GroupNorm sums + halo frames together), one all_reduce per
        # TemporalTransformer norm (17), one K|V gather per camera block's attention pair (16) and per second temporal attention
        # (17) + init_attn's (1), one gather of the output: 140 with the camera, fewer without (round 2: ~250)
        assert max(res["collectives per forward"]) <= 140, res["collectives per forward"]
        for name, v in res.items():
            if name in ("ranks agree", "collectives per forward"):
                continue
            l2, mx = v
            if name.startswith("vs reference"):
                tol = (2.5e-2, 5e-2)          # the fixture bound of the unsharded path (test_medium_fixture_tight_tolerance)
            elif name.startswith("25-step"):
                tol = (4.2e-2, 1.0)           # TOL_CAM of tests/test_trajectory_gpu.py (unsharded: 3.4e-2)
            elif name.startswith("layer:"):
                tol = (1e-3, 1e-2)            # exchanges, halos, frame offsets, local mask rows (only GEMM tile choices differ)
            else:
                tol = (3e-2, 6e-2)            # two bf16 evaluations of the whole UNet, see the header
            assert l2 <= tol[0] and mx <= tol[1], f"{name}: rel-L2 {l2:.3e}, max {mx:.3e}"


def _cfg_split_worker(rank, world, port, golden_dir, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from camc2v_amd import camera, parallel
    from oracle.golden_inputs import MEDIUM_CFG, SEED, medium_inputs
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config as inst
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(golden_dir, "unet_medium.npz"))
    tr = np.load(os.path.join(golden_dir, "traj_medium.npz"))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    model = inst({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(MEDIUM_CFG)}, linear_start=0.00085, linear_end=0.012,
        conditioning_key="hybrid", channels=4, image_size=[16, 16], temporal_length=16, add_type="add_to_main_branch",
        pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": {}},
        epipolar_config=dict(origin_h=128, origin_w=128, is_3d_full_attn=False, num_register_tokens=4, attention_resolution=[8, 4, 2, 1],
                             compression_factor=1))})
    model.model.diffusion_model.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    model = model.to(dev).eval()
    inp = medium_inputs()
    g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    packed = camera.epipolar_masks_packed(torch.from_numpy(fx["F128"]).to(dev), 16, 128, 128)
    cam = dict(pluker_embedding_features=[f.to(dev) for f in inp["feats"]], sample_locs_dict=None, sample_locs_packed=packed,
               cond_frame_index=torch.zeros(1, dtype=torch.long, device=dev), add_type="add_to_main_branch")
    torch.manual_seed(int(tr["noise_seed_cam"]))
    zs = [torch.randn(1, 4, 16, 16, 16) for _ in range(25)]

    def sample(use_graph):
        cond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_rep"]], camera_condition=cam)
        uncond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_pf"]])
        samples, inter = model.sample_log(cond, 1, True, 25, unconditional_conditioning=uncond, log_every_t=1, eta=1.0, x_T=inp["x_T"],
                                          unconditional_guidance_scale=7.5, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                          fs=g["fs"], enable_camera_condition=True, injected_noise=zs, use_graph=use_graph)
        return samples.clone(), [x.clone() for x in inter["x_inter"][1:]]

    plain, _ = sample(True)
    model.cfg_split = parallel.CfgSplit()
    eager, xs = sample(False)
    graphed, _ = sample(True)
    del model.cfg_split
    worst = 0.0
    for i, ref in zip(tr["keep_steps"], tr["cam_x_steps"]):
        r_ = torch.from_numpy(ref)
        worst = max(worst, ((xs[int(i)].float().cpu() - r_).norm() / r_.norm()).item())
    parts = [torch.empty_like(eager) for _ in range(world)]
    dist.all_gather(parts, eager)
    res = {"trajectory vs REFERENCE": worst, "graph equals eager": bool(torch.equal(graphed, eager)),
           "ranks agree": float((parts[0] - parts[1]).abs().max().item()),
           "vs unsplit (one 2b forward)": ((eager - plain).norm() / plain.norm()).item()}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_split_over_two_ranks(golden_dir, tmp_path):
    """SURVEY.md section 8e "CFG split": rank 0 runs the conditional forward of every step, rank 1 the unconditional one, one all_gather
    of the noise prediction per step (parallel.CfgSplit).  The 25-step CFG-7.5 trajectory of the reference's own sampler (fixture of
    tests/test_trajectory_gpu.py) must be met within that test's tolerance (4.2e-2; unsplit 3.4e-2) by the eager and by the graph form
    (each rank's forward a hipGraph, exchange + update after every replay), which must agree bit for bit; both ranks end with the
    same latents; against the unsplit sampler (one 2b-batch forward instead of two b-batch ones, i.e. other GEMM tiles: another
    rounding realisation, each 3.3e-2 from the reference after 25 steps) within 6e-2 (measured 4.0e-2)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    mp.spawn(_cfg_split_worker, args=(2, _free_port(), golden_dir, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = json.load(open(tmp_path / f"rank{r}.json"))
        print(f"[parity] CFG split, rank {r}: {res}")
        assert res["graph equals eager"] and res["ranks agree"] == 0.0
        assert res["trajectory vs REFERENCE"] <= 4.2e-2
        assert res["vs unsplit (one 2b forward)"] <= 6e-2
