"""Generation harness: the eval_config.yaml that 02_generate_videos.py writes -> per-sample output directories, without
PyTorch-Lightning (reference flow: CamContextI2V/02_generate_videos.py:197-355 -> main/trainer.py:80,146-194 -> ImageLogger
(main/callbacks.py:163-196, 238-245) -> model.log_images -> utils/save_video.py:65-157).

    python generate.py <eval_config.yaml> [--out DIR] [--num-samples N] [--synthetic-data] [--random-init] [--no-graph] [--lanes L] [--seed S]

What is honoured from the yaml: ``model`` (target / params / pretrained_checkpoint), ``data.params.{batch_size, test,
test_max_n_samples}``, ``lightning.callbacks.batch_logger.params.{log_images_kwargs, test_directory}``.  One process per GPU
(RANK / WORLD_SIZE from torchrun): the test set is sharded over ranks like Lightning's DistributedSampler does and every rank
writes its own samples (``log_all_gpus: True`` in the reference's eval config); no collective is needed.

Randomness: every test batch draws (start latent, per-step DDIM noise at eta = 1, first-stage posterior sample) from a generator
of its own seeded from ``--seed`` (default 20230211 = the reference's, main/trainer.py:21,62) and the dataset index of the
batch's first sample, so a sample's video does not depend on the lane, the rank, the world size or the run that produced it.
"""
import argparse
import logging
import os
import threading
import time

import torch
import yaml

from . import rng
from .checkpoint import load_checkpoints
from .config import instantiate_from_config
from .data import SyntheticRealEstate, collate
from .video_io import log_evaluation, prepare_to_log

log = logging.getLogger("mainlogger")


def build_model(cfg, device, random_init=False, report=None, encoders=False):
    """``encoders``: also build the OpenCLIP text / image embedders (their weights are the ``cond_stage_model.*`` / ``embedder.*``
    slices of the checkpoint); needed when the batches carry captions and frames only -- the synthetic dataset carries the
    embedders' outputs."""
    model = instantiate_from_config(cfg["model"])
    model.build_feeders(encoders=encoders)
    ckpt = cfg["model"].get("pretrained_checkpoint")
    if ckpt and os.path.exists(ckpt):
        load_checkpoints(model, cfg["model"], report)
        return model.to(device).eval()
    if not random_init:
        raise FileNotFoundError(f"pretrained_checkpoint {ckpt!r} not found (pass --random-init to sample from seeded random weights)")
    log.warning("no checkpoint at %r: seeded random weights (smoke / benchmark mode)", ckpt)
    model = model.to(device)
    g = torch.Generator(device=device).manual_seed(20230211)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.normal_(0.0, 0.02, generator=g)
            if p.dim() == 1 and name.endswith(".weight"):
                p.add_(1.0)
    unet = model.model.diffusion_model
    if hasattr(unet, "invalidate_all"):
        unet.invalidate_all()          # in-place parameter edits are invisible to the packed-operand caches
    for m in model.modules():
        if hasattr(m, "invalidate") and m is not unet:
            m.invalidate()
    return model.eval()


def build_dataset(cfg, synthetic=False, num_samples=None):
    dcfg = cfg.get("data", {}).get("params", {})
    test = dcfg.get("test") or dcfg.get("validation") or {}
    params = dict(test.get("params", {}))
    n = num_samples if num_samples is not None else dcfg.get("test_max_n_samples") or dcfg.get("validation_max_n_samples") or 4
    ds = None
    if not synthetic and test.get("target") and os.path.isdir(str(params.get("data_dir", ""))):
        try:
            ds = instantiate_from_config(test)
        except Exception as e:      # the reference's dataset class needs decord + the RealEstate10K files
            log.warning("could not build %s (%s): falling back to synthetic clips", test.get("target"), e)
    if ds is None:
        ds = SyntheticRealEstate(num_samples=n, **{k: v for k, v in params.items()
                                                    if k in ("video_length", "resolution", "frame_stride", "num_additional_cond_frames",
                                                             "exclude_samples")})
    return ds, int(dcfg.get("batch_size", 1)), int(n)


def generate(cfg, out_dir=None, synthetic=False, num_samples=None, random_init=False, use_graph=True, device=None, lanes=1, encoders=None,
             seed=20230211):
    """Returns the list of sample directories written by this rank."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if device is None:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.set_grad_enabled(False)
    ds, batch_size, n = build_dataset(cfg, synthetic, num_samples)
    model = build_model(cfg, device, random_init, encoders=encoders if encoders is not None else not isinstance(ds, SyntheticRealEstate))
    logger = cfg.get("lightning", {}).get("callbacks", {}).get("batch_logger", {}).get("params", {})
    kw = dict(logger.get("log_images_kwargs") or {})
    save_dir = out_dir or logger.get("test_directory") or os.path.join("results", "test")
    os.makedirs(save_dir, exist_ok=True)
    idx = list(range(min(n, len(ds))))[rank::world]
    starts = list(range(0, len(idx), batch_size))
    written, t0 = [], time.perf_counter()

    def one_batch(s):
        batch = collate([ds[i] for i in idx[s:s + batch_size]])
        with rng.seeded(int(seed) * 1000003 + idx[s], device):      # this batch's own stream of draws (module docstring)
            logs = model.log_images(batch, split="test", use_graph=use_graph, **kw)
        logs = prepare_to_log(logs, -1, True)
        return log_evaluation(logs, save_dir, save_fps=7, rescale=True, print_out=(rank == 0))

    lanes = max(1, min(int(lanes), len(starts)))
    if lanes == 1:
        for s in starts:
            written += one_batch(s)
    else:
        # `lanes` batches in flight: one host thread + HIP stream each (own hipGraphs and static buffers per stream, shared
        # weights); the second batch fills the CUs the first one's small-latent layers leave idle
        from . import ops
        prev_hint = ops.set_streams_in_flight(lanes)
        lock, nxt, done, errors = threading.Lock(), [0], {}, []

        def lane(l, stream):
            try:
                torch.cuda.set_device(device)
                with torch.no_grad(), torch.cuda.stream(stream):
                    while True:
                        with lock:
                            j = nxt[0]
                            nxt[0] += 1
                        if j >= len(starts):
                            break
                        done[j] = one_batch(starts[j])
                    stream.synchronize()
            except BaseException as e:      # re-raised below, on the main thread
                errors.append(e)

        threads = [threading.Thread(target=lane, args=(l, torch.cuda.Stream(device))) for l in range(lanes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        ops.set_streams_in_flight(prev_hint)
        if errors:
            raise errors[0]
        for j in sorted(done):
            written += done[j]
    dt = time.perf_counter() - t0
    log.info("[rank %d] %d clips in %.1f s -> %s", rank, len(written), dt, save_dir)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\\n")[0])
    ap.add_argument("config", help="eval_config.yaml written by 02_generate_videos.py (or any config with a `model:` section)")
    ap.add_argument("--out", default=None, help="output directory (default: batch_logger.test_directory)")
    ap.add_argument("--num-samples", type=int, default=None)
    ap.add_argument("--synthetic-data", action="store_true", help="iterate synthetic clips even if the dataset directory exists")
    ap.add_argument("--random-init", action="store_true", help="sample from seeded random weights when the checkpoint is absent")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--encoders", action="store_true", help="build the OpenCLIP embedders even for the synthetic dataset")
    ap.add_argument("--lanes", type=int, default=2, help="batches in flight per GPU (host thread + HIP stream each)")
    ap.add_argument("--seed", type=int, default=20230211, help="base seed; every batch draws from a generator seeded with (seed, dataset index)")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s: %(message)s")
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    written = generate(cfg, args.out, args.synthetic_data, args.num_samples, args.random_init, not args.no_graph, lanes=args.lanes, encoders=True if args.encoders else None,
                       seed=args.seed)
    print(f"wrote {len(written)} sample directories")
    return 0
