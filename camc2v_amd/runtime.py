"""Autoregressive long-video generation: 16-frame chunks chained on the last generated frame (reference
main/runtime.py:141-346, the loop of ``Image2Video.get_image`` at :260-326, and its pose extension at :193-202).

The reference drives this from its gradio demo with a single-image preprocessor and per-host model metadata; what is
restated here is the part that sits on the hot path's boundary: given a model with ``log_images`` and a batch whose
``video`` / ``RT`` / ``camera_intrinsics`` cover (auto_reg_steps + 1) x video_length frames (or a trajectory to be extended),
run the chunks, feed each chunk's last frame to the next as conditioning frame 0, and write ``step<i>.mp4`` /
``generated.mp4`` / ``ground_truth.mp4`` / ``cond_step<i>.png`` / ``config.txt`` like the reference does.
"""
import copy
import json
import os

import torch

from .video_io import write_png, write_video


def extend_trajectory(c2ws, total_frames):
    """c2ws [t, 4, 4] -> [>= total_frames, 4, 4]: main/runtime.py:193-202 appends new_poses = last @ (last^-1 @ c2ws) until the
    trajectory is long enough.  That product is c2ws again (up to rounding), i.e. every further chunk replays the trajectory from
    its first pose; restated as written."""
    t = c2ws.shape[0]
    if total_frames <= t:
        return c2ws
    poses = [c2ws]
    for _ in range(-(-total_frames // t)):
        last = poses[-1][-1]
        rel = torch.einsum("ik,tkj->tij", torch.linalg.inv(last), c2ws)
        poses.append(torch.einsum("ik,tkj->tij", last, rel))
    return torch.cat(poses, 0)


def _video_uint8(clip):
    """[c, t, h, w] in [-1, 1] -> uint8 [t, h, w, c]."""
    return (((clip.clamp(-1.0, 1.0) + 1.0) / 2.0) * 255).to(torch.uint8).permute(1, 2, 3, 0).cpu()


@torch.no_grad()
def generate_autoregressive(model, batch, auto_reg_steps=0, video_length=16, save_dir=None, cond_frame_index=0, save_fps=8,
                            caption_prefix="", **log_images_kwargs):
    """batch: one un-batched sample with the dataset's keys (video [3, T, H, W], RT [T, 4, 4], camera_intrinsics [T, 3, 3],
    caption, video_path, frame_stride, fps, optional cond_frames [n, 3, H, W] / RT_cond [n, 4, 4] and the precomputed encoder
    outputs); T may be shorter than (auto_reg_steps + 1) * video_length, in which case every chunk re-uses the first
    video_length frames of the video and the poses are extended relative to the last one.  Returns the concatenated clip
    [1, 3, (auto_reg_steps + 1) * video_length, H, W]."""
    dev = model.device
    total = video_length * (auto_reg_steps + 1)
    w2c = batch["RT"].float()
    if w2c.shape[0] < total:
        w2c = torch.linalg.inv(extend_trajectory(torch.linalg.inv(w2c), total))
    kw = dict(log_images_kwargs)
    kw.setdefault("auto_regressive_steps", auto_reg_steps)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "config.txt"), "w") as f:
            f.write(json.dumps({k: v for k, v in kw.items() if isinstance(v, (int, float, str, bool, type(None)))}, indent=4))
    clips, next_first, last_input = [], None, None
    for i in range(auto_reg_steps + 1):
        idx = list(range(i * video_length, (i + 1) * video_length))
        inp = copy.deepcopy({k: v for k, v in batch.items()})
        if max(idx) < inp["video"].shape[1]:
            inp["video"] = inp["video"][:, idx]
            inp["camera_intrinsics"] = inp["camera_intrinsics"][idx]
        else:
            inp["video"] = inp["video"][:, :video_length]
            inp["camera_intrinsics"] = inp["camera_intrinsics"][:video_length]
        inp["RT"] = w2c[idx]
        if next_first is not None:
            inp["video"] = inp["video"].clone()
            inp["video"][:, 0] = next_first          # the previous chunk's last frame conditions this chunk
        if save_dir is not None:
            write_png(os.path.join(save_dir, f"cond_step{i + 1}.png"), _video_uint8(inp["video"][:, :1])[0].permute(2, 0, 1))
        one = {}
        for k, v in inp.items():                      # add the batch dimension the way the reference does (:288-300)
            if torch.is_tensor(v):
                one[k] = v.unsqueeze(0).to(dev) if v.dim() > 0 else v.reshape(1).to(dev)
            elif isinstance(v, str):
                one[k] = [caption_prefix + v if k == "caption" else v]
            else:
                one[k] = v
        one["cond_frame_index"] = torch.full((1,), cond_frame_index, device=dev, dtype=torch.long)
        out = model.log_images(one, cond_frame_index=one["cond_frame_index"].clone(), **kw)
        clip = out["samples"].clamp(-1.0, 1.0).float().cpu()      # 1, c, f, h, w
        next_first = clip[0, :, -1]
        if save_dir is not None:
            write_video(os.path.join(save_dir, f"step{i + 1}.mp4"), _video_uint8(clip[0]), save_fps)
        clips.append(clip)
        last_input = one
    full = clips[0] if len(clips) == 1 else torch.cat(clips, 2)
    if save_dir is not None:
        write_video(os.path.join(save_dir, "generated.mp4"), _video_uint8(full[0]), save_fps)
        write_video(os.path.join(save_dir, "ground_truth.mp4"), _video_uint8(last_input["video"][0].float().cpu()), save_fps)
    return full


__all__ = ["generate_autoregressive", "extend_trajectory"]
