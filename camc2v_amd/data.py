"""Synthetic stand-in for the RealEstate10K test split (reference data/realestate10k.py:17-380): the dataset is not
available here (external download, no network), so the generation harness iterates clips with the SAME batch keys, shapes
and value ranges, built from a seed:

    video [3,t,H,W] in [-1,1] | caption | video_path | fps | frame_stride | RT [t,4,4] world-to-camera |
    camera_data [t,19] (RealEstate10K row: timestamp, fx fy cx cy (normalised), 0 0, 3x4 pose) | camera_intrinsics [t,3,3] (pixels)
    | cond_frames [n,3,H,W] | RT_cond [n,4,4]

plus the outputs of the OpenCLIP encoders the reference runs inside get_batch_input (third-party weights, not part of this
package): caption_emb [77,1024], null_caption_emb [77,1024], image_clip_tokens [1+n,257,1280], uncond_image_clip_tokens
[257,1280].  The camera is the SURVEY.md section 8(d) trajectory (yaw 0.02 rad/frame + translation (0.05, 0, 0.02)/frame), context
frames continue it behind the clip ('random_back'-like)."""
import math

import torch


class SyntheticRealEstate(torch.utils.data.Dataset):
    def __init__(self, num_samples=4, video_length=16, resolution=(256, 256), frame_stride=8, num_additional_cond_frames=2,
                 seed=20230211, exclude_samples=None, **ignored):
        if isinstance(num_additional_cond_frames, (list, tuple)):
            num_additional_cond_frames = num_additional_cond_frames[-1] if len(num_additional_cond_frames) else 0
        self.n, self.t = int(num_samples), int(video_length)
        self.H, self.W = (resolution, resolution) if isinstance(resolution, int) else tuple(resolution)
        self.fs = int(frame_stride[-1] if isinstance(frame_stride, (list, tuple)) else frame_stride)
        self.nc = int(num_additional_cond_frames)
        self.seed = seed
        excl = set(exclude_samples or [])
        self.names = [nm for nm in (f"synthetic_{i:05d}" for i in range(self.n)) if nm not in excl]

    def __len__(self):
        return len(self.names)

    @staticmethod
    def _pose(f, yaw_step=0.02, dx=0.05, dz=0.02):
        a = yaw_step * f
        c2w = torch.eye(4)
        c2w[0, 0], c2w[0, 2], c2w[2, 0], c2w[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        c2w[0, 3], c2w[2, 3] = dx * f, dz * f
        return torch.linalg.inv(c2w)

    def __getitem__(self, i):
        name = self.names[i]
        g = torch.Generator().manual_seed(self.seed + 7919 * int(name.split("_")[1]))
        t, H, W, n = self.t, self.H, self.W, self.nc
        # smooth pseudo-video: low-frequency colour field drifting with the frame index
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        ph = torch.rand(3, 3, generator=g) * 6.28
        frames = []
        for f in range(t + n):
            shift = 0.03 * f
            frames.append(torch.stack([torch.sin(4 * (xx + shift) + ph[c, 0]) * torch.cos(3 * yy + ph[c, 1]) * 0.8 for c in range(3)]))
        vid = torch.stack(frames, 1)                                     # 3, t+n, H, W
        w2c = torch.stack([self._pose(f) for f in range(t)])
        w2c_ctx = torch.stack([self._pose(-(j + 1) * 2.0) for j in range(n)]) if n else None
        K = torch.tensor([[W / 2.0, 0, W / 2.0], [0, H / 2.0, H / 2.0], [0, 0, 1.0]]).repeat(t, 1, 1)
        cam = torch.zeros(t, 19)
        cam[:, 0] = torch.arange(t) * 1000.0
        cam[:, 1:5] = torch.tensor([0.5, 0.5, 0.5, 0.5])
        cam[:, 7:] = w2c[:, :3, :].reshape(t, 12)
        item = {
            "video": vid[:, :t].contiguous(), "caption": f"a synthetic walkthrough, clip {name}", "video_path": f"{name}.mp4",
            "fps": torch.tensor(30.0 / self.fs), "frame_stride": torch.tensor(self.fs), "RT": w2c, "camera_data": cam,
            "camera_intrinsics": K,
            "caption_emb": torch.randn(77, 1024, generator=g), "null_caption_emb": torch.randn(77, 1024, generator=torch.Generator().manual_seed(self.seed)),
            "image_clip_tokens": torch.randn(1 + n, 257, 1280, generator=g),
            "uncond_image_clip_tokens": torch.randn(257, 1280, generator=torch.Generator().manual_seed(self.seed + 1)),
        }
        if n:
            item["cond_frames"] = vid[:, t:].permute(1, 0, 2, 3).contiguous()   # n, 3, H, W
            item["RT_cond"] = w2c_ctx
        return item


def collate(items):
    """Default-collate semantics for the keys above: tensors stacked, strings listed."""
    out = {}
    for k in items[0]:
        v = [it[k] for it in items]
        out[k] = torch.stack(v) if torch.is_tensor(v[0]) else v
    return out


__all__ = ["SyntheticRealEstate", "collate"]
