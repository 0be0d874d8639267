"""Per-sample outputs of the generation harness in the layout 03_evaluation.py reads (reference
utils/save_video.py:65-157 ``log_evaluation``, utils/save_video.py:234-251 ``prepare_to_log``):

    <save_dir>/<video name>/generated.mp4      the sampled clip
                            ground_truth.mp4   the input clip
                            camera_data.npy    the clip's camera rows (RealEstate10K text-file layout)
                            captions.txt       one caption per line, with the frame stride appended ("..._fs=8.0")
                            context_<j>.png    the extra context frames
                            condition.png      (optional) the conditioning frame

The reference encodes h264 through torchvision.io.write_video (PyAV).  When torchvision / PyAV are importable they are
used the same way; otherwise (this image has neither, nor ffmpeg) the clip is written by a small pure-Python muxer as
Motion-JPEG in an ISO-BMFF (.mp4) container -- 'mp4v' sample entry, objectTypeIndication 0x6C, one JPEG per frame,
q=95 -- which ffmpeg / PyAV / decord / VLC decode like any other mp4.  ``read_mjpeg_mp4`` reads that form back (tests).
"""
import io
import os
import struct
from pathlib import Path

import numpy as np
import torch


# ---- tiny ISO-BMFF muxer ----------------------------------------------------------------------------------------------
def _box(kind, payload):
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind, version, flags, payload):
    return _box(kind, struct.pack(">B3s", version, flags.to_bytes(3, "big")) + payload)


def _descr(tag, payload):
    assert len(payload) < 128
    return bytes([tag, len(payload)]) + payload


def write_mjpeg_mp4(path, frames, fps, quality=95):
    """frames uint8 [T, H, W, 3] (numpy) -> .mp4 with one JPEG sample per frame."""
    from PIL import Image
    frames = np.ascontiguousarray(frames)
    T, H, W, _ = frames.shape
    jpegs = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=quality, subsampling=0)
        jpegs.append(buf.getvalue())
    timescale = 1000 * int(round(fps)) if float(fps).is_integer() else 90000
    delta = int(round(timescale / float(fps)))
    duration = delta * T
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomiso2mp41")
    mdat_payload = b"".join(jpegs)
    mdat = _box(b"mdat", mdat_payload)
    first = len(ftyp) + 8
    offsets, pos = [], first
    for j in jpegs:
        offsets.append(pos)
        pos += len(j)
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">IH", 0x10000, 0x0100) + b"\0" * 10 + matrix
                 + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) + matrix
                 + struct.pack(">II", W << 16, H << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">HH", 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    vmhd = _full(b"vmhd", 0, 1, b"\0" * 8)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    bitrate = int(8 * len(mdat_payload) * float(fps) / max(T, 1))
    dec = _descr(0x04, bytes([0x6C, 0x11]) + (0).to_bytes(3, "big") + struct.pack(">II", bitrate, bitrate))
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 1, 0) + dec + _descr(0x06, b"\x02")))
    entry = (b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HH", W, H) + struct.pack(">II", 0x480000, 0x480000)
             + struct.pack(">I", 0) + struct.pack(">H", 1) + b"\0" * 32 + struct.pack(">Hh", 0x18, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"mp4v", entry))
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, T, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, T) + b"".join(struct.pack(">I", len(j)) for j in jpegs))
    stco = _full(b"stco", 0, 0, struct.pack(">I", T) + b"".join(struct.pack(">I", o) for o in offsets))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    minf = _box(b"minf", vmhd + dinf + stbl)
    mdia = _box(b"mdia", mdhd + hdlr + minf)
    moov = _box(b"moov", mvhd + _box(b"trak", tkhd + mdia))
    with open(path, "wb") as f:
        f.write(ftyp + mdat + moov)


def read_mjpeg_mp4(path):
    """Inverse of write_mjpeg_mp4: -> (frames uint8 [T, H, W, 3], fps)."""
    from PIL import Image
    data = open(path, "rb").read()

    def find(buf, start, end, kind):
        pos = start
        while pos + 8 <= end:
            size, k = struct.unpack(">I4s", buf[pos:pos + 8])
            if k == kind:
                return pos + 8, pos + size
            pos += size
        raise ValueError(f"box {kind!r} not found")

    s, e = find(data, 0, len(data), b"moov")
    s, e = find(data, s, e, b"trak")
    s, e = find(data, s, e, b"mdia")
    ms, _ = find(data, s, e, b"mdhd")
    timescale = struct.unpack(">I", data[ms + 12:ms + 16])[0]
    s, e = find(data, s, e, b"minf")
    s, e = find(data, s, e, b"stbl")
    ts, _ = find(data, s, e, b"stts")
    delta = struct.unpack(">I", data[ts + 12:ts + 16])[0]
    zs, _ = find(data, s, e, b"stsz")
    n = struct.unpack(">I", data[zs + 8:zs + 12])[0]
    sizes = struct.unpack(f">{n}I", data[zs + 12:zs + 12 + 4 * n])
    cs, _ = find(data, s, e, b"stco")
    offs = struct.unpack(f">{n}I", data[cs + 8:cs + 8 + 4 * n])
    frames = [np.asarray(Image.open(io.BytesIO(data[o:o + z])).convert("RGB")) for o, z in zip(offs, sizes)]
    return np.stack(frames), timescale / delta


def write_video(path, frames, fps):
    """frames: uint8 tensor / array [T, H, W, 3].  h264 through torchvision (as the reference) when it is installed, else MJPEG."""
    frames_t = torch.as_tensor(np.asarray(frames)) if not torch.is_tensor(frames) else frames.cpu()
    try:
        import torchvision.io
        torchvision.io.write_video(str(path), frames_t, fps=fps, video_codec="h264", options={"crf": "10"})
        return "h264"
    except Exception:
        write_mjpeg_mp4(str(path), frames_t.numpy(), fps)
        return "mjpeg"


def write_png(path, img):
    """img uint8 [3, H, W] (the reference's torchvision.io.write_png layout) or [H, W, 3]."""
    from PIL import Image
    a = img.cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
    if a.ndim == 3 and a.shape[0] in (1, 3) and a.shape[-1] not in (1, 3):
        a = np.transpose(a, (1, 2, 0))
    Image.fromarray(np.ascontiguousarray(a.squeeze(-1) if a.shape[-1] == 1 else a)).save(str(path), format="PNG")


# ---- batch logs -> files ------------------------------------------------------------------------------------------------
def prepare_to_log(batch_logs, max_images=100000, clamp=True):
    """utils/save_video.py:234-251: first max_images entries of every value, tensors detached to the CPU (the reference's
    clamp is commented out there; ``log_evaluation`` clamps when it converts to uint8)."""
    if batch_logs is None:
        return None
    max_images = max_images if max_images > 0 else 100000
    for key in list(batch_logs):
        v = batch_logs[key]
        if v is None:
            continue
        n = v.shape[0] if hasattr(v, "shape") else len(v)
        v = v[:min(n, max_images)]
        batch_logs[key] = v.detach().cpu() if torch.is_tensor(v) else v
    return batch_logs


def log_evaluation(batch_logs, save_dir, save_fps=10, save_as_gif=False, rescale=False, save_image_condition=False, print_out=False):
    """utils/save_video.py:65-157.  Returns the list of sample directories written."""
    if batch_logs is None or "samples" not in batch_logs:
        return []
    if save_as_gif:
        raise NotImplementedError("gif output is not used by 02_generate_videos.py")
    to8 = lambda v: (((v + 1.0) / 2.0 if rescale else v) * 255).clamp(0, 255).to(torch.uint8)
    samples = to8(batch_logs["samples"].permute(0, 2, 3, 4, 1))          # 'B C T H W -> B T H W C'
    ground_truth = to8(batch_logs["gt_video"].permute(0, 2, 3, 4, 1))
    cond_images = to8(batch_logs["image_condition"].squeeze(2))            # B C H W
    add_cond = to8(batch_logs["cond_frames"]) if batch_logs.get("cond_frames") is not None else None   # B N C H W
    camera_data = batch_logs.get("camera_data")
    captions = batch_logs["condition"]
    names = [str(Path(vp).stem) for vp in batch_logs["video_path"]]
    written = []
    for i, name in enumerate(names):
        d = Path(save_dir) / name
        os.makedirs(d, exist_ok=True)
        write_video(d / "generated.mp4", samples[i], save_fps)
        write_video(d / "ground_truth.mp4", ground_truth[i], save_fps)
        if add_cond is not None:
            for j in range(add_cond.shape[1]):
                write_png(d / f"context_{j}.png", add_cond[i, j])
        if camera_data is not None:
            cd = camera_data[i]
            np.save(d / "camera_data.npy", cd.cpu().numpy() if torch.is_tensor(cd) else np.asarray(cd))
        with open(d / "captions.txt", "w") as f:
            for txt in captions:
                f.write(f"{txt}\n")
        if save_image_condition:
            write_png(d / "condition.png", cond_images[i])
        if print_out:
            print(f"Saved evaluation results for: {name}")
        written.append(str(d))
    return written


__all__ = ["write_video", "write_png", "write_mjpeg_mp4", "read_mjpeg_mp4", "prepare_to_log", "log_evaluation"]
