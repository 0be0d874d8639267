"""DDIM sampler with classifier-free guidance for the MI355X path.

Same constructor / ``sample`` / ``ddim_sampling`` / ``p_sample_ddim`` signatures and the same
schedule arithmetic as the reference sampler (``lvdm/models/samplers/ddim.py:10-346``,
``lvdm/models/utils_diffusion.py:31-91,147-157``), re-designed for the device:

  * the per-step coefficients live in one device table, so a step needs no host->device scalar
    traffic (the reference rebuilds ``torch.full`` tensors from python floats every step);
  * guidance, the std rescale and the x_{t-1} update are one fused HIP launch pair
    (``ccv_ddim_cfg_step``);
  * when the model offers ``apply_model_pair`` the conditional and unconditional passes run as
    ONE UNet forward on a 2b batch (weights are read once per step instead of twice);
  * the camera dict is shared with the unconditional branch by reference (the reference
    deep-copies the 268 MB mask every step, ddim.py:258-260).
"""
import numpy as np
import torch

from . import ops, rng
from .lib import CcvError


# ---- schedule (host side, float64/float32 exactly as the reference mixes them) ------------------
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        a = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def rescale_zero_terminal_snr(betas):
    """Betas with zero terminal SNR (lvdm/models/utils_diffusion.py:112-144, arXiv 2305.08891 algorithm 1): sqrt(alpha_bar) shifted so
    that the last step is zero and scaled so that the first keeps its value."""
    ab_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    first, last = ab_sqrt[0].copy(), ab_sqrt[-1].copy()
    ab_sqrt = (ab_sqrt - last) * (first / (first - last))
    ab = ab_sqrt ** 2
    alphas = np.concatenate([ab[0:1], ab[1:] / ab[:-1]])
    return 1.0 - alphas


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1
    elif ddim_discr_method == "uniform_trailing":
        c = num_ddpm_timesteps / num_ddim_timesteps
        steps = np.flip(np.round(np.arange(num_ddpm_timesteps, 0, -c))).astype(np.int64) - 1
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps)) ** 2).astype(int) + 1
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps}")
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphacums: fp32 tensor (CPU).  Returns fp32 numpy arrays (sigmas, alphas, alphas_prev): the values
    p_sample_ddim ends up using after its ``torch.full`` rounding (see oracle/ddim_oracle.py)."""
    ac = alphacums.detach().float().cpu()
    a = ac[torch.as_tensor(np.ascontiguousarray(ddim_timesteps))]
    a_prev = torch.tensor([ac[0].item()] + ac[torch.as_tensor(np.ascontiguousarray(ddim_timesteps[:-1]))].tolist(),
                          dtype=torch.float64)
    a64 = a.double()
    sig = eta * torch.sqrt((1 - a_prev) / (1 - a64) * (1 - a64 / a_prev))
    return sig.float().numpy(), a.numpy(), a_prev.float().numpy()


# one capture at a time (captures of different lanes come from different host threads), and no default-generator draw on
# another thread meanwhile: see rng.py
_CAPTURE_LOCK = rng.LOCK


def _unet_of(model):
    return getattr(getattr(model, "model", None), "diffusion_model", None)


def _shared_draw(model, t):
    """Under the CFG split / frame sharding every rank runs the guidance + update on what must be identical latents, so a draw
    (start latent, per-step noise) is rank 0's on every rank: each rank's own generator state may differ (reference-style seeding
    adds the rank; any extra draw on one rank shifts its stream).  A no-op for ordinary (clip-parallel) sampling."""
    unet = _unet_of(model)
    for s in (getattr(model, "cfg_split", None), unet.__dict__.get("frame_shard") if unet is not None else None):
        if s is not None:
            from . import parallel
            return parallel.broadcast_from_first(t, s.group)
    return t


class _StaticTree:
    """Device-resident copy of a conditioning tree (dicts / lists / tuples of tensors and plain values) with stable
    addresses: a hipGraph captured on the copies serves every later clip of the same signature -- ``load`` copies a
    new clip's tensors over the old ones.  Aliasing is part of the signature (the same tensor object in two places
    stays one static tensor: the CFG halves share ``c_concat`` and the camera dict)."""

    def __init__(self, tree):
        self._memo = {}
        self.slots = []           # static tensors in walk order
        self._loaded = None       # (id, version) of the tensors last copied in, per slot
        self.tree = self._build(tree)
        self.signature = self.describe(tree)

    # ---- structure ------------------------------------------------------------------------------------------------
    @staticmethod
    def _children(o):
        """(kind, ordered children) of a container node, or None for a leaf."""
        if isinstance(o, dict):
            return "D", sorted(o.items(), key=lambda kv: str(kv[0]))
        if isinstance(o, ops.MaskPack):
            return "M", [(0, o[0]), (1, o[1]), ("wave_bits", o.wave_bits), ("group_order", o.group_order)]
        if isinstance(o, (list, tuple)):
            return ("L" if isinstance(o, list) else "U"), list(enumerate(o))
        return None

    @classmethod
    def describe(cls, tree):
        """Hashable signature: shapes, dtypes, aliasing pattern and the plain values."""
        order = {}

        def walk(o):
            if torch.is_tensor(o):
                n = order.setdefault(id(o), len(order))
                return ("T", n, tuple(o.shape), str(o.dtype), str(o.device))
            kids = cls._children(o)
            if kids is None:
                return ("V", repr(o))
            return (kids[0],) + tuple((str(k), walk(v)) for k, v in kids[1])
        return walk(tree)

    def _build(self, o):
        if torch.is_tensor(o):
            hit = self._memo.get(id(o))
            if hit is None:
                hit = o.detach().clone(memory_format=torch.contiguous_format)
                hit._ccv_static = True     # UNetModel's input cache then keys on the address, not on the version counter
                self._memo[id(o)] = hit
                self.slots.append(hit)
            return hit
        kids = self._children(o)
        if kids is None:
            return o
        kind, items = kids
        vals = [(k, self._build(v)) for k, v in items]
        if kind == "D":
            return {k: v for k, v in vals}
        if kind == "M":
            d = dict(vals)
            return ops.MaskPack(d[0], d[1], d["wave_bits"], d["group_order"])
        seq = [v for _, v in vals]
        return seq if kind == "L" else tuple(seq)

    # ---- per clip -------------------------------------------------------------------------------------------------
    def load(self, tree):
        """Copy the tensors of ``tree`` (same signature) into the static ones.  Returns True when anything changed."""
        srcs, seen = [], set()

        def walk(o):
            if torch.is_tensor(o):
                if id(o) not in seen:
                    seen.add(id(o))
                    srcs.append(o)
                return
            kids = self._children(o)
            if kids is not None:
                for _, v in kids[1]:
                    walk(v)
        walk(tree)
        assert len(srcs) == len(self.slots)
        stamp = [(id(t), t.data_ptr(), t._version) for t in srcs]
        if stamp == self._loaded:
            return False
        for i, (dst, src) in enumerate(zip(self.slots, srcs)):
            if self._loaded is None or self._loaded[i] != stamp[i]:
                dst.copy_(src, non_blocking=True)
        self._loaded = stamp
        self._pinned = srcs       # keeps id() / data_ptr() of the sources meaningful until the next load
        return True


class _GraphedClip:
    """One CFG DDIM step (batched UNet forward + fused guidance/update) captured into a hipGraph that serves every clip
    of the same signature.

    All conditioning tensors live in static copies (:class:`_StaticTree`).  Two graphs are captured on them:
      * the *prologue* -- the step-invariant work of a clip (context K/V projections of all 16 cross-attention layers,
        Pluecker feature rows, mask packing when bool masks were handed over): the UNet runs in ``inputs_only`` mode,
        filling its input cache with tensors that live in the graph's pool, once per clip;
      * the *step* -- the ~1.1k launches of a CFG step, replayed 25 times; only three tiny device copies (timestep
        row, coefficient row, noise) precede each replay.
    A new clip costs one copy of its conditioning (~0.1 GB) + one prologue replay; only a new *signature* (shapes,
    kwargs, weights generation) captures again.  Graphs are cached on the model."""

    MAX_CACHED = 2      # signatures per stream
    MAX_TOTAL = 6       # graph sets in all (streams x signatures)

    @staticmethod
    def _split(cond, kw):
        kw = dict(kw)
        uc = kw.get("unconditional_conditioning")
        if (isinstance(uc, dict) and kw.get("enable_camera_condition") and isinstance(cond, dict)
                and "camera_condition" in cond):
            # its camera entry is derived from `cond` inside the step (p_sample_ddim writes it, like the reference)
            kw["unconditional_conditioning"] = {k: v for k, v in uc.items() if k != "camera_condition"}
        return {"cond": cond, "kw": kw}

    @classmethod
    def get(cls, sampler, x, cond, stochastic, kw):
        cache = sampler.model.__dict__.setdefault("_ccv_graph_cache", {})
        unet = _unet_of(sampler.model)
        gen = getattr(unet, "weights_generation", 0)
        tree = cls._split(cond, kw)
        # graphs (and their static conditioning buffers) belong to the stream they were captured on: a second clip in flight
        # on another stream (bench.py --lanes, one host thread per lane) gets its own set instead of sharing buffers
        lane = torch.cuda.current_stream(x.device).cuda_stream
        split = getattr(sampler.model, "cfg_split", None)
        key = (gen, tuple(x.shape), stochastic, _StaticTree.describe(tree), lane, None if split is None else split.rank)
        with _CAPTURE_LOCK:
            for k in [k for k in cache if k[0] != gen]:      # graphs captured on weights that were since replaced
                cache.pop(k)
            hit = cache.get(key)
            if hit is None:
                # at most MAX_CACHED signatures per stream and MAX_TOTAL sets in all (a set pins its graphs' memory pool; streams
                # that no longer exist would otherwise keep theirs for ever): oldest of this stream first, then oldest overall
                while sum(1 for k in cache if k[4] == lane) >= cls.MAX_CACHED:
                    cache.pop(next(k for k in cache if k[4] == lane))
                while len(cache) >= cls.MAX_TOTAL:
                    cache.pop(next(iter(cache)))
                hit = cache[key] = cls(sampler, x, tree, stochastic)
        hit.sampler = sampler
        hit.load(tree)
        if isinstance(kw.get("unconditional_conditioning"), dict) and "camera_condition" in hit.static.tree["kw"]["unconditional_conditioning"]:
            uc_cam = dict(cond["camera_condition"])      # what the eager path leaves in the caller's dict (ddim.py:259-260)
            uc_cam["is_uc"] = True
            kw["unconditional_conditioning"]["camera_condition"] = uc_cam
        return hit

    def __init__(self, sampler, x, tree, stochastic):
        dev = x.device
        self.static = _StaticTree(tree)
        self.static.load(tree)
        cond, kw = self.static.tree["cond"], self.static.tree["kw"]
        self.x = torch.empty_like(x)
        self.t = torch.zeros(x.shape[0], dtype=torch.long, device=dev)
        self.coef = torch.zeros(4, dtype=torch.float32, device=dev)
        self.noise = torch.zeros_like(x) if stochastic else None
        self.x.copy_(x)
        self.coef.copy_(sampler.ddim_coef[0])
        unet = _unet_of(sampler.model)

        # CFG split over two ranks (parallel.CfgSplit): the graph holds THIS rank's forward only; the exchange of the two noise
        # predictions and the (two-launch) guidance + update run after every replay
        self.split = getattr(sampler.model, "cfg_split", None)
        guided = kw.get("unconditional_conditioning") is not None and kw.get("unconditional_guidance_scale", 1.0) != 1.0
        if self.split is not None and not guided:
            self.split = None

        def step():
            if self.split is not None:
                skw = {k: v for k, v in kw.items() if k not in ("unconditional_conditioning", "unconditional_guidance_scale", "temperature",
                                                                  "guidance_rescale")}
                uc = sampler._uncond_with_camera(cond, kw["unconditional_conditioning"], skw)
                return sampler._own_half(self.split, self.x, cond, self.t, uc, skw), None
            return sampler.p_sample_ddim(self.x, cond, self.t, index=0, noise=self.noise, coef=self.coef, **kw)
        self._kw = kw

        # warm-up on a side stream: packs weights, primes the allocator
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            warm = step()
        torch.cuda.current_stream().wait_stream(side)
        if self.split is not None:
            # one exchange of the real size before any capture: the backend allocates its staging buffers here (with gloo the first
            # all_gather of a new size hung when it came right after a graph replay; seen in the one-GPU rehearsal of bench.py)
            self.split.exchange(warm[0])
        self.prologue = None
        if unet is not None and hasattr(unet, "inputs_only"):
            unet.forget_inputs(self.static.slots)      # the warm-up's derived tensors live outside any graph pool
            self.prologue = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.prologue, capture_error_mode="thread_local"):   # other lanes keep replaying meanwhile
                with unet.inputs_only():
                    step()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.x_prev, self.pred_x0 = step()
        # the captures baked in the addresses of the UNet's derived inputs and packed weights: pin both
        self.keepalive = []
        if unet is not None:
            cached = getattr(unet, "_inputs", None)
            self.keepalive.append(list(cached.items.values()) if cached is not None else [])
            self.keepalive.append([m.__dict__.get("_pk_cache") for m in unet.modules()])
        if self.prologue is not None:
            self.prologue.replay()     # a capture does not execute: fill the derived inputs for the clip just loaded

    def load(self, tree):
        if self.static.load(tree) and self.prologue is not None:
            self.prologue.replay()

    def run(self, x, t_row, coef_row, noise):
        if self.split is not None:
            self.x.copy_(x)
            self.t.copy_(t_row)
            self.graph.replay()                      # x_prev holds this rank's noise prediction here
            e_c, e_uc = self.split.exchange(self.x_prev)
            if self.noise is not None and noise is None:
                noise = _shared_draw(self.sampler.model, rng.randn(self.x.shape, device=self.x.device))
            temperature = self._kw.get("temperature", 1.0)
            if noise is not None and temperature != 1.0:
                noise = noise * temperature
            return ops.ddim_cfg_step(self.x, e_c, e_uc, noise, coef_row.contiguous(), self._kw.get("unconditional_guidance_scale", 1.0),
                                     self._kw.get("guidance_rescale", 0.0))
        if x.data_ptr() != self.x_prev.data_ptr():
            self.x.copy_(x)
        else:
            self.x.copy_(self.x_prev)
        self.t.copy_(t_row)
        self.coef.copy_(coef_row)
        if self.noise is not None:
            if noise is None:
                rng.normal_(self.noise)
            else:
                self.noise.copy_(noise)
        self.graph.replay()
        return self.x_prev, self.pred_x0


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor):
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        if getattr(self.model, "use_dynamic_rescale", False):      # ddim.py:31-33
            self.ddim_scale_arr = self.model.scale_arr[torch.as_tensor(np.ascontiguousarray(self.ddim_timesteps), device=self.model.scale_arr.device)]
            self.ddim_scale_arr_prev = torch.cat([self.ddim_scale_arr[0:1], self.ddim_scale_arr[:-1]])
        f32 = lambda t: t.detach().clone().float()
        self.register_buffer("betas", f32(self.model.betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(self.model.alphas_cumprod_prev))
        sig, a, a_prev = make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, a_prev
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - torch.from_numpy(a)).numpy()
        # device-resident per-step coefficient rows: (a_t, a_prev, sigma_t, sqrt(1 - a_t))
        table = np.stack([a, a_prev, sig, self.ddim_sqrt_one_minus_alphas], axis=1).astype(np.float32)
        self.register_buffer("ddim_coef", torch.from_numpy(table).contiguous())

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0,
               score_corrector=None, corrector_kwargs=None, verbose=True, schedule_verbose=False, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1.0, unconditional_conditioning=None, precision=None,
               fs=None, timestep_spacing="uniform", guidance_rescale=0.0, **kwargs):
        if conditioning is not None and isinstance(conditioning, dict):
            first = conditioning[list(conditioning.keys())[0]]
            cbs = first.shape[0] if hasattr(first, "shape") else first[0].shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta, verbose=schedule_verbose)
        size = (batch_size, *shape)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  precision=precision, fs=fs, guidance_rescale=guidance_rescale, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1.0, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, verbose=True, precision=None,
                      fs=None, guidance_rescale=0.0, injected_noise=None, use_graph=False, **kwargs):
        """``injected_noise``: optional sequence of per-step N(0,1) tensors used instead of torch.randn
        (parity tests; the reference draws with noise_like, lvdm/common.py:31-34)."""
        if ddim_use_original_steps or timesteps is not None:
            raise NotImplementedError("only the DDIM sub-schedule is supported (ddim_use_original_steps=False)")
        if mask is not None and x0 is None:
            raise ValueError("mask needs x0 (ddim.py:175)")
        # latent edits in front of / inside a step (ddim.py:174-199, 316-326): mask / x0 blending, pasted overlap frames of the
        # autoregressive loop, scene-constrained noise shaping, v / dynamically rescaled schedules -- none is on the path of
        # 02_generate_videos.py; they take the sampler's general (torch) step, eager
        n_overlap = int(kwargs.get("num_overlap", 0)) if kwargs.get("paste_overlap_frames") else 0
        shaping = bool(kwargs.get("noise_shaping"))
        side = (mask is not None or n_overlap > 0 or shaping or bool(kwargs.get("paste_cond_frame"))
                or getattr(self.model, "parameterization", "eps") != "eps" or getattr(self.model, "use_dynamic_rescale", False))
        device = self.model.betas.device
        b = shape[0]
        img = _shared_draw(self.model, rng.randn(shape, device=device)) if x_T is None else x_T.to(device).float().contiguous()
        steps = self.ddim_timesteps
        total = steps.shape[0]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        clean_cond = kwargs.pop("clean_cond", False)
        # every timestep tensor is built up front: no host->device traffic inside the loop
        ts_all = torch.from_numpy(np.ascontiguousarray(np.flip(steps))).to(device=device, dtype=torch.long)
        ts_all = ts_all[:, None].expand(total, b).contiguous()
        step_kw = dict(temperature=temperature, unconditional_guidance_scale=unconditional_guidance_scale,
                       unconditional_conditioning=unconditional_conditioning, fs=fs, guidance_rescale=guidance_rescale,
                       **kwargs)
        graphed = None
        if use_graph and not (callback or img_callback) and not side:
            stochastic = bool(np.any(self.ddim_sigmas != 0.0))
            graphed = _GraphedClip.get(self, img, cond, stochastic, step_kw)
        for i in range(total):
            index = total - i - 1
            z = injected_noise[i].to(device).float().contiguous() if injected_noise is not None else None
            if mask is not None:           # keep the (noised) original where the mask is set (ddim.py:174-181)
                img_orig = x0 if clean_cond else self.model.q_sample(x0, ts_all[i])
                img = img_orig * mask + (1.0 - mask) * img
            if n_overlap > 0:              # the overlap frames follow the noised previous clip (ddim.py:183-189)
                img = img.clone()
                img[:, :, :n_overlap] = self.model.q_sample(cond["origin_z_0"][:, :, :n_overlap], ts_all[i])
            if shaping and int(ts_all[i][0]) >= kwargs["noise_shaping_minimum_timesteps"]:      # ddim.py:191-201
                src = kwargs["scene_frames"] if "scene_frames" in kwargs else cond["origin_z_0"]
                img_orig = self.model.q_sample(src, ts_all[i])
                scene_mask = kwargs["scene_mask"]
                img = img_orig * scene_mask + (1.0 - scene_mask) * img
            if graphed is not None:
                img, pred_x0 = graphed.run(img, ts_all[i], self.ddim_coef[index], z)
                if index % log_every_t == 0 or index == total - 1:
                    intermediates["x_inter"].append(img.clone())
                    intermediates["pred_x0"].append(pred_x0.clone())
                continue
            img, pred_x0 = self.p_sample_ddim(img, cond, ts_all[i], index=index, temperature=temperature,
                                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                                              corrector_kwargs=corrector_kwargs, quantize_denoised=quantize_denoised,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, mask=mask, x0=x0,
                                              fs=fs, guidance_rescale=guidance_rescale, noise=z, **kwargs)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        if graphed is not None:
            img = img.clone()  # detach the result from the graph's static output buffer
        if n_overlap > 0:                  # ddim.py:228-231
            img = img.clone()
            img[:, :, :n_overlap] = cond["origin_z_0"][:, :, :n_overlap]
        if kwargs.get("paste_cond_frame"):
            idx = cond["c_cond_frame_index"]
            bi = torch.arange(img.shape[0], device=device)
            img = img.clone()
            img[bi, :, idx] = cond["origin_z_0"][bi, :, idx]
        return img, intermediates

    def _uncond_with_camera(self, c, unconditional_conditioning, kwargs):
        if kwargs.get("enable_camera_condition") and isinstance(c, dict) and "camera_condition" in c:
            # shared by reference; the marker key the reference sets on its copy is kept (ddim.py:259-260)
            uc_cam = dict(c["camera_condition"])
            uc_cam["is_uc"] = True
            unconditional_conditioning["camera_condition"] = uc_cam
        return unconditional_conditioning

    def _own_half(self, split, x, c, t, unconditional_conditioning, kwargs):
        """This rank's forward of a CFG step under parallel.CfgSplit: rank 0 the conditional, rank 1 the unconditional one."""
        return self.model.apply_model(x, t, c if split.rank == 0 else unconditional_conditioning, **kwargs).float().contiguous()

    def _predict_noise(self, x, c, t, unconditional_guidance_scale, unconditional_conditioning, kwargs):
        """(e_c, e_uc) of one step (ddim.py:252-280); e_uc is None without guidance.  With ``model.cfg_split`` set
        (parallel.CfgSplit, two ranks) this rank runs ONE of the two forwards and the halves are exchanged."""
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.0:
            return self.model.apply_model(x, t, c, **kwargs), None
        if not isinstance(c, (dict, torch.Tensor)):
            raise NotImplementedError
        unconditional_conditioning = self._uncond_with_camera(c, unconditional_conditioning, kwargs)
        camera_cfg = kwargs.get("camera_cfg", 1.0)
        third = kwargs.get("enable_camera_condition") and camera_cfg != 1.0 and isinstance(c, dict)
        split = getattr(self.model, "cfg_split", None)
        if split is not None:
            if third:
                raise NotImplementedError("camera_cfg != 1 (a third forward per step) with the CFG split")
            return split.exchange(self._own_half(split, x, c, t, unconditional_conditioning, kwargs))
        pair = getattr(self.model, "apply_model_pair", None)
        if pair is not None:
            e_c, e_uc = pair(x, t, c, unconditional_conditioning, **kwargs)
        else:
            e_c = self.model.apply_model(x, t, c, **kwargs)
            e_uc = self.model.apply_model(x, t, unconditional_conditioning, **kwargs)
        if third:
            # third forward without the camera (ddim.py:268-280): model_output += (camera_cfg - 1) w (e_c - e_nc); the term
            # goes into the unconditional prediction, so the fused guidance + rescale + update kernel runs unchanged
            scheduler = kwargs.get("camera_cfg_scheduler", "constant")
            if scheduler not in ("constant", "cosine"):
                raise NotImplementedError(f"camera_cfg_scheduler {scheduler!r}")
            c_no_cam = {k: v for k, v in c.items() if k != "camera_condition"}
            e_nc = self.model.apply_model(x, t, c_no_cam, **kwargs)
            e_uc = ops.camera_cfg_fold(e_uc.float().contiguous(), e_c.float().contiguous(), e_nc.float().contiguous(),
                                       (camera_cfg - 1.0) / (1.0 - unconditional_guidance_scale),
                                       t.contiguous() if scheduler == "cosine" else None)
        return e_c, e_uc

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1.0, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, uc_type=None,
                      conditional_guidance_scale_temporal=None, mask=None, x0=None, guidance_rescale=0.0,
                      noise=None, coef=None, **kwargs):
        """``coef``: optional device tensor [4] = (a_t, a_prev, sigma_t, sqrt(1-a_t)) overriding the table row of
        ``index`` (lets a captured hipGraph be replayed for every step)."""
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.0:
            raise NotImplementedError("original-step / quantised / corrected / dropout sampling is not on the hot path")
        x = x.float().contiguous()
        e_c, e_uc = self._predict_noise(x, c, t, unconditional_guidance_scale, unconditional_conditioning, kwargs)
        n_overlap = int(kwargs.get("num_overlap", 0)) if kwargs.get("paste_overlap_frames") else 0
        v_param = getattr(self.model, "parameterization", "eps") == "v"
        dyn = getattr(self.model, "use_dynamic_rescale", False)
        if v_param or dyn or n_overlap > 0 or kwargs.get("paste_cond_frame"):
            return self._general_step(x, c, t, index, e_c, e_uc, unconditional_guidance_scale, guidance_rescale, noise, coef, temperature,
                                      repeat_noise, v_param, dyn, n_overlap, bool(kwargs.get("paste_cond_frame")))
        if noise is None and coef is None and float(self.ddim_sigmas[index]) != 0.0:
            shape = (1, *x.shape[1:]) if repeat_noise else x.shape
            noise = _shared_draw(self.model, rng.randn(shape, device=x.device).expand(x.shape).contiguous())
        if noise is not None and temperature != 1.0:
            noise = noise * temperature
        x_prev, pred_x0 = ops.ddim_cfg_step(x, e_c.float().contiguous(), None if e_uc is None else e_uc.float().contiguous(),
                                            noise, coef if coef is not None else self.ddim_coef[index],
                                            unconditional_guidance_scale, guidance_rescale)
        return x_prev, pred_x0

    def _general_step(self, x, c, t, index, e_c, e_uc, scale, guidance_rescale, noise, coef, temperature, repeat_noise, v_param, dyn,
                      n_overlap, paste_cond):
        """The step of ddim.py:266-346 in plain fp32 tensor arithmetic, for the branches the fused kernel does not cover: v
        parameterisation, dynamic rescale of the predicted x0, frames pasted into the predicted x0 (``paste_cond_frame``,
        ``paste_overlap_frames``).  Guidance, the camera term (already folded into e_uc by ``_predict_noise``) and the std rescale as
        in the reference (utils_diffusion.py:147-157)."""
        e_c = e_c.float()
        out = e_c if e_uc is None else e_uc.float() + scale * (e_c - e_uc.float())
        if e_uc is not None and guidance_rescale > 0.0:
            dims = list(range(1, out.dim()))
            rescaled = out * (e_c.std(dim=dims, keepdim=True) / out.std(dim=dims, keepdim=True))
            out = guidance_rescale * rescaled + (1.0 - guidance_rescale) * out
        row = coef if coef is not None else self.ddim_coef[index]
        a_t, a_prev, sigma_t, sqrt_one_minus_at = row[0], row[1], row[2], row[3]
        if v_param:
            e_t = self.model.predict_eps_from_z_and_v(x, t, out)
            pred_x0 = self.model.predict_start_from_z_and_v(x, t, out)
        else:
            e_t = out
            pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        if dyn:
            pred_x0 = pred_x0 * (self.ddim_scale_arr_prev[index] / self.ddim_scale_arr[index]).to(pred_x0)
        if paste_cond:
            bi = torch.arange(pred_x0.shape[0], device=x.device)
            pred_x0 = pred_x0.clone()
            pred_x0[bi, :, c["c_cond_frame_index"]] = c["origin_z_0"][bi, :, c["c_cond_frame_index"]].to(pred_x0)
        if n_overlap > 0:
            pred_x0 = pred_x0.clone()
            pred_x0[:, :, :n_overlap] = c["origin_z_0"][:, :, :n_overlap].to(pred_x0)
        dir_xt = (1.0 - a_prev - sigma_t ** 2).clamp(min=0).sqrt() * e_t
        if noise is None and float(sigma_t) != 0.0:
            shape = (1, *x.shape[1:]) if repeat_noise else x.shape
            noise = _shared_draw(self.model, rng.randn(shape, device=x.device).expand(x.shape).contiguous())
        x_prev = a_prev.sqrt() * pred_x0 + dir_xt
        if noise is not None:
            x_prev = x_prev + sigma_t * noise * temperature
        return x_prev, pred_x0


__all__ = ["DDIMSampler", "make_beta_schedule", "make_ddim_timesteps", "make_ddim_sampling_parameters", "rescale_zero_terminal_snr", "CcvError"]
