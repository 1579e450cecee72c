"""GaussianDiffusion -- MI355X-native drop-in for the reference class of the same name
(denoising_diffusion_pytorch/video_denoising_diffusion_pytorch.py:841-1067, "vddp.py").

Same constructor signature, attributes, registered buffers and method names, so
``main.py:82-91`` and ``Trainer`` (vddp.py:1449-1481, 1622, 1734, 1826) can use it unchanged.
All per-element arithmetic runs in libvmm_hip.so; the 256-step ancestral sampler replays
one captured hipGraph per step (denoiser at batch 2B for classifier-free guidance,
x0 prediction, exact radix-select quantile, posterior update) with no host sync inside.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from . import _native as N
from . import hostmath
from .plan import Q_STRIDE, _stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def normalize_img(t):
    return lincomb(t, a=2.0, d=-1.0)  # vddp.py:1109


def unnormalize_img(t):
    return lincomb(t, a=0.5, d=0.5)  # (t + 1) * 0.5, vddp.py:1112


def lincomb(x, y=None, z=None, a=1.0, b=0.0, c=0.0, d=0.0, out=None):
    out = torch.empty_like(x) if out is None else out
    N.check(N.lib().vmm_lincomb(_ptr(x), _ptr(y), _ptr(z), a, b, c, d, _ptr(out), x.numel(), _stream()), "vmm_lincomb")
    return out


class GaussianDiffusion(nn.Module):
    def __init__(
        self,
        denoise_fn,
        *,
        image_size,
        num_frames,
        channels=4,
        timesteps=1000,
        loss_type="l1",
        use_dynamic_thres=False,
        dynamic_thres_percentile=0.9,
        sampling_timesteps=1000,
        ddim_sampling_eta=0.0,
    ):
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.num_frames = num_frames
        self.denoise_fn = denoise_fn
        for name, val in hostmath.schedule_buffers(timesteps).items():  # float64 -> fp32, vddp.py:862-900
            self.register_buffer(name, val)
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        self.use_dynamic_thres = use_dynamic_thres
        self.dynamic_thres_percentile = dynamic_thres_percentile
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps  # vddp.py:910
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self._graph_cache = {}
        self.graph_cache_size = 6  # captured steps kept (LRU); bench.py's guidance sweep needs five
        self.use_graph = True
        self.register_load_state_dict_pre_hook(GaussianDiffusion._ckpt_pre_hook)

    @staticmethod
    def _ckpt_pre_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """A checkpoint written from the DDP / Accelerate-wrapped model carries `module.` in front of every key
        (`module.betas`, `module.denoise_fn...`); Trainer.load (vddp.py:1577,1587) hands it to this class unchanged."""
        wrapped = prefix + "module."
        for k in [k for k in state_dict if k.startswith(wrapped)]:
            state_dict[prefix + k[len(wrapped):]] = state_dict.pop(k)

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_graph_cache"] = {}
        return st

    # ------------------------------------------------------------------ elementwise pieces (vddp.py:914-933, 1036-1042)
    def q_mean_variance(self, x_start, t):
        shp = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
        mean = self.q_sample(x_start, t, noise=torch.zeros_like(x_start))
        return mean, (1.0 - self.alphas_cumprod).gather(-1, t).reshape(shp), self.log_one_minus_alphas_cumprod.gather(-1, t).reshape(shp)

    def q_sample(self, x_start, t, noise=None, _normalize=False):
        noise = torch.randn_like(x_start) if noise is None else noise
        x_start, noise = x_start.contiguous(), noise.contiguous()
        out = torch.empty_like(x_start)
        B = x_start.shape[0]
        N.check(N.lib().vmm_q_sample(_ptr(x_start), _ptr(noise), _ptr(t), _ptr(self.sqrt_alphas_cumprod), _ptr(self.sqrt_one_minus_alphas_cumprod),
                                     1 if _normalize else 0, _ptr(out), B, x_start.numel() // B, _stream()), "vmm_q_sample")
        return out

    def _predict_x0(self, x_t, t, eps_c, eps_n, w, want_abs):
        x_t = x_t.contiguous()
        B = x_t.shape[0]
        x0 = torch.empty_like(x_t)
        ax0 = torch.empty_like(x_t) if want_abs else None
        N.check(N.lib().vmm_predict_x0(_ptr(x_t), _ptr(eps_c), _ptr(eps_n), float(w), _ptr(t), _ptr(self.sqrt_recip_alphas_cumprod),
                                       _ptr(self.sqrt_recipm1_alphas_cumprod), _ptr(x0), _ptr(ax0), B, x_t.numel() // B, _stream()), "vmm_predict_x0")
        return x0, ax0

    def predict_start_from_noise(self, x_t, t, noise):
        return self._predict_x0(x_t, t, noise.contiguous(), None, 1.0, False)[0]

    def _posterior(self, x0, x_t, t, noise, s, clip_mode):
        B = x_t.shape[0]
        out = torch.empty_like(x_t)
        N.check(N.lib().vmm_posterior_step(_ptr(x0), _ptr(x_t), _ptr(noise), _ptr(s), _ptr(t), _ptr(self.posterior_mean_coef1),
                                           _ptr(self.posterior_mean_coef2), _ptr(self.posterior_log_variance_clipped), clip_mode, _ptr(out), B,
                                           x_t.numel() // B, _stream()), "vmm_posterior_step")
        return out

    def q_posterior(self, x_start, x_t, t):
        shp = (x_t.shape[0],) + (1,) * (x_t.dim() - 1)
        mean = self._posterior(x_start.contiguous(), x_t.contiguous(), t, None, None, 0)
        return mean, self.posterior_variance.gather(-1, t).reshape(shp), self.posterior_log_variance_clipped.gather(-1, t).reshape(shp)

    def _quantile(self, absx):
        B = absx.shape[0]
        n = absx.numel() // B
        k_lo, frac = hostmath.quantile_rank(n, self.dynamic_thres_percentile)
        s = torch.empty(B, dtype=torch.float32, device=absx.device)
        scratch = torch.empty(B * Q_STRIDE, dtype=torch.int32, device=absx.device)
        N.check(N.lib().vmm_quantile_rows(_ptr(absx), B, n, k_lo, frac, 1.0, _ptr(s), _ptr(scratch), _stream()), "vmm_quantile_rows")
        return s

    # ------------------------------------------------------------------ sampling (vddp.py:935-1018)
    def _eps_pair(self, x, t, cond, guidance_scale):
        if guidance_scale == 1:
            return self.denoise_fn.forward(x, t, cond=cond, null_cond_prob=0.0), None
        return self.denoise_fn.guided_pair(x, t, cond)

    def p_mean_variance(self, x, t, clip_denoised: bool, cond=None, guidance_scale=1.0):
        x = x.contiguous()
        eps_c, eps_n = self._eps_pair(x, t, cond, guidance_scale)
        dyn = clip_denoised and self.use_dynamic_thres
        x0, ax0 = self._predict_x0(x, t, eps_c, eps_n, guidance_scale, dyn)
        s = self._quantile(ax0) if dyn else None  # s = max(quantile(|x0|, p), 1)  (vddp.py:941-947)
        mean = self._posterior(x0, x, t, None, s, 2 if dyn else (1 if clip_denoised else 0))
        shp = (x.shape[0],) + (1,) * (x.dim() - 1)
        return mean, self.posterior_variance.gather(-1, t).reshape(shp), self.posterior_log_variance_clipped.gather(-1, t).reshape(shp)

    @torch.inference_mode()
    def p_sample(self, x, t, cond=None, clip_denoised=True, guidance_scale=1.0, noise=None):
        """One ancestral step (vddp.py:956-963).  `noise` may be injected for parity tests; default = device RNG."""
        x = x.contiguous()
        eps_c, eps_n = self._eps_pair(x, t, cond, guidance_scale)
        dyn = clip_denoised and self.use_dynamic_thres
        x0, ax0 = self._predict_x0(x, t, eps_c, eps_n, guidance_scale, dyn)
        s = self._quantile(ax0) if dyn else None
        noise = torch.randn_like(x) if noise is None else noise.contiguous()
        return self._posterior(x0, x, t, noise, s, 2 if dyn else (1 if clip_denoised else 0))

    @torch.inference_mode()
    def p_sample_loop(self, shape, cond=None, guidance_scale=1.0, noises=None, x_T=None):
        """vddp.py:965-975.  `x_T` / `noises` (one tensor per step, order t = T-1 .. 0) may be injected for parity tests."""
        device = self.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device).clone()
        if cond is not None:
            cond = cond.to(device).contiguous()
        stepper = None
        was_static = getattr(self.denoise_fn, "static_weights", False)
        # the optimiser, an EMA update or load_state_dict may have changed the parameters since the last sample(): re-pack the
        # cached plans' operand layouts ONCE (static addresses, so captured graphs stay valid), then skip the per-step scan
        self.denoise_fn.refresh_plans()
        self.denoise_fn.static_weights = True  # weights cannot change inside the sampling loop
        try:
            per_sample = 1
            for v in shape[1:]:
                per_sample *= int(v)
            # (the captured step's elementwise kernels move 16 bytes per thread: samples of a multiple of four elements; others take the eager step)
            if self.use_graph and cond is not None and per_sample % 4 == 0 and (noises is None or self.use_graph == "inject"):
                # (guidance_scale == 1: a B-row plan, the conditional branch alone -- vddp.py:715-728; use_graph = "inject": the captured step with
                # the caller's noise tensors instead of the in-kernel generator, for parity tests against golden loops)
                stepper = self._graphed_step(tuple(shape), cond, float(guidance_scale), inject=noises is not None)
                if self.use_graph == "eager" and not stepper.captured:
                    stepper.captured = True  # the step's launch list without the capture (same kernels, same in-kernel noise)
                stepper.reseed()
            for j, i in enumerate(reversed(range(0, self.num_timesteps))):
                if stepper is not None:
                    if noises is not None:
                        stepper.set_noise(noises[j])
                    img = stepper(img, i)
                else:
                    t = torch.full((b,), i, device=device, dtype=torch.long)
                    img = self.p_sample(img, t, cond=cond, guidance_scale=guidance_scale, noise=None if noises is None else noises[j].to(device))
        finally:
            self.denoise_fn.static_weights = was_static
        return unnormalize_img(img)

    @torch.inference_mode()
    def sample(self, cond=None, batch_size=16, guidance_scale=1.0):
        batch_size = cond.shape[0] if cond is not None else batch_size
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        return sample_fn((batch_size, self.channels, self.num_frames, self.image_size, self.image_size), cond=cond, guidance_scale=guidance_scale)

    @torch.inference_mode()
    def ddim_sample(self, shape, cond=None, guidance_scale=1.0, noises=None, x_T=None):
        """vddp.py:986-1018 (no clipping / thresholding on this path; INTEGER time list bit-exact)."""
        batch, device, eta = shape[0], self.betas.device, self.ddim_sampling_eta
        pairs = hostmath.ddim_time_pairs(self.num_timesteps, self.sampling_timesteps)
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device).clone()
        from .plan import cfg_combine
        if self.use_graph and noises is None and cond is not None and self.use_graph != "inject" and (img.numel() // batch) % 4 == 0:
            # the captured DDIM step: denoiser + vmm_ddim_step_rng (coefficients of the whole time list in a device table, the next timestep left
            # on the device by the step itself): one hipGraph replay per step, no host arithmetic in the loop
            cond = cond.to(device).contiguous()
            was_static = getattr(self.denoise_fn, "static_weights", False)
            self.denoise_fn.refresh_plans()
            self.denoise_fn.static_weights = True
            try:
                stepper = self._graphed_step(tuple(shape), cond, float(guidance_scale), ddim=True)
                if self.use_graph == "eager" and not stepper.captured:
                    stepper.captured = True
                stepper.reseed()
                for time, time_next in pairs:
                    img = stepper(img, time, nxt=time_next)
            finally:
                self.denoise_fn.static_weights = was_static
            return unnormalize_img(img)
        acp = self.alphas_cumprod.cpu()  # one transfer; the per-step scalars below are evaluated on 0-d fp32 host tensors
        for j, (time, time_next) in enumerate(pairs):
            tt = torch.full((batch,), time, device=device, dtype=torch.long)
            eps_c, eps_n = self._eps_pair(img, tt, cond, guidance_scale)
            eps = eps_c if eps_n is None else cfg_combine(eps_c, eps_n, float(guidance_scale))
            x0 = self.predict_start_from_noise(img, tt, eps)
            if time_next < 0:
                img = x0
                continue
            # scalar coefficients evaluated like the reference (fp32 0-d tensors, vddp.py:1006-1010)
            a, an = acp[time], acp[time_next]
            sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = (1 - an - sigma ** 2).sqrt()
            noise = torch.randn_like(img) if noises is None else noises[j].to(device)
            img = lincomb(x0, eps, noise, a=float(an.sqrt()), b=float(c), c=float(sigma))
        return unnormalize_img(img)

    @torch.inference_mode()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        raise NotImplementedError("interpolate() crashes in the reference for conditional models (SURVEY quirk 9) and is not built")

    # ------------------------------------------------------------------ hipGraph-captured guided sampling step
    def _graphed_step(self, shape, cond, w, inject: bool = False, ddim: bool = False):
        # everything the captured launch list bakes in: shapes, guidance weight, thresholding mode / rank, the denoiser's arithmetic, the sampler
        key = (shape, w, cond.shape[-1], str(cond.device), bool(self.use_dynamic_thres), float(self.dynamic_thres_percentile),
               self.denoise_fn.precision, bool(inject), (self.sampling_timesteps, float(self.ddim_sampling_eta)) if ddim else None)
        st = self._graph_cache.pop(key, None)
        if st is None:
            st = _GraphedStep(self, shape, cond.shape[-1], w, inject=inject, ddim=ddim)
            # a small LRU: every entry owns its img / x0 / noise buffers and an instantiated graph, and a guidance sweep over many scales or shapes
            # would otherwise grow device memory for the life of the model (round-4 advisor).  Evicted entries are freed with their last reference.
            while self._graph_cache and len(self._graph_cache) >= max(1, int(self.graph_cache_size)):  # (a size of 0 or less keeps one entry: the step in use)
                self._graph_cache.pop(next(iter(self._graph_cache)))
        self._graph_cache[key] = st  # (re-inserted: most recently used last)
        st.refresh_weights()
        st.set_cond(cond)
        return st

    # ------------------------------------------------------------------ training (vddp.py:1044-1067)
    def p_losses(self, x_start, t, cond=None, noise=None, **kwargs):
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        x_recon = self.denoise_fn(x_noisy, t, cond=cond, **kwargs)
        from .autograd import noise_loss
        if self.loss_type not in ("l1", "l2"):
            raise NotImplementedError()
        return noise_loss(noise, x_recon, squared=self.loss_type == "l2")

    def forward(self, x, *args, **kwargs):
        b, device, img_size = x.shape[0], x.device, self.image_size
        want = (self.channels, self.num_frames, img_size, img_size)
        if tuple(x.shape[1:]) != want:  # check_shape, vddp.py:1064
            raise ValueError(f"expected input of shape (b, {want[0]}, {want[1]}, {want[2]}, {want[3]}), got {tuple(x.shape)}")
        t = torch.randint(0, self.num_timesteps, (b,), device=device).long()
        x = normalize_img(x.contiguous())
        return self.p_losses(x, t, *args, **kwargs)


class _GraphedStep:
    """One sampling step captured as a hipGraph: img <- p_sample(img, t) (or the DDIM update) entirely on device.
    guidance_scale != 1: both guidance branches as ONE batch of 2B rows (mirrored plan); == 1: the conditional branch alone, B rows.
    inject: the step noise comes from a static buffer the caller fills before every replay (parity tests) instead of the in-kernel generator."""

    def __init__(self, diff: GaussianDiffusion, shape, cond_len: int, w: float, inject: bool = False, ddim: bool = False):
        self.diff, self.shape, self.w = diff, shape, w
        self.guided = w != 1
        self.inject, self.ddim = bool(inject), bool(ddim)
        with torch.inference_mode(False):
            self._alloc(diff, shape, cond_len)

    def _alloc(self, diff, shape, cond_len):
        dev = diff.betas.device
        B = shape[0]
        self.B = B
        self.nb = 2 * B if self.guided else B
        self.img = torch.zeros(shape, device=dev)
        self.t = torch.zeros(B, dtype=torch.long, device=dev)       # the step's timestep (every kernel of the step reads it)
        self.t_src = torch.zeros(B, dtype=torch.long, device=dev)   # where the step takes it from: the previous step left the next one here
        self.t_expect = None                                        # host mirror of t_src (None = unknown)
        self.rng = torch.zeros(2, dtype=torch.long, device=dev)     # [0] = Philox key of the step noise (vmm_posterior_step_rng)
        self.cond2 = torch.zeros(self.nb, cond_len, device=dev)
        self.mask2 = torch.cat([torch.zeros(B, dtype=torch.uint8, device=dev), torch.ones(B, dtype=torch.uint8, device=dev)])[: self.nb]
        self.x0 = torch.empty(shape, device=dev)
        self.ax0 = torch.empty(shape, device=dev)
        self.s = torch.empty(B, device=dev)
        self.noise = torch.zeros(shape, device=dev) if self.inject else None
        self.scratch = torch.empty(B * Q_STRIDE, dtype=torch.int32, device=dev)
        if os.environ.get("VMM_POISON_ARENA"):
            self.x0.fill_(float("nan")); self.ax0.fill_(float("nan")); self.s.fill_(float("nan")); self.scratch.fill_(-1)
        n = self.img.numel() // B
        if n % 4:
            raise NotImplementedError("the captured sampling step takes samples of a multiple of four elements (use_graph = False samples any shape)")
        self.k_lo, self.frac = hostmath.quantile_rank(n, diff.dynamic_thres_percentile)
        if self.ddim:
            # coefficients of every step of the time list, evaluated like the eager path (fp32 0-d tensors, vddp.py:1006-1010), indexed by timestep
            T_ = diff.num_timesteps
            acp = diff.alphas_cumprod.cpu()
            coef = torch.zeros(T_, 4)
            nxt = torch.zeros(T_, dtype=torch.long)
            eta = diff.ddim_sampling_eta
            for time, time_next in hostmath.ddim_time_pairs(T_, diff.sampling_timesteps):
                nxt[time] = time_next
                if time_next < 0:
                    coef[time, 3] = 1.0
                    continue
                a, an = acp[time], acp[time_next]
                sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                coef[time, 0], coef[time, 1], coef[time, 2] = float(an.sqrt()), float(c), float(sigma)
            self.coef, self.next_of = coef.to(dev), nxt.to(dev)
        self.plan = None
        self.refresh_weights()
        self.graph = None
        self.captured = False

    def refresh_weights(self):
        """Re-pack the plan's operand layouts if the parameters changed since they were packed (the packed buffers have static
        addresses, so a captured graph stays valid)."""
        _, _, T, H, W = self.shape
        self.plan = self.diff.denoise_fn.get_plan(self.nb, T, H, W, self.cond2.shape[1], self.img.device, mirrored=self.guided)

    def set_cond(self, cond):
        self.cond2[: self.B].copy_(cond)
        if self.guided:
            self.cond2[self.B:].copy_(cond)
        self.plan.cond_in.copy_(self.cond2)
        self.plan.mask_in.copy_(self.mask2)

    def set_noise(self, noise):
        self.noise.copy_(noise)

    def reseed(self):
        """A new Philox key for the step noise, drawn from torch's device generator (so torch.manual_seed governs it, and no host
        synchronisation): once per sample() call."""
        torch.randint(-2 ** 62, 2 ** 62, (1,), dtype=torch.long, device=self.rng.device, out=self.rng[:1])

    def _body(self):
        d, lib, B = self.diff, N.lib(), self.B
        pl = self.plan
        # kernel nodes only (a same-dtype contiguous copy_ would be a memcpy NODE once captured, and a memset node of this graph was
        # observed to run unordered with its neighbouring kernels); no torch kernels either: inputs in one launch
        flags = (0 if pl.mirrored else 1) if self.guided else 2
        N.check(lib.vmm_step_inputs(_ptr(self.img), _ptr(self.t_src), _ptr(pl.x_in), flags, _ptr(self.t), _ptr(pl.time_in), B,
                                    self.img.numel(), _stream()), "vmm_step_inputs")
        pl.launch()
        n = self.img.numel() // B
        eps_c, eps_n = pl.out[:B], (pl.out[B:] if self.guided else None)
        if self.ddim:
            N.check(lib.vmm_ddim_step_rng(_ptr(self.img), _ptr(eps_c), _ptr(eps_n), self.w, _ptr(self.t), _ptr(d.sqrt_recip_alphas_cumprod),
                                          _ptr(d.sqrt_recipm1_alphas_cumprod), _ptr(self.coef), _ptr(self.next_of), _ptr(self.rng), _ptr(self.img), B, n,
                                          _ptr(self.t_src), _stream()), "vmm_ddim_step_rng")
            return
        dyn = d.use_dynamic_thres
        N.check(lib.vmm_predict_x0(_ptr(self.img), _ptr(eps_c), _ptr(eps_n), self.w, _ptr(self.t), _ptr(d.sqrt_recip_alphas_cumprod),
                                   _ptr(d.sqrt_recipm1_alphas_cumprod), _ptr(self.x0), _ptr(self.ax0) if dyn else None, B, n, _stream()), "vmm_predict_x0")
        if dyn:
            N.check(lib.vmm_quantile_rows(_ptr(self.ax0), B, n, self.k_lo, self.frac, 1.0, _ptr(self.s), _ptr(self.scratch), _stream()), "vmm_quantile_rows")
        if self.inject:  # the caller's noise (the same kernel the eager p_sample runs)
            N.check(lib.vmm_posterior_step(_ptr(self.x0), _ptr(self.img), _ptr(self.noise), _ptr(self.s), _ptr(self.t), _ptr(d.posterior_mean_coef1),
                                           _ptr(d.posterior_mean_coef2), _ptr(d.posterior_log_variance_clipped), 2 if dyn else 1, _ptr(self.img), B, n,
                                           _stream()), "vmm_posterior_step")
            return
        # posterior mean + sigma * noise with the noise generated in the kernel (Philox keyed by self.rng); leaves t - 1 in t_src
        N.check(lib.vmm_posterior_step_rng(_ptr(self.x0), _ptr(self.img), _ptr(self.rng), _ptr(self.s), _ptr(self.t), _ptr(d.posterior_mean_coef1),
                                           _ptr(d.posterior_mean_coef2), _ptr(d.posterior_log_variance_clipped), 2 if dyn else 1, _ptr(self.img), B, n,
                                           _ptr(self.t_src), _stream()), "vmm_posterior_step_rng")

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._body()  # warm-up (also primes the RNG state registration)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self.graph = g

    def __call__(self, img, i: int, nxt: Optional[int] = None):
        if self.img.data_ptr() != img.data_ptr():
            self.img.copy_(img)
        if not self.captured:
            saved = self.img.clone()
            self.t_src.fill_(i)
            self._capture()  # a capture failure is an error, not a reason to go eager silently (use_graph = False is the opt-out)
            self.captured = True
            self.img.copy_(saved)
            self.t_expect = None
        if self.t_expect != i:  # (inside a sampling loop the previous step has already left i here: no launch)
            self.t_src.fill_(i)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()
        # what the step left in t_src: t - 1 (ancestral, in-kernel noise), the time list's next entry (DDIM); nothing with injected noise
        self.t_expect = None if self.inject else (nxt if self.ddim else i - 1)
        if self.t_expect is not None and self.t_expect < 0:
            self.t_expect = None
        return self.img
