"""Host side of the DDIM scheduler the reference takes from diffusers 0.11.1
(``DDIMScheduler(num_train_timesteps, beta_schedule='squaredcos_cap_v2', clip_sample=True,
prediction_type='epsilon')``, generator/train.py:83; dynamics/trainer.py:36).

Only the tables live here (a few hundred floats, built once); the per-element update is the HIP
kernel behind ``dgdm_ddim_guided_step``.  diffusers is not vendored by the reference and is absent
from this image, so this follows the published algorithm - parity unpinned.

Duck-types what ``generator/diffusion.py`` touches: ``timesteps``, ``alphas_cumprod``,
``config.num_train_timesteps``, ``set_timesteps``, ``step(...).prev_sample``, ``add_noise``.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional, Tuple

import numpy as np
import torch

from . import engine


class DDIMSchedulerOutput:
    def __init__(self, prev_sample: torch.Tensor):
        self.prev_sample = prev_sample


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2", clip_sample: bool = True,
                 prediction_type: str = "epsilon", **unused):
        if beta_schedule != "squaredcos_cap_v2" or not clip_sample or prediction_type != "epsilon":
            raise NotImplementedError("only the configuration generator/train.py:83 uses is implemented")
        T = int(num_train_timesteps)
        self.config = SimpleNamespace(num_train_timesteps=T, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      prediction_type=prediction_type)
        bar = lambda s: math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2      # noqa: E731
        self.betas = torch.tensor([min(1 - bar((i + 1) / T) / bar(i / T), 0.999) for i in range(T)], dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, T)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, self.num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))

    def coefficients(self, t: int) -> Tuple[float, float, float, float]:
        """float32 sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev) as the reference's tensor ops give them."""
        t = int(t)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0, **unused) -> DDIMSchedulerOutput:
        if eta != 0.0:
            raise NotImplementedError("the reference always steps with eta = 0")
        return DDIMSchedulerOutput(engine.ddim_guided_step(sample, model_output, None, 0, self.coefficients(int(timestep)), 0.0))

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        ts = torch.unique(timesteps.detach().cpu())
        if len(ts) != 1:
            raise NotImplementedError("add_noise is used with one shared timestep on the sampling path (generator/diffusion.py:184-189)")
        a = self.alphas_cumprod[int(ts[0])]
        return engine.ddim_add_noise(original_samples, noise, float(a ** 0.5), float((1 - a) ** 0.5))
