"""Flow-match Euler scheduler, "Wan" template only (mirror of diffsynth/diffusion/flow_match.py:30-39,132-154).

Host logic: the sigma / timestep tables are CPU fp32 tensors exactly like the reference's.  The per-step
latent update itself runs in ``fairygen_amd.hip.cfg_euler`` (fused with the CFG combine); ``step_scalars``
hands it the (sigma' - sigma) the reference's ``step`` would use.
"""
import torch


class FlowMatchScheduler:
    def __init__(self, template="Wan"):
        if template != "Wan":
            raise NotImplementedError(f"template {template!r}: only the 'Wan' template is on the FairyGen hot path")
        self.num_train_timesteps = 1000
        self.training = False

    @staticmethod
    def set_timesteps_wan(num_inference_steps=100, denoising_strength=1.0, shift=None):
        sigma_min, sigma_max = 0.0, 1.0
        shift = 5 if shift is None else shift
        sigma_start = sigma_min + (sigma_max - sigma_min) * denoising_strength
        sigmas = torch.linspace(sigma_start, sigma_min, num_inference_steps + 1)[:-1]
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        return sigmas, sigmas * 1000

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False, **kwargs):
        if training:
            raise NotImplementedError("training weights are outside the inference hot path")
        self.sigmas, self.timesteps = self.set_timesteps_wan(num_inference_steps, denoising_strength, **kwargs)
        self.training = False

    def _timestep_id(self, timestep):
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.cpu()
        return int(torch.argmin((self.timesteps - timestep).abs()))

    def step_scalars(self, timestep, to_final=False):
        """(sigma, sigma_next) as the fp32 0-dim tensors `step` uses; sigma_next = 0 after the last step."""
        i = self._timestep_id(timestep)
        sigma = self.sigmas[i]
        sigma_next = torch.zeros(()) if (to_final or i + 1 >= len(self.timesteps)) else self.sigmas[i + 1]
        return sigma, sigma_next

    def step(self, model_output, timestep, sample, to_final=False, **kwargs):
        """x + v*(sigma' - sigma) on the HIP kernel (cfg_scale == 1 form)."""
        from . import hip
        sigma, sigma_next = self.step_scalars(timestep, to_final)
        return hip.cfg_euler(sample.contiguous(), model_output.contiguous(), None, 1.0, float(sigma_next - sigma))

    def add_noise(self, original_samples, noise, timestep):
        sigma = self.sigmas[self._timestep_id(timestep)]
        return (1 - sigma) * original_samples + sigma * noise
