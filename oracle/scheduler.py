"""TEST INFRASTRUCTURE ONLY -- CPU/torch restatement of the two schedulers the reference pipeline drives
(reference models/pipeline_bindyouravatar.py:228, 936-947; infer.py:202, 289) and of its CFG combine (:924-933).

PARITY UNPINNED: ``CogVideoXDDIMScheduler`` / ``CogVideoXDPMScheduler`` belong to diffusers==0.34.0.dev0
(requirements.txt:23), which is not vendored under /root/reference and not installed here; their configuration
(scaled_linear betas 0.00085..0.012, snr_shift_scale 3.0, zero-terminal-SNR rescale, trailing spacing, v_prediction) ships
with the CogVideoX-5B-I2V checkpoint.  The classes below restate the published algorithm expression by expression
(so torch's own type promotion decides every rounding point, exactly as it would inside diffusers); the reference holds
no tests or golden vectors at this boundary.
"""
import torch


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=3.0):
    """float64 table: scaled-linear betas -> cumprod -> SNR shift -> zero terminal SNR rescale."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    s = ac.sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * (s0 / (s0 - sT))
    return s ** 2


def trailing_timesteps(num_train_timesteps, num_inference_steps):
    t = torch.arange(num_train_timesteps, 0, -num_train_timesteps / num_inference_steps)
    return (t.round() - 1).long()


def cfg_combine(noise_pred, guidance_scale):
    """reference :924-933 -- fp32 on the float() of the bf16 prediction, batch order [uncond, cond]."""
    noise_pred = noise_pred.float()
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)


class DDIM:
    def __init__(self, num_train_timesteps=1000, **kw):
        self.T = num_train_timesteps
        self.ac = alphas_cumprod(num_train_timesteps, **kw)
        self.final = torch.tensor(1.0, dtype=torch.float64)

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = trailing_timesteps(self.T, n)
        return self.timesteps

    def step(self, model_output, timestep, sample):
        """eta = 0, v_prediction.  ``model_output`` fp32, ``sample`` bf16 (as in the reference loop)."""
        t = int(timestep)
        prev_t = t - self.T // self.n
        a_t = self.ac[t]
        a_prev = self.ac[prev_t] if prev_t >= 0 else self.final
        beta_t = 1 - a_t
        x0 = (a_t ** 0.5) * sample - (beta_t ** 0.5) * model_output
        a = ((1 - a_prev) / (1 - a_t)) ** 0.5
        b = a_prev ** 0.5 - a_t ** 0.5 * a
        return a * sample + b * x0


class DPM:
    """CogVideoXDPMScheduler: SDE DPM-Solver++ (2M after the first step); the caller supplies the noise draws."""

    def __init__(self, num_train_timesteps=1000, **kw):
        self.T = num_train_timesteps
        self.ac = alphas_cumprod(num_train_timesteps, **kw)
        self.final = torch.tensor(1.0, dtype=torch.float64)

    set_timesteps = DDIM.set_timesteps

    @staticmethod
    def variables(a_t, a_prev, a_back):
        lamb = ((a_t / (1 - a_t)) ** 0.5).log()
        lamb_next = ((a_prev / (1 - a_prev)) ** 0.5).log()
        h = lamb_next - lamb
        r = None
        if a_back is not None:
            lamb_prev = ((a_back / (1 - a_back)) ** 0.5).log()
            r = (lamb - lamb_prev) / h
        return h, r

    @staticmethod
    def mult(h, r, a_t, a_prev, a_back):
        m1 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
        m2 = (-2 * h).expm1() * a_prev ** 0.5
        if a_back is not None:
            return m1, m2, 1 + 1 / (2 * r), 1 / (2 * r)
        return m1, m2

    def step(self, model_output, old_x0, timestep, timestep_back, sample, noise):
        """-> (prev_sample, x0).  ``noise``: the draw that reaches the returned sample (bf16, like ``sample``)."""
        t = int(timestep)
        prev_t = t - self.T // self.n
        a_t = self.ac[t]
        a_prev = self.ac[prev_t] if prev_t >= 0 else self.final
        a_back = self.ac[int(timestep_back)] if timestep_back is not None else None
        x0 = (a_t ** 0.5) * sample - ((1 - a_t) ** 0.5) * model_output
        h, r = self.variables(a_t, a_prev, a_back)
        m = self.mult(h, r, a_t, a_prev, a_back)
        m_noise = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        if old_x0 is None or prev_t < 0:
            return m[0] * sample - m[1] * x0 + m_noise * noise, x0
        d = m[2] * x0 - m[3] * old_x0
        return m[0] * sample - m[1] * d + m_noise * noise, x0
