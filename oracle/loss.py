"""CPU restatement of the six unlearning objectives (oracle; test infrastructure only).

Follows /root/reference/losses/ddpm_deletion_loss.py.  Every function returns the
reference's uniform 7-tuple
``(loss, loss_x, loss_a, iw_x, iw_a, weighted_loss_x, weighted_loss_a)``.

The one deliberate difference: the keep/forget Bernoulli draw can be injected
(``u=`` uniforms in [0,1)) so that HIP and CPU paths see the same mask; with
``u=None`` the draw is ``torch.rand(B)`` from the default CPU generator exactly as
ddpm_deletion_loss.py:18 / :101 do, which is how the golden fixtures were made.
"""
import torch


def _bcast(v, like):
    return v.reshape(-1, *([1] * (like.dim() - 1)))


def _keep_mask(batch, lambd, u):
    # ddpm_deletion_loss.py:18  -- keep-sample with probability 1 - lambd
    if u is None:
        u = torch.rand(batch)
    return u > lambd


def mix(noisy_keep, noisy_forget, keep):
    """ddpm_deletion_loss.py:21-23: row-select the defensive mixture."""
    keep = keep.to(noisy_keep.device)
    return torch.where(_bcast(keep, noisy_keep), noisy_keep, noisy_forget)


def siss_terms(x_mix, x0, a0, gamma_t, sigma_t, lambd):
    """ddpm_deletion_loss.py:26-45: score targets, squared distances, IS weights.

    Returns eps_x, eps_a, dist_x, dist_a, iw_x, iw_a (fp32 promotion as in the
    reference: gamma/sigma are fp32 so bf16 inputs promote)."""
    g = _bcast(gamma_t, x_mix)
    s = _bcast(sigma_t, x_mix)
    rx = x_mix - g * x0
    ra = x_mix - g * a0
    eps_x = rx / s
    eps_a = ra / s
    dims = list(range(1, x_mix.dim()))
    dist_x = (rx ** 2).sum(dim=dims) / (2 * sigma_t ** 2)
    dist_a = (ra ** 2).sum(dim=dims) / (2 * sigma_t ** 2)
    # :41-45 -- may overflow to inf; 1/inf = 0 gives the saturated weights {0, 1/(1-l)}.
    iw_x = 1 / ((1 - lambd) + lambd * torch.exp(dist_x - dist_a))
    iw_a = 1 / ((1 - lambd) * torch.exp(dist_a - dist_x) + lambd)
    return eps_x, eps_a, dist_x, dist_a, iw_x, iw_a


class OracleDeletionLoss:
    """Same surface as the reference's DDPMDeletionLoss (:3-7)."""

    def __init__(self, gamma, sigma):
        self.all_gamma = gamma
        self.all_sigma = sigma

    # :11-56  SISS
    def importance_sampling_with_mixture(self, unet, timesteps, noise, conditioning,
                                         all_samples_dict, deletion_samples_dict, lambd, u=None):
        gt, st = self.all_gamma[timesteps], self.all_sigma[timesteps]
        nk, nf = all_samples_dict["noisy_latents"], deletion_samples_dict["noisy_latents"]
        keep = _keep_mask(nk.shape[0], lambd, u)
        x_mix = mix(nk, nf, keep)
        pred = unet(x_mix, timesteps, **conditioning, return_dict=False)[0]
        eps_x, eps_a, _, _, iw_x, iw_a = siss_terms(
            x_mix, all_samples_dict["og_latents"], deletion_samples_dict["og_latents"], gt, st, lambd)
        loss_x = (pred - eps_x) ** 2
        loss_a = (pred - eps_a) ** 2
        return (None, loss_x, loss_a, iw_x, iw_a,
                _bcast(iw_x, loss_x) * loss_x, _bcast(iw_a, loss_a) * loss_a)

    # :60-67  SISS (No IS): two forwards, plain noise target
    def double_forward_with_neg_del(self, unet, timesteps, noise, conditioning,
                                    all_samples_dict, deletion_samples_dict):
        lx = (unet(all_samples_dict["noisy_latents"], timesteps, **conditioning,
                   return_dict=False)[0] - noise) ** 2
        la = (unet(deletion_samples_dict["noisy_latents"], timesteps, **conditioning,
                   return_dict=False)[0] - noise) ** 2
        return None, lx, la, None, None, lx, la

    # :70-78  EraseDiff: forget target is U[0,1) noise
    def erasediff(self, unet, timesteps, noise, conditioning, all_samples_dict,
                  deletion_samples_dict):
        lx = (unet(all_samples_dict["noisy_latents"], timesteps, **conditioning,
                   return_dict=False)[0] - noise) ** 2
        pa = unet(deletion_samples_dict["noisy_latents"], timesteps, **conditioning,
                  return_dict=False)[0]
        la = (pa - torch.rand_like(pa)) ** 2
        return None, lx, la, None, None, lx, la

    # :82-88  NegGrad
    def simple_neg_del(self, unet, timesteps, noise, conditioning, all_samples_dict,
                       deletion_samples_dict, superfactor):
        la = (unet(deletion_samples_dict["noisy_latents"], timesteps, **conditioning,
                   return_dict=False)[0] - noise) ** 2
        return -superfactor * la, None, la, None, None, None, None

    # :91-96  naive fine-tune on the keep set
    def naive_del(self, unet, timesteps, noise, conditioning, all_samples_dict,
                  deletion_samples_dict):
        lx = (unet(all_samples_dict["noisy_latents"], timesteps, **conditioning,
                   return_dict=False)[0] - noise) ** 2
        return lx, lx, None, None, None, None, None

    # :99-122  Bernoulli sub-score (row selection, zero-size guards)
    def subscore_bernoulli(self, unet, timesteps, noise, conditioning, all_samples_dict,
                           deletion_samples_dict, lambd, u=None):
        nk, nf = all_samples_dict["noisy_latents"], deletion_samples_dict["noisy_latents"]
        keep = _keep_mask(nk.shape[0], lambd, u)
        pred = unet(mix(nk, nf, keep), timesteps, **conditioning, return_dict=False)[0]
        sq = (pred - noise) ** 2
        lx = (1 / (1 - lambd)) * sq[keep]      # python-float division: raises at lambd == 1 like :110
        la = sq[~keep]
        if lx.shape[0] == 0:          # :113-116
            lx = torch.zeros(1, 1, 1, 1, requires_grad=True)
            la = torch.zeros(1, 1, 1, 1, requires_grad=True)
        if la.shape[0] == 0:          # :118-120
            la = torch.zeros(1, 1, 1, 1, requires_grad=True)
        return None, lx, la, None, None, lx, la
