"""Adversarial losses of the GAN step (libraries/gan/loss.py:5-31): hinge or non-saturating logistic ("ce") with a
temperature, and the R1 penalty E[|d D(x) / d x|^2] on real images. R1 differentiates the discriminator twice; the HIP
ops under it (libraries/custom_stylegan2/op.py) are differentiable to any order."""
import torch
import torch.nn.functional as F


def _check(kind: str) -> None:
    if kind not in ("hinge", "ce"):
        raise AssertionError(f"{kind} is not supported")


def adv_loss_dis(real: torch.Tensor, fake: torch.Tensor, adv_loss_type: str, tmp: float = 1.0) -> torch.Tensor:
    _check(adv_loss_type)
    if adv_loss_type == "hinge":
        return F.relu(1 - real).mean() + F.relu(1 + fake).mean()
    return F.softplus(-real * tmp).mean() + F.softplus(fake * tmp).mean()


def adv_loss_gen(fake: torch.Tensor, adv_loss_type: str, tmp: float = 1.0) -> torch.Tensor:
    _check(adv_loss_type)
    return -fake.mean() if adv_loss_type == "hinge" else F.softplus(-fake * tmp).mean()


def d_r1_loss(real_pred: torch.Tensor, real_img: torch.Tensor) -> torch.Tensor:
    (grad,) = torch.autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad.pow(2).flatten(1).sum(1).mean()
