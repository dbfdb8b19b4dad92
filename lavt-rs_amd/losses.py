"""Training criteria with the reference's names and call shape (losses.py of the reference; selected by `--loss` at train.py:703-713):

    criterion = cross_entropy_loss            # F.cross_entropy(input, target, weight=[0.9, 1.1])       (losses.py:7-11)
    criterion = MultiClassDiceLoss()          # `--loss mc_dice`, the loss of the released lavt_one      (losses.py:38-77)
    loss = criterion(output, target)          # output (B, 2, H, W) logits, target (B, H, W) int64

Both run on liblavt_hip (the fused upsample + loss kernels with an identity upsample); the step harness uses the same kernels directly on
the decoder's low-resolution output (lib._utils.fused_loss / fused_dice_loss) so that the (B, 2, H, W) logits are never materialised.
DiceFocalLoss / DiceBoundaryLoss (ablation criteria) are outside the hot path and raise.
"""
import torch
from torch import nn

from lavt_hip import ops


def _rows(input):
    if input.dim() != 4 or input.shape[1] != 2:
        raise ValueError(f"Invalid input shape, we expect Bx2xHxW. Got: {tuple(input.shape)}")
    B, _, H, W = input.shape
    return ops.transpose_last2(input.reshape(B, 2, H * W), input.dtype).view(B * H * W, 2), B, H, W


def cross_entropy_loss(input, target):
    rows, B, H, W = _rows(input)
    return ops.upsample_cross_entropy(rows, target, B, H, W, H, W, (0.9, 1.1))[0]


class MultiClassDiceLoss(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.eps = 1e-6

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if not torch.is_tensor(input):
            raise TypeError(f"Input type is not a torch.Tensor. Got {type(input)}")
        if not input.shape[-2:] == target.shape[-2:]:
            raise ValueError(f"input and target shapes must be the same. Got: {input.shape}, {target.shape}")
        if not input.device == target.device:
            raise ValueError(f"input and target must be in the same device. Got: {input.device}, {target.device}")
        rows, B, H, W = _rows(input)
        return ops.upsample_dice_loss(rows, target, B, H, W, H, W)[0]


class DiceFocalLoss(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("DiceFocalLoss is an ablation criterion outside the LAVT hot path")


class DiceBoundaryLoss(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("DiceBoundaryLoss is an ablation criterion outside the LAVT hot path")
