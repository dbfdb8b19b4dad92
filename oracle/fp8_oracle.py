"""CPU restatement of the fp8 (OCP e4m3) operand path of BASELINE.json configs[4] -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench baseline).

The reference (Yxxxb/LAVT-RS) has no fp8 arithmetic: it is a PyTorch fp32 / AMP code base.  What this file pins is therefore the build's own
stated format, so that the HIP kernels can be checked bit-for-bit / to rounding against an independent evaluation:

* quantisation: q = e4m3fn(clamp(x * 448 / amax, +-448)) with round-to-nearest-even (torch.float8_e4m3fn's cast), per-tensor amax
  (amax <= 0 means scale 1);
* contraction: exact products of the e4m3 values, fp32 accumulation, result multiplied by (amax_a / 448) * (amax_b / 448).

The model-level gate of the fp8 configuration is taken against the REFERENCE's fp32 run (tests/golden/full_swin_b_480_b2.npz), like bf16's.
"""
import torch
import torch.nn.functional as F

E4M3_MAX = 448.0


def scale_of(amax: float) -> float:
    return E4M3_MAX / amax if amax > 0 else 1.0


def quantize_bytes(x: torch.Tensor, amax: float) -> torch.Tensor:
    """-> uint8 tensor holding the e4m3fn encodings"""
    return (x.float() * scale_of(amax)).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).view(torch.uint8)


def dequantize(x: torch.Tensor, amax: float) -> torch.Tensor:
    """quantise-dequantise: the values the fp8 contraction effectively multiplies"""
    return (x.float() * scale_of(amax)).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() / scale_of(amax)


def linear_fp8(x, w, b, amax_x, amax_w):
    y = dequantize(x, amax_x) @ dequantize(w, amax_w).t()
    return y + b if b is not None else y


def conv3x3_fp8(x_nchw, w, amax_x, amax_w):
    return F.conv2d(dequantize(x_nchw, amax_x), dequantize(w, amax_w), padding=1)


def conv3x3_fp8_dgrad(dy_nchw, w, amax_dy, amax_w):
    """data gradient of the 3x3 convolution on quantise-dequantised dY and W (the transposed convolution PyTorch's autograd runs for
    F.conv2d(padding=1); reference lib/mask_predictor.py:60-97 backward)"""
    return F.conv_transpose2d(dequantize(dy_nchw, amax_dy), dequantize(w, amax_w), padding=1)



def conv3x3_fp8_wgrad(x_nchw, dy_nchw, amax_x, amax_dy):
    """weight gradient of the 3x3 convolution on quantise-dequantised X and dY (what autograd computes for nn.Conv2d(3x3, padding=1, bias=False);
    reference lib/mask_predictor.py:60-97 backward): dW[co][ci][ky][kx] = sum_p dY[p][co] X[p + (ky - 1, kx - 1)][ci], zero outside the image"""
    xq, dq = dequantize(x_nchw, amax_x), dequantize(dy_nchw, amax_dy)
    return torch.nn.grad.conv2d_weight(xq, (dq.shape[1], xq.shape[1], 3, 3), dq, padding=1)
