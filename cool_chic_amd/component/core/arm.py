"""Rate model of the ARM (reference: coolchic/component/core/arm.py:448-485), evaluated on the MI355X.

Only the inference form is provided (no autograd): the encoder's training loop is out of scope, the rate of given
latents under given Laplace parameters is what rate-distortion checks on decoded material need."""
import ctypes as C

import torch

from ..._lib import check, lib


def compute_rate(x: torch.Tensor, expectation: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """Per-symbol rate in bits (same shape as the inputs), arm.py:468-485.  Inputs: float32 CUDA tensors."""
    for t in (x, expectation, scale):
        if not t.is_cuda or t.dtype != torch.float32 or t.shape != x.shape:
            raise ValueError("compute_rate needs three float32 CUDA tensors of one shape")
    x, expectation, scale = x.contiguous(), expectation.contiguous(), scale.contiguous()
    rate = torch.empty_like(x)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    check(lib().ccd_compute_rate(x.device.index or 0, C.c_void_p(stream or None), C.c_void_p(x.data_ptr()),
                                 C.c_void_p(expectation.data_ptr()), C.c_void_p(scale.data_ptr()), x.numel(),
                                 C.c_void_p(rate.data_ptr()), None), "ccd_compute_rate")
    return rate


def total_rate_bits(x: torch.Tensor, expectation: torch.Tensor, scale: torch.Tensor) -> float:
    """Sum of compute_rate() in float64 without materialising the per-symbol tensor."""
    x, expectation, scale = x.contiguous(), expectation.contiguous(), scale.contiguous()
    total = torch.zeros(1, dtype=torch.float64, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    check(lib().ccd_compute_rate(x.device.index or 0, C.c_void_p(stream or None), C.c_void_p(x.data_ptr()),
                                 C.c_void_p(expectation.data_ptr()), C.c_void_p(scale.data_ptr()), x.numel(), None,
                                 C.c_void_p(total.data_ptr())), "ccd_compute_rate")
    return float(total.item())
