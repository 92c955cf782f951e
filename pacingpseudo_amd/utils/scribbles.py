"""Offline scribble tools of the reference, on the GPU: artificial scribbles from full labels
(utils/utils_artificial_scribbles.py:5-35, used for the 29k-slice LVSC set) and end-point erosion that shortens them
(utils/utils_shorten_scribble_length.py:32-75).  The morphology (Zhang-Suen skeletonisation, masked anti-diagonal
dilation, end-point detection) runs in HIP kernels, one workgroup per mask with the image resident in LDS; the
control flow and the return values are the reference's."""
from __future__ import annotations

import math

import numpy as np
import torch

from .._lib import lib, stream_ptr


def _dev():
    return torch.device('cuda', torch.cuda.current_device())


def skeletonize(masks: torch.Tensor) -> torch.Tensor:
    """skimage.morphology.skeletonize on every (H,W) mask of a uint8/bool CUDA tensor (..., H, W); returns uint8."""
    m = (masks != 0).to(torch.uint8).contiguous().clone()
    H, W = m.shape[-2:]
    lib.pp_skeletonize(m.data_ptr(), m.numel() // (H * W), H, W, stream_ptr())
    return m


def generate_scribble_fn(lab, num_classes, ignored_index):
    """(H,W) label map -> (H,W) artificial scribble map: the skeleton of every class region, `ignored_index` elsewhere;
    an image that holds background only gets a 40-pixel anti-diagonal stroke instead of a point
    (utils_artificial_scribbles.py:5-35).  Accepts numpy or torch, returns the same kind."""
    as_numpy = not torch.is_tensor(lab)
    lab_t = torch.as_tensor(np.asarray(lab) if as_numpy else lab, device=_dev()).to(torch.int64)
    lab_oh = torch.stack([(lab_t == c) for c in range(num_classes)]).to(torch.uint8)        # (K,H,W)
    scb_oh = skeletonize(lab_oh) * lab_oh
    scb = torch.full_like(lab_t, ignored_index)
    for c in range(num_classes - 1, -1, -1):                  # argmax over [classes..., ignored]: lowest index wins
        scb[scb_oh[c] != 0] = c
    present = set(torch.unique(scb).tolist())
    if present == {0, ignored_index}:
        seed = scb_oh[0:1].contiguous().clone()
        H, W = seed.shape[-2:]
        lib.pp_dilate_antidiagonal(seed.data_ptr(), lab_oh[0:1].contiguous().data_ptr(), 1, H, W, 40, stream_ptr())
        bg = skeletonize(seed)[0]
        # argmax over [new stroke, 0, ..., 0, ignored]: the pixels of the ORIGINAL point skeleton have ignored == 0
        # (the ignored plane was formed before the stroke replaced channel 0), so they stay background as well
        scb = torch.where((bg != 0) | (scb_oh[0] != 0), torch.zeros_like(scb), torch.full_like(scb, ignored_index))
    return scb.cpu().numpy() if as_numpy else scb


def detect_endpoints(image: torch.Tensor) -> torch.Tensor:
    """(1,1,h,w) 0/1 tensor -> (1,1,h,w) float map of the curve's end points (utils_shorten_scribble_length.py:64-75)."""
    img = (image != 0).to(torch.uint8).to(_dev()).contiguous()
    out = torch.empty_like(img)
    h, w = img.shape[-2:]
    lib.pp_curve_endpoints(img.data_ptr(), out.data_ptr(), img.numel() // (h * w), h, w, stream_ptr())
    return out.to(torch.float32)


def delete_endpoints(image: torch.Tensor, unknown: torch.Tensor, length, ratio) -> None:
    """Erode a scribble from its end points, in place, until ceil(length*ratio) pixels are left; removed pixels are
    marked in `unknown` (utils_shorten_scribble_length.py:32-62; same visiting order: row-major over the end points)."""
    target = math.ceil(length * ratio)
    while True:
        endpoints = detect_endpoints(image).cpu()
        if not endpoints.sum():
            rows, cols = np.where(image[0, 0].cpu().numpy() == 1)
            if len(rows) == 0:
                break
            endpoints[0, 0, rows[0], cols[0]] = 1.
        rows, cols = np.where(endpoints[0, 0].numpy() == 1)
        done = False
        for i, j in zip(rows, cols):
            if float(image.sum()) > target:
                image[0, 0, i, j] = 0.
                unknown[0, 0, i, j] = 1.
            else:
                done = True
                break
        if done:
            break
