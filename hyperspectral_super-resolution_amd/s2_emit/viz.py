"""Display / raster helpers.  Mirrors reference ``s2_emit/viz.py`` names; outside the accelerated
path.  Heavy dependencies are imported lazily."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def load_s2_rgb_u8(s2_path: str, bands=(1, 2, 3)) -> np.ndarray:
    """Loads a 3-band S2 raster and returns (H,W,3) in the file's dtype."""
    import rasterio
    with rasterio.open(s2_path) as src:
        return np.stack([src.read(b) for b in bands], axis=-1)


def resize_s2_rgb_to(s2_rgb: np.ndarray, target_hw: Tuple[int, int]) -> np.ndarray:
    """Area resampling to target (H, W).  Uses OpenCV INTER_AREA like the reference when cv2 is
    installed; for exact integer shrink factors falls back to a block mean (identical result)."""
    H, W = target_hw
    try:
        import cv2
        return cv2.resize(s2_rgb, (W, H), interpolation=cv2.INTER_AREA)
    except ImportError:
        h, w = s2_rgb.shape[:2]
        if h % H or w % W:
            raise
        fy, fx = h // H, w // W
        x = s2_rgb.reshape(H, fy, W, fx, -1).astype(np.float64).mean(axis=(1, 3))
        if np.issubdtype(s2_rgb.dtype, np.integer):
            x = np.rint(x)
        x = x.astype(s2_rgb.dtype)
        return x if s2_rgb.ndim == 3 else x[..., 0]


def show_side_by_side(left: np.ndarray, right: np.ndarray, left_title: str, right_title: str, figsize=(12, 5)):
    """Two images next to each other without axes (display helper; same signature as the reference's)."""
    import matplotlib.pyplot as plt
    fig, axes = plt.subplots(1, 2, figsize=figsize)
    for ax, image, title in zip(axes, (left, right), (left_title, right_title)):
        ax.imshow(image)
        ax.set_title(title)
        ax.set_axis_off()
    fig.tight_layout()
    plt.show()
