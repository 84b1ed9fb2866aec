"""CameraInfo -- argument type of the operator boundary (reference Camera.py:6-11)."""
from dataclasses import dataclass

import torch


@dataclass
class CameraInfo:
    camera_intrinsics: torch.Tensor  # 3x3 matrix
    camera_height: int  # height of the image
    camera_width: int  # width of the image
    camera_id: int  # camera id
