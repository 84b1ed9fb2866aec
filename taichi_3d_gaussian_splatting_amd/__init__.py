"""MI355X-native rasterisation operator, drop-in for the hot path of
Wenri/taichi_3d_gaussian_splatting.  Importing the package does not
load the HIP library; constructing the operator does, and fails loudly without it."""
from .Camera import CameraInfo  # noqa: F401
from .GaussianPointCloudRasterisation import GaussianPointCloudRasterisation  # noqa: F401
from .controller_stats import ControllerAccumulators  # noqa: F401
