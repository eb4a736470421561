"""`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:22) -> HIP kernel."""
from gaussmart_amd.knn import distCUDA2  # noqa: F401
