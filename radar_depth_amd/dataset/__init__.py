"""Host-side mirror of the reference's input staging (dataset/nuscenes_dataset_torch_new.py), GPU-backed."""
from .staging import center_crop_params, stage_val_batch  # noqa: F401
