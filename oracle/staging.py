"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's deterministic input staging, numpy only.

Follows dataset/nuscenes_dataset_torch_new.py: get_data :191-195 (int16 -> depth / 256.), transform_val :417-455 (float32 cast,
CenterCrop, rgb / 255., ToTensor HWC -> CHW) and :503-512 (radar > max_depth -> 0, cat(rgb, radar)); CenterCrop as in
dataset/transforms.py:347-385.  Pinned by tests/golden/staging.npz, which was produced with the reference's own CenterCrop /
ToTensor classes (tests/golden/make_golden_staging.py).  Imported only by tests."""
import numpy as np


def center_crop(img, size):
    h, w = img.shape[0], img.shape[1]
    th, tw = size
    i, j = int(round((h - th) / 2.)), int(round((w - tw) / 2.))     # transforms.py:358-359
    return img[i:i + th, j:j + tw]


def stage_val_frame(image_u8, lidar_i16, radar_i16, crop_size, max_depth):
    lidar = lidar_i16 / 256.                                           # get_data :193
    radar = radar_i16 / 256.                                           # :195
    rgb = np.array(image_u8).astype(np.float32)                        # transform_val :417
    lidar = np.array(lidar).astype(np.float32)
    radar = np.array(radar).astype(np.float32)
    rgb = center_crop(rgb, crop_size) / 255.                           # :446-447 (float32 / python float stays float32)
    rgb = np.array(rgb).astype(np.float32).transpose(2, 0, 1)          # :450,453 ToTensor: HWC -> CHW
    lidar = center_crop(lidar, crop_size)[None]                        # :448,498
    radar = center_crop(radar, crop_size)[None].copy()                 # :456,499
    md = np.float32(np.inf if max_depth < 0.0 else max_depth)          # main.py:71; torch compares in the tensor's dtype
    radar[radar > md] = 0                                              # :506-507
    return np.concatenate([rgb, radar], 0), lidar                      # :508, labels :528


def stage_val_batch(image_u8, lidar_i16, radar_i16, crop_size=(450, 800), max_depth=-1.0):
    outs = [stage_val_frame(image_u8[b], lidar_i16[b], radar_i16[b], crop_size, max_depth) for b in range(image_u8.shape[0])]
    return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])
