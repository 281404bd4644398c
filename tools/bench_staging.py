"""Throughput of the input-staging kernel (rd_stage_frames) at BASELINE.json's batch geometry vs the HBM roofline:
algorithmic bytes = 7 B read + 20 B written per output pixel."""
import sys
import torch
sys.path.insert(0, ".")
from radar_depth_amd.dataset import stage_val_batch
for B, H0, W0 in [(16, 450, 800), (16, 900, 1600), (128, 450, 800)]:
    img = torch.randint(0, 256, (B, H0, W0, 3), dtype=torch.uint8, device="cuda")
    lid = torch.randint(0, 20000, (B, H0, W0), dtype=torch.int16, device="cuda")
    rad = torch.randint(0, 20000, (B, H0, W0), dtype=torch.int16, device="cuda")
    for _ in range(3): stage_val_batch(img, lid, rad, (450, 800), 80.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): stage_val_batch(img, lid, rad, (450, 800), 80.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = B * 450 * 800 * 27
    print("B=%d frames %dx%d -> 450x800: %.1f us/batch, %.0f frames/s, %.2f TB/s algorithmic (HBM peak 8, ~6.3 achievable)" % (
        B, H0, W0, us, B / us * 1e6, by / us / 1e6))
