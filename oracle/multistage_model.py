"""CPU ORACLE (test infrastructure, NOT the product path).

Restates /root/reference/model/multistage_model.py:
  ResNet_multistage  <- :22-83    two late-fusion stages coupled through a radar filter
  Filter_layer       <- :87-119   |dense - sparse| <= 5 * 3.6^(dense/100)
  ResNet_latefusion2 <- :123-276  late fusion whose depth stem takes in_channels-3 inputs

Differences kept deliberately: no torchvision download for
the inner stages, Filter_layer constants are Python floats.  The `pretrained=True` file
lookup is kept with the reference's error text.
"""
import math
import os

import torch
import torch.nn as nn

from .models import ResNet_latefusion, _DEPTHS, _conv


class ResNet_latefusion2(ResNet_latefusion):
    def _depth_inputs(self):
        return self.in_channels - 3

    def _split(self, x):
        assert x.shape[1] >= 4
        if self.in_channels == 4:
            return x[:, :3], x[:, 3:]
        return x[:, :3], torch.cat((x[:, 3:4], x[:, 4:5]), dim=1)


class Filter_layer(nn.Module):
    ALPHA, BETA, K = 5.0, 18.0, 100.0

    # fp32 constants evaluated exactly as the reference's 0-dim tensors evaluate them (:91-93,98)
    LOG_RATIO = float(torch.log(torch.tensor(BETA) / torch.tensor(ALPHA)))
    LOG_ALPHA = float(torch.log(torch.tensor(ALPHA)))

    def sid_depth_thresh(self, input_depth):
        return torch.exp(((input_depth * self.LOG_RATIO) / self.K) + self.LOG_ALPHA)

    def compute_valid_mask(self, sparse_depth, dense_depth):
        return torch.abs(dense_depth - sparse_depth) <= self.sid_depth_thresh(dense_depth)

    def forward(self, sparse_depth, dense_depth):
        mask = self.compute_valid_mask(sparse_depth, dense_depth).to(torch.float32)
        return sparse_depth * mask, mask


class ResNet_multistage(nn.Module):
    def __init__(self, layers, decoder, output_size, pretrained=True, project_root="YOUR_PATH/radar_depth"):
        if layers not in _DEPTHS:
            raise RuntimeError("Only 18, 34, 50, 101, and 152 layer model are defined for ResNet. Got {}".format(layers))
        super().__init__()
        self.stage1 = ResNet_latefusion2(layers, decoder, output_size, in_channels=4, pretrained=False)
        self.stage2 = ResNet_latefusion2(layers, decoder, output_size, in_channels=5, pretrained=False)
        self.filter_layer = Filter_layer()
        if pretrained is True:
            path = os.path.join(project_root, "pretrained/resnet18_latefusion.pth.tar")
            if not os.path.exists(path):
                raise ValueError("[Error] Can't find pretrained latefusion model. "
                                 "Please follow the instructions in README.md to download the weights!")
            weights = torch.load(path)["model_state_dict"]
            self.stage1.load_state_dict(weights)
            self.stage2.load_state_dict(self.filter_state_dict(weights, self.stage2.state_dict()), strict=False)

    @staticmethod
    def filter_state_dict(pretrain_dict, target_dict):
        return {k: v for k, v in pretrain_dict.items() if target_dict[k].shape == v.shape}

    def forward(self, x):
        rgb, radar = x[:, :3], x[:, 3:]
        depth1 = self.stage1(x)
        radar_kept, mask = self.filter_layer(radar, depth1)
        depth2 = self.stage2(torch.cat((rgb, radar_kept, depth1), dim=1))
        return {"stage1": depth1, "stage2": depth2, "mask": mask, "radar_filtered": radar_kept}
