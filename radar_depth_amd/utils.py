"""Counterpart of the reference's utils.py for the hot path: the argparse flag surface (utils.py:16-74) and the
learning-rate schedule (utils.py:85-89).  save_checkpoint (utils.py:77-82) keeps the wire format; visualisation helpers are outside the path."""
import argparse
import os
import shutil

import torch

from .model.models import Decoder

MODEL_NAMES = ['resnet18', 'resnet34', 'resnet50', 'resnet18_new', 'resnet18_latefusion', 'resnet18_multistage',
               'resnet18_multistage_uncertainty', 'resnet18_multistage_uncertainty_fixs']


def parse_command(argv=None):
    loss_names = ['l1', 'l2']
    data_names = ['nuscenes']
    sparsifier_names = ["uniform", "lidar_radar", "radar", "radar_filtered", "radar_filtered2"]
    modality_names = ['rgb', 'rgbd', 'd']
    ap = argparse.ArgumentParser(description='Sparse-to-Dense')
    ap.add_argument('--arch', '-a', metavar='ARCH', default='resnet18', choices=MODEL_NAMES)
    ap.add_argument('--data', metavar='DATA', default='nyudepthv2', choices=data_names)
    ap.add_argument('--modality', '-m', metavar='MODALITY', default='rgb', choices=modality_names)
    ap.add_argument('-s', '--num-samples', default=0, type=int, metavar='N')
    ap.add_argument('--max-depth', default=-1.0, type=float, metavar='D')
    ap.add_argument('--sparsifier', metavar='SPARSIFIER', default=None, choices=sparsifier_names)
    ap.add_argument('--decoder', '-d', metavar='DECODER', default='deconv2', choices=Decoder.names)
    ap.add_argument('-j', '--workers', default=10, type=int, metavar='N')
    ap.add_argument('--epochs', default=15, type=int, metavar='N')
    ap.add_argument('-c', '--criterion', metavar='LOSS', default='l1', choices=loss_names)
    ap.add_argument('-b', '--batch-size', default=8, type=int)
    ap.add_argument('--lr', '--learning-rate', default=0.01, type=float, metavar='LR')
    ap.add_argument('--momentum', default=0.9, type=float, metavar='M')
    ap.add_argument('--weight-decay', '--wd', default=1e-4, type=float, metavar='W')
    ap.add_argument('--print-freq', '-p', default=50, type=int, metavar='N')
    ap.add_argument('--resume', default='', type=str, metavar='PATH')
    ap.add_argument('-e', '--evaluate', dest='evaluate', type=str, default='')
    ap.add_argument('--no-pretrain', dest='pretrained', action='store_false')
    ap.add_argument('--no-validation', dest="validation", action='store_false')
    ap.set_defaults(pretrained=True)
    ap.set_defaults(validation=True)
    args = ap.parse_args(argv)
    if args.modality == 'rgb' and args.num_samples != 0:
        print("number of samples is forced to be 0 when input modality is rgb")
        args.num_samples = 0
    if args.modality == 'rgb' and args.max_depth != 0.0:
        print("max depth is forced to be 0.0 when input modality is rgb/rgbd")
        args.max_depth = 0.0
    return args


def save_checkpoint(state, is_best, epoch, output_directory):
    """Reference wire format (utils.py:77-82, main.py:358-374): torch.save of {args, epoch, arch, model_state_dict,
    best_result, optimizer_state_dict} as checkpoint-<epoch>.pth.tar (+ model_best.pth.tar)."""
    checkpoint_filename = os.path.join(output_directory, 'checkpoint-' + str(epoch) + '.pth.tar')
    torch.save(state, checkpoint_filename)
    if is_best:
        shutil.copyfile(checkpoint_filename, os.path.join(output_directory, 'model_best.pth.tar'))
    return checkpoint_filename


def load_checkpoint(path, map_location="cpu"):
    """Read a reference-format .pth.tar (main.py:194-266).  The reference pickles an argparse.Namespace under `args` and an
    evaluation.metrics.Result under `best_result`, so the file needs the full unpickler (torch >= 2.6 defaults to
    weights_only=True, which rejects both) and the reference's module path `evaluation.metrics` must resolve: it is aliased
    to radar_depth_amd.evaluation.metrics (same class name, same attributes) for the duration of the load."""
    import sys
    from . import evaluation as _ev
    from .evaluation import metrics as _metrics
    added = [k for k in ("evaluation", "evaluation.metrics") if k not in sys.modules]
    sys.modules.setdefault("evaluation", _ev)
    sys.modules.setdefault("evaluation.metrics", _metrics)
    try:
        return torch.load(path, map_location=map_location, weights_only=False)
    finally:
        for k in added:
            sys.modules.pop(k, None)


def adjust_learning_rate(optimizer, epoch, lr_init):
    """lr = lr_init * 0.1^(epoch // 5); works on torch optimizers and on radar_depth_amd.main.HipTrainStep."""
    lr = lr_init * (0.1 ** (epoch // 5))
    if hasattr(optimizer, "param_groups"):
        for group in optimizer.param_groups:
            group['lr'] = lr
    else:
        optimizer.set_lr(lr)
    return lr
