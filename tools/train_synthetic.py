#!/usr/bin/env python3
"""The reference's training loop (main.py:268-375, 388-447) on the MI355X-native path, with synthetic batches instead of the
nuScenes loader: parse_command -> create_model -> fused HipTrainStep per batch -> LR schedule per epoch -> on-device metrics ->
checkpoint in the reference's .pth.tar layout (args, epoch, arch, model_state_dict, best_result, optimizer_state_dict) -> --resume.

    python tools/train_synthetic.py -a resnet18_latefusion -d upproj -m rgbd --data nuscenes -b 4 --epochs 2 --no-pretrain \
        --steps-per-epoch 5 --height 225 --width 400 --output /tmp/run1
    python tools/train_synthetic.py ... --resume /tmp/run1/checkpoint-1.pth.tar
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radar_depth_amd import utils  # noqa: E402
from radar_depth_amd.evaluation.metrics import AverageMeter, Result  # noqa: E402
from radar_depth_amd.main import HipInference, HipTrainStep, create_model  # noqa: E402
from radar_depth_amd.synthetic import make_batch  # noqa: E402


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    # --operands split|fp32|bf16: the convolution arithmetic of the fused step (default "split": what bench.py measures; "fp32": every
    # convolution on the fp32 MFMA, the arithmetic of the eager autograd path)
    extra = {"--steps-per-epoch": 5, "--height": 450, "--width": 800, "--output": "/tmp/radar_depth_run", "--storage": "fp32", "--operands": "split"}
    for key in list(extra):                      # options of this script, stripped before the reference's parser sees argv
        if key in argv:
            i = argv.index(key)
            extra[key] = type(extra[key])(argv[i + 1])
            del argv[i:i + 2]
    args = utils.parse_command(argv)
    h, w, spe, out_dir = extra["--height"], extra["--width"], extra["--steps-per-epoch"], extra["--output"]
    os.makedirs(out_dir, exist_ok=True)
    made = create_model(args, [h, w])
    model, loss_weights = made if isinstance(made, tuple) else (made, None)
    model = model.cuda()
    start_epoch, best = 0, Result()
    best.set_to_worst()
    step = HipTrainStep(model, args.batch_size, h, w, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay,
                        loss_weights=loss_weights, criterion=args.criterion, storage=extra["--storage"],
                        operands=None if extra["--storage"] == "bf16" else extra["--operands"],
                        autotune=os.environ.get("RD_AUTOTUNE", "1") == "1")     # main.py:11,47 (cudnn.benchmark = True): time the plans once
    if args.resume:                              # main.py:235-266
        ck = utils.load_checkpoint(args.resume)
        model.load_state_dict(ck["model_state_dict"], strict=False)
        step.load_state_dict(ck["optimizer_state_dict"])
        start_epoch, best = ck["epoch"] + 1, ck["best_result"]
        print("=> resumed from %s (epoch %d, best rmse %.3f)" % (args.resume, ck["epoch"], best.rmse))
    for epoch in range(start_epoch, args.epochs):
        lr = utils.adjust_learning_rate(step, epoch, args.lr)
        model.train()
        meter, t0 = AverageMeter(), time.time()
        for it in range(spe):
            x, t = make_batch(args.batch_size, h, w, 1234 + 1000 * epoch + it)
            loss, pred = step.step(x.cuda(), t.cuda())
            res = Result()
            res.evaluate(pred.detach(), t.cuda())          # one fused reduction + one small readback (metrics.py:34-58)
            meter.update(res, 0.0, 0.0, args.batch_size)
        torch.cuda.synchronize()
        avg = meter.average()
        print("epoch %d lr %.4g: loss %.4f rmse %.3f mae %.3f delta1 %.3f  (%.1f samples/s)"
              % (epoch, lr, loss.item(), avg.rmse, avg.mae, avg.delta1, spe * args.batch_size / (time.time() - t0)))
        # validate() (main.py:564-595): eval mode, batch 1, folded-BatchNorm inference graph
        infer = HipInference(model, 1, h, w)
        vm = AverageMeter()
        for it in range(2):
            x, t = make_batch(1, h, w, 99000 + it)
            r = Result()
            r.evaluate(infer(x.cuda()), t.cuda())
            vm.update(r, 0.0, 0.0, 1)
        val = vm.average()
        is_best = val.rmse < best.rmse
        if is_best:
            best = val
        path = utils.save_checkpoint({"args": args, "epoch": epoch, "arch": args.arch, "model_state_dict": model.state_dict(),
                                      "best_result": best, "optimizer_state_dict": step.state_dict()}, is_best, epoch, out_dir)
        print("   val rmse %.3f -> %s%s" % (val.rmse, path, " (best)" if is_best else ""))
    return model, step


if __name__ == "__main__":
    main()
