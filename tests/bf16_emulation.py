"""CPU emulation of the HIP bf16-STORAGE plan on the oracle modules (test infrastructure, imported by the GPU parity tests and by
the CPU conditioning test): same rounding points as the plan -- see emulate_bf16_storage."""
import types

import torch
import torch.nn.functional as F


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class BfStem(torch.autograd.Function):
    """rd_stem_fwd_bf16: forward with both operands rounded to bf16; the stem's weight and input gradients are the fp32 kernels."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        return F.conv2d(_bf(x), _bf(w), None, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.sp
        dx = torch.nn.grad.conv2d_input(x.shape, w, dy, stride, pad) if ctx.needs_input_grad[0] else None
        return dx, torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, pad), None, None


def emulate_bf16_storage(model):
    """The oracle with the HIP bf16-storage plan's rounding points: conv operands / gradients bf16 (test_gpu_bf16._BfConv,
    weight gradients of EVERY gconv-lowered layer from rounded operands), and every tensor the plan stores in HBM rounded to bf16
    in the forward (conv outputs, activated tensors, the two activation-free BatchNorm outputs of the fusion chain) with its
    gradient rounded in the backward (the stored gradient tensors)."""

    class RoundSTE(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return _bf(x)

        @staticmethod
        def backward(ctx, g):
            return _bf(g)

    class BfConvAll(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, stride, pad):
            ctx.save_for_backward(x, w)
            ctx.sp = (stride, pad)
            return F.conv2d(_bf(x), _bf(w), None, stride, pad)

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            stride, pad = ctx.sp
            dx = torch.nn.grad.conv2d_input(x.shape, _bf(w), _bf(dy), stride, pad) if ctx.needs_input_grad[0] else None
            return dx, torch.nn.grad.conv2d_weight(_bf(x), w.shape, _bf(dy), stride, pad), None, None

    def stored(raw):
        """The tensor as the plan stores it (bf16) -- remembering the fp32 accumulator values it was rounded from: the conv
        epilogue takes the BatchNorm partial sums from the accumulators, not from the rounded tensor (with 48 values per channel
        at the bottleneck of the test geometry the two means differ by a visible fraction of a bf16 ulp)."""
        y = RoundSTE.apply(raw)
        y._unrounded = raw
        return y

    def bn_forward(self, x):
        if not self.training:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
        xu = getattr(x, "_unrounded", x)
        mean = xu.mean((0, 2, 3))
        var = xu.var((0, 2, 3), unbiased=False)
        scale = self.weight * torch.rsqrt(var + self.eps)
        return x * scale[None, :, None, None] + (self.bias - mean * scale)[None, :, None, None]

    n = 0
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.Conv2d) and m.in_channels % 16 == 0 and m.out_channels > 1:
            m.forward = types.MethodType(lambda self, x: stored(BfConvAll.apply(x, self.weight, self.stride, self.padding)), m)
            n += 1
        elif isinstance(m, torch.nn.Conv2d) and m.kernel_size == (7, 7):
            if m.out_channels >= 64:       # RGB stem: bf16 operands forward, fp32 weight gradient
                m.forward = types.MethodType(lambda self, x: stored(BfStem.apply(x, self.weight, self.stride, self.padding)), m)
            else:                          # depth stem: fp32 arithmetic, bf16 output tensor
                m.forward = types.MethodType(lambda self, x: stored(F.conv2d(x, self.weight, None, self.stride, self.padding)), m)
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.forward = types.MethodType(bn_forward, m)
        elif isinstance(m, (torch.nn.ReLU, torch.nn.LeakyReLU)):
            slope = getattr(m, "negative_slope", None)
            m.forward = types.MethodType(lambda self, x, slope=slope: RoundSTE.apply(F.relu(x) if slope is None else F.leaky_relu(x, slope)), m)
        if name.split(".")[-1] in ("bn_fusion", "bn2") and isinstance(m, torch.nn.BatchNorm2d) and name.count(".") <= 1:
            m.register_forward_hook(lambda mod, inp, out: RoundSTE.apply(out))
    return n


