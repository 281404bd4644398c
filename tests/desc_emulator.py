"""Test-only CPU interpreter of RdConvDesc (the semantics documented in include/radar_depth_hip.h),
used to check the host-side descriptor builders against torch.nn.functional convolutions without a GPU."""
import torch
import torch.nn.functional as F


def _gather(x, p, t, in_stride):
    """x [N,Hi,Wi,C] -> [N,lh,lw,C] of x[n, oh*IS+dh, ow*IS+dw] with zeros outside."""
    n, hi, wi, c = x.shape
    dh, dw = p.dh[t], p.dw[t]
    pad = 8
    xp = F.pad(x, (0, 0, pad, pad + in_stride * p.lw, pad, pad + in_stride * p.lh))
    h0, w0 = pad + dh, pad + dw
    return xp[:, h0:h0 + in_stride * p.lh:in_stride, w0:w0 + in_stride * p.lw:in_stride, :][:, :p.lh, :p.lw]


def run_desc(d, x, w):
    """x [N,Hi,Wi,Cin] (first Cin of ldi), w [slabs,Cin,Cout] -> out [N,Ho,Wo,Cout]; untouched pixels stay NaN."""
    out = torch.full((d.N, d.Ho, d.Wo, d.Cout), float("nan"), dtype=x.dtype)
    for i in range(d.n_phases):
        p = d.phase[i]
        acc = torch.zeros(d.N, p.lh, p.lw, d.Cout, dtype=x.dtype)
        for t in range(p.n_taps):
            acc += _gather(x, p, t, d.in_stride) @ w[p.widx[t]]
        out[:, p.out_off_h::d.out_stride, p.out_off_w::d.out_stride][:, :p.lh, :p.lw] = acc
    return out


def run_wgrad(d, x, dout, n_slabs):
    """dW[slab] [Cin,Cout] = sum over pixels of in(...)^T dout(...)."""
    dw = torch.zeros(n_slabs, d.Cin, d.Cout, dtype=x.dtype)
    for i in range(d.n_phases):
        p = d.phase[i]
        dy = dout[:, p.out_off_h::d.out_stride, p.out_off_w::d.out_stride][:, :p.lh, :p.lw]
        for t in range(p.n_taps):
            g = _gather(x, p, t, d.in_stride)
            dw[p.widx[t]] += torch.einsum("nhwi,nhwo->io", g, dy)
    return dw


def pack_fwd(w_oihw):
    o, i, kh, kw = w_oihw.shape
    return w_oihw.permute(2, 3, 1, 0).reshape(kh * kw, i, o).contiguous()


def pack_dgrad(w_oihw):
    o, i, kh, kw = w_oihw.shape
    return w_oihw.permute(2, 3, 0, 1).reshape(kh * kw, o, i).contiguous()
