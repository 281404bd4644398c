"""Host-side builders of the generalised-convolution descriptors (RdConvDesc, see
include/radar_depth_hip.h).  Pure index arithmetic, testable without a GPU.

Every convolution on the hot path and every input-gradient of one is a list of at most four
*phases*; a phase is a set of taps (dh, dw, weight-slab) applied on a logical output grid:

    out[n, oh*OS + off_h, ow*OS + off_w, :] = sum_t in[n, oh*IS + dh_t, ow*IS + dw_t, :] @ W[widx_t]

  conv_fwd      nn.Conv2d k x k, stride 1|2, pad p              (model/models.py:96-112, resnet BasicBlock)
  conv_dgrad    its input gradient; stride 2 becomes 4 parity phases of a stride-1 conv over dy
  upproj_fwd    Unpool(2) + 5x5 conv as 4 phases with 9/6/6/4 taps on the LOW-RES input
                (models.py:13-27,181-209; the zero-skipping identity of SURVEY.md 8a row 6)
  upproj_dgrad  its input gradient: one 25-tap phase reading dout with input stride 2
"""
from ._lib import RD_MAX_PHASES, RD_MAX_TAPS, RdConvDesc, RdPhase


def _phase(taps, lh, lw, off=(0, 0)):
    """taps: list of (dh, dw, widx)."""
    assert 1 <= len(taps) <= RD_MAX_TAPS
    p = RdPhase()
    p.n_taps = len(taps)
    p.out_off_h, p.out_off_w = off
    p.lh, p.lw = lh, lw
    p.dh_min = min(t[0] for t in taps)
    p.dh_max = max(t[0] for t in taps)
    p.dw_min = min(t[1] for t in taps)
    p.dw_max = max(t[1] for t in taps)
    for i, (dh, dw, wi) in enumerate(taps):
        p.dh[i], p.dw[i], p.widx[i] = dh, dw, wi
    return p


def _desc(N, Hi, Wi, Cin, ldi, Ho, Wo, Cout, ldo, in_stride, out_stride, phases):
    assert 1 <= len(phases) <= RD_MAX_PHASES
    d = RdConvDesc()
    d.N, d.Hi, d.Wi, d.Cin, d.ldi = N, Hi, Wi, Cin, ldi
    d.Ho, d.Wo, d.Cout, d.ldo = Ho, Wo, Cout, ldo
    d.in_stride, d.out_stride, d.n_phases = in_stride, out_stride, len(phases)
    for i, p in enumerate(phases):
        d.phase[i] = p
    return d


def conv_out_size(size, k, stride, pad):
    return (size + 2 * pad - k) // stride + 1


def conv_fwd(N, Hi, Wi, Cin, Cout, k, stride, pad, ldi=None, ldo=None):
    Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
    taps = [(kh - pad, kw - pad, kh * k + kw) for kh in range(k) for kw in range(k)]
    return _desc(N, Hi, Wi, Cin, ldi or Cin, Ho, Wo, Cout, ldo or Cout, stride, 1, [_phase(taps, Ho, Wo)])


def conv_dgrad(N, Hi, Wi, Cin, Cout, k, stride, pad, ld_dy=None, ld_dx=None):
    """Gradient w.r.t. the input of conv_fwd(...).  The descriptor's 'input' is dy [N,Ho,Wo,Cout] and its
    'output' is dx [N,Hi,Wi,Cin]; weights must be packed transposed ([slab][Cout][Cin]).
    Returns (desc, needs_zero_fill): stride-2 parity phases without taps leave dx untouched."""
    Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
    if stride == 1:
        taps = [(pad - kh, pad - kw, kh * k + kw) for kh in range(k) for kw in range(k)]
        return _desc(N, Ho, Wo, Cout, ld_dy or Cout, Hi, Wi, Cin, ld_dx or Cin, 1, 1, [_phase(taps, Hi, Wi)]), False
    assert stride == 2
    phases, zero_fill = [], False
    for pi in range(2):
        for pj in range(2):
            lh, lw = (Hi - pi + 1) // 2, (Wi - pj + 1) // 2
            if lh <= 0 or lw <= 0:
                continue
            taps = [((pi + pad - kh) // 2, (pj + pad - kw) // 2, kh * k + kw)
                    for kh in range(k) for kw in range(k)
                    if (pi + pad - kh) % 2 == 0 and (pj + pad - kw) % 2 == 0]
            if not taps:
                zero_fill = True
                continue
            phases.append(_phase(taps, lh, lw, (pi, pj)))
    return _desc(N, Ho, Wo, Cout, ld_dy or Cout, Hi, Wi, Cin, ld_dx or Cin, 1, 2, phases), zero_fill


def upproj_fwd(N, H, W, Cin, Cout, ldi=None, ldo=None):
    """Unpool(stride 2) followed by a 5x5/pad-2 conv, evaluated on the low-res input [N,H,W,Cin];
    output [N,2H,2W,Cout] (Cout = both branches' 5x5 convs side by side)."""
    phases = []
    for pi in range(2):
        for pj in range(2):
            taps = [((pi + kh - 2) // 2, (pj + kw - 2) // 2, kh * 5 + kw)
                    for kh in range(5) for kw in range(5) if (pi + kh) % 2 == 0 and (pj + kw) % 2 == 0]
            phases.append(_phase(taps, H, W, (pi, pj)))
    return _desc(N, H, W, Cin, ldi or Cin, 2 * H, 2 * W, Cout, ldo or Cout, 1, 2, phases)


def upproj_dgrad(N, H, W, Cin, Cout, ld_dy=None, ld_dx=None):
    """dx[i,j] = sum_{kh,kw} dout[2i+2-kh, 2j+2-kw] @ W[kh,kw]^T : 'input' dout [N,2H,2W,Cout], 'output' dx [N,H,W,Cin]."""
    taps = [(2 - kh, 2 - kw, kh * 5 + kw) for kh in range(5) for kw in range(5)]
    return _desc(N, 2 * H, 2 * W, Cout, ld_dy or Cout, H, W, Cin, ld_dx or Cin, 2, 1, [_phase(taps, H, W)])


def desc_to_dict(d):
    """Plain-Python view of a descriptor (for tests and debugging)."""
    out = {k: getattr(d, k) for k in ("N", "Hi", "Wi", "Cin", "ldi", "Ho", "Wo", "Cout", "ldo", "in_stride", "out_stride")}
    out["phases"] = []
    for i in range(d.n_phases):
        p = d.phase[i]
        out["phases"].append({"lh": p.lh, "lw": p.lw, "off": (p.out_off_h, p.out_off_w),
                              "taps": [(p.dh[t], p.dw[t], p.widx[t]) for t in range(p.n_taps)]})
    return out
