"""TEST INFRASTRUCTURE (never imported by the product): a float64 model of the opt-in RZ_NET_SPLIT_F16_FP8 trunk arithmetic
(rlzero_amd/csrc/rz_net.hip: rt::slot_r F8, pack_rows_f8) -- not of the reference, which has no such mode.  parity: this mode is
OUTSIDE the reference's f32 arithmetic by design; the model says what the device should compute, the tests say how far that is
from PolicyValueNet.forward (rlzero/games/gomoku/policy_value_net.py:34-52) in float64.

conv1 / conv2 / the heads: every f32 operand as hi + lo f16 pieces, three products (hi hi + hi lo + lo hi) -- as exact as f32.
conv3: hi x hi on f16 pieces; the two cross terms with 8-bit operands:
    e5m2(z) x e4m3(w_lo 2^5) 2^-5  +  e5m2((z - f16(z)) 2^11) x e4m3(w_hi 2^-6) 2^-5,   w scaled into [2^13, 2^14) first.
Power-of-two activation scales commute with all of these roundings (away from the subnormals) and are left out."""
import numpy as np
import torch
import torch.nn.functional as F


def _round_to(x, mant_bits, min_exp, max_value):
    """x (float64 tensor) to a binary float with ``mant_bits`` explicit mantissa bits, smallest normal exponent ``min_exp``
    (below it: subnormals on the same grid), round to nearest even, saturating at ``max_value``."""
    x = x.double()
    sign = torch.sign(x)
    a = x.abs().clamp(max=max_value)
    _, e = torch.frexp(a)            # a = f 2^e, f in [0.5, 1)
    ex = (e - 1).clamp(min=min_exp)  # a = 1.m 2^ex
    step = torch.ldexp(torch.ones_like(a), ex - mant_bits)
    return sign * torch.round(a / step) * step


def e4m3(x):
    return _round_to(x, 3, -6, 448.0)


def e5m2(x):
    return _round_to(x, 2, -14, 57344.0)


def f16(x):
    return x.float().half().double()


def split(x):
    hi = f16(x)
    return hi, f16(x.double() - hi)


def conv_split3(a, w, b, pad):
    ah, al = split(a)
    wh, wl = split(w)
    c = lambda x, y: F.conv2d(x, y, padding=pad)
    return c(ah, wh) + c(ah, wl) + c(al, wh) + b.double().view(1, -1, 1, 1)


def weight_scale(w):
    """pack_split: the power of two that brings the largest |w| into [2^13, 2^14)."""
    wmax = float(w.abs().max())
    return 2.0 ** (14 - np.frexp(wmax)[1]) if wmax > 0 and np.isfinite(wmax) else 1.0


def conv3_fp8(a, w, b):
    sw = weight_scale(w)
    v = (w.double() * sw).float()   # (a power of two: exact)
    vh, vl = split(v)
    ah = f16(a)
    al = a.double() - ah             # (exact in f32; the kernel rounds it only to e5m2)
    c = lambda x, y: F.conv2d(x, y, padding=1)
    main = c(ah, vh)
    cross = (c(e5m2(a.double()), e4m3(vl * 32.0)) + c(e5m2(al * 2048.0), e4m3(vh / 64.0))) / 32.0
    return (main + cross) / sw + b.double().view(1, -1, 1, 1)


def forward(sd, planes, mode):
    """``sd``: PolicyValueNet.state_dict() (float32 tensors); ``planes`` [N, 4, B, B]; mode 'f64' (the reference's forward in
    float64), 'split' (the default trunk's arithmetic), 'fp8' (this mode's), 'f16' (what plain f16 operands would give: conv3's
    cross terms dropped).  -> (log_softmax [N, S], value [N]) in float64; activations pass between layers as f32."""
    x = planes.double()
    f32 = lambda t: t.float().double()
    if mode == 'f64':
        conv = lambda a, w, b, pad: F.conv2d(a, w.double(), b.double(), padding=pad)
        a = torch.relu(conv(x, sd['conv1.weight'], sd['conv1.bias'], 1))
        a = torch.relu(conv(a, sd['conv2.weight'], sd['conv2.bias'], 1))
        a = torch.relu(conv(a, sd['conv3.weight'], sd['conv3.bias'], 1))
    else:
        a = f32(torch.relu(conv_split3(x, sd['conv1.weight'], sd['conv1.bias'], 1)))
        a = f32(torch.relu(conv_split3(a, sd['conv2.weight'], sd['conv2.bias'], 1)))
        if mode == 'split':
            a = f32(torch.relu(conv_split3(a, sd['conv3.weight'], sd['conv3.bias'], 1)))
        elif mode == 'fp8':
            a = f32(torch.relu(conv3_fp8(a, sd['conv3.weight'], sd['conv3.bias'])))
        elif mode == 'f16':
            a = f32(torch.relu(F.conv2d(f16(a), f16(sd['conv3.weight']), padding=1) + sd['conv3.bias'].double().view(1, -1, 1, 1)))
        else:
            raise ValueError(mode)
    n = x.shape[0]
    pol = torch.relu(F.conv2d(a, sd['act_conv1.weight'].double(), sd['act_conv1.bias'].double())).reshape(n, -1)
    logits = pol @ sd['act_fc1.weight'].double().t() + sd['act_fc1.bias'].double()
    val = torch.relu(F.conv2d(a, sd['val_conv1.weight'].double(), sd['val_conv1.bias'].double())).reshape(n, -1)
    hid = torch.relu(val @ sd['val_fc1.weight'].double().t() + sd['val_fc1.bias'].double())
    value = torch.tanh(hid @ sd['val_fc2.weight'].double().t() + sd['val_fc2.bias'].double()).reshape(-1)
    return torch.log_softmax(logits, dim=1), value
