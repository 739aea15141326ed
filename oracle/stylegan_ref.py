"""Oracle (CPU, plain torch) for the StyleGAN2 operator layer and generator blocks.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Functional restatement: every network
function takes a flat dict ``P`` of tensors keyed by the reference ``state_dict`` names.

Reference lines followed (relative to /root/reference/eg3d):
  bias_act            torch_utils/ops/bias_act.py:92-122
  upfirdn2d           torch_utils/ops/upfirdn2d.py:168-213, setup_filter :72-116
  filtered_lrelu      torch_utils/ops/filtered_lrelu.py:122-155
  conv2d_resample     torch_utils/ops/conv2d_resample.py:48-143
  modulated_conv2d    training/networks_stylegan2.py:34-91
  fully_connected     training/networks_stylegan2.py:114-127
  mapping             training/networks_stylegan2.py:233-268
  synthesis layer     training/networks_stylegan2.py:311-330, torgb :353-357
  synthesis block     training/networks_stylegan2.py:417-461
  synthesis network   training/networks_stylegan2.py:503-518
  SR 8XDC             training/superresolution.py:264-290
"""
import math
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)

# name -> (fn, default alpha, default gain)
_ACTS = {
    'linear':   (lambda x, a: x, 0.0, 1.0),
    'relu':     (lambda x, a: F.relu(x), 0.0, SQRT2),
    'lrelu':    (lambda x, a: F.leaky_relu(x, a), 0.2, SQRT2),
    'tanh':     (lambda x, a: torch.tanh(x), 0.0, 1.0),
    'sigmoid':  (lambda x, a: torch.sigmoid(x), 0.0, 1.0),
    'elu':      (lambda x, a: F.elu(x), 0.0, 1.0),
    'selu':     (lambda x, a: F.selu(x), 0.0, 1.0),
    'softplus': (lambda x, a: F.softplus(x), 0.0, 1.0),
    'swish':    (lambda x, a: torch.sigmoid(x) * x, 0.0, SQRT2),
}


def act_default_gain(act):
    return _ACTS[act][2]


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    fn, def_alpha, def_gain = _ACTS[act]
    alpha = def_alpha if alpha is None else float(alpha)
    gain = def_gain if gain is None else float(gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def fir_filter(taps=(1, 3, 3, 1), gain=1.0, flip=False):
    """2-D low-pass FIR as the reference builds it for <8 taps (outer product, normalised)."""
    f = torch.as_tensor(taps, dtype=torch.float32)
    if f.ndim == 1:
        f = torch.outer(f, f)
    f = f / f.sum()
    if flip:
        f = f.flip([0, 1])
    return f * gain


def _pad4(p):
    if isinstance(p, int):
        return p, p, p, p
    if len(p) == 2:
        return p[0], p[0], p[1], p[1]
    return tuple(p)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """zero-insert upsample -> pad/crop -> FIR -> decimate (per channel)."""
    n, c, h, w = x.shape
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones(1, 1, dtype=torch.float32, device=x.device)
    if up > 1:
        z = x.new_zeros(n, c, h, up, w, up)
        z[:, :, :, 0, :, 0] = x
        x = z.reshape(n, c, h * up, w * up)
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        x = F.conv2d(x, k[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        x = F.conv2d(x, k[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        x = F.conv2d(x, k[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return x[:, :, ::down, ::down]


def upsample2d(x, f, up=2):
    fw = f.shape[-1]
    p0 = (fw + up - 1) // 2
    p1 = (fw - up) // 2
    return upfirdn2d(x, f, up=up, padding=(p0, p1, p0, p1), gain=up * up)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=SQRT2, slope=0.2,
                   clamp=None, flip_filter=False):
    x = bias_act(x, b)
    x = upfirdn2d(x, fu, up=up, padding=padding, gain=up ** 2, flip_filter=flip_filter)
    x = bias_act(x, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    return upfirdn2d(x, fd, down=down, flip_filter=flip_filter)


def conv2d_resample(x, w, f=None, up=1, padding=0, groups=1, flip_weight=True, store_f16=False):
    """Only the branches the generator reaches: up in {1,2}, down=1.  store_f16: the transposed convolution's output is an fp16 tensor
    (rounded once) before the FIR reads it -- the use_fp16 blocks' activation storage, see modulated_conv2d."""
    oc, icg, kh, kw = w.shape
    px0, px1, py0, py1 = _pad4(padding)
    fw = 1 if f is None else f.shape[-1]
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fw + up - 1) // 2
        py1 += (fw - up) // 2
    if kw == 1 and kh == 1 and up > 1:
        x = F.conv2d(x, w, groups=groups)
        return upfirdn2d(x, f, up=up, padding=(px0, px1, py0, py1), gain=up ** 2)
    if up > 1:
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, oc // groups, icg, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * icg, oc // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        if flip_weight:   # transposed conv wants the opposite flip convention
            wt = wt.flip([2, 3])
        x = F.conv_transpose2d(x, wt, stride=up, padding=[pyt, pxt], groups=groups)
        if store_f16:
            x = x.half().float()
        return upfirdn2d(x, f, padding=(px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt), gain=up ** 2)
    assert px0 == px1 and py0 == py1 and px0 >= 0
    if not flip_weight and (kw > 1 or kh > 1):
        w = w.flip([2, 3])
    return F.conv2d(x, w, padding=[py0, px0], groups=groups)


def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, f=None, demodulate=True,
                     flip_weight=True, fp16_operands=False, fp16_storage=False):
    """Fused (grouped-conv) path only: the one SPI takes (G.eval(), 'inference_only').
    fp16_operands: the arithmetic of the build's fp16-MFMA blocks (BASELINE config 5, `--sr_fp16`): tensors stay fp32, the two conv
    operands -- the activation and the modulated (demodulated) weight -- are rounded to fp16, products accumulate in fp32.  (The
    reference's own use_fp16 blocks also STORE activations in fp16, networks_stylegan2.py:421-461.)
    fp16_storage (round 5, with fp16_operands): the activations ARE fp16 tensors, as in the reference's use_fp16 blocks -- restated at the build's
    tensor boundaries: the block input, the transposed convolution's output (before its FIR) and every layer output (after bias / activation /
    clamp) are rounded to fp16 once; products and sums in between are fp32 (cuDNN / the plugins accumulate half tensors in fp32 too)."""
    n = x.shape[0]
    oc, ic, kh, kw = weight.shape
    w = weight.unsqueeze(0) * styles.reshape(n, 1, ic, 1, 1)
    if demodulate:
        d = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        w = w * d.reshape(n, oc, 1, 1, 1)
    if fp16_operands:
        x, w = x.half().float(), w.half().float()
    x = x.reshape(1, n * ic, *x.shape[2:])
    x = conv2d_resample(x, w.reshape(n * oc, ic, kh, kw), f=f, up=up, padding=padding, groups=n,
                        flip_weight=flip_weight, store_f16=fp16_storage)
    x = x.reshape(n, oc, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x


def fully_connected(x, weight, bias=None, lr_mul=1.0, act='linear'):
    w = weight * (lr_mul / math.sqrt(weight.shape[1]))
    b = bias * lr_mul if (bias is not None and lr_mul != 1) else bias
    if act == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return bias_act(x.matmul(w.t()), b, act=act)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


# --------------------------------------------------------------------------------------
# Generator pieces (functional over a state-dict P)
# --------------------------------------------------------------------------------------

def mapping(P, z, c, prefix='backbone.mapping.', num_layers=2, num_ws=14, lr_mul=0.01):
    x = normalize_2nd_moment(z.float())
    y = normalize_2nd_moment(fully_connected(c.float(), P[prefix + 'embed.weight'], P[prefix + 'embed.bias']))
    x = torch.cat([x, y], dim=1)
    for i in range(num_layers):
        x = fully_connected(x, P[f'{prefix}fc{i}.weight'], P[f'{prefix}fc{i}.bias'], lr_mul=lr_mul, act='lrelu')
    return x.unsqueeze(1).repeat(1, num_ws, 1)


def synthesis_layer(P, pfx, x, w, up=1, noise_mode='const', conv_clamp=None, gain=1.0, noise_rng=None, fp16_operands=False, fp16_storage=False):
    styles = fully_connected(w, P[pfx + 'affine.weight'], P[pfx + 'affine.bias'])
    noise = None
    if noise_mode == 'const':
        noise = P[pfx + 'noise_const'] * P[pfx + 'noise_strength']
    elif noise_mode == 'random':                     # networks_stylegan2.py:317-318: a fresh [N,1,res,res] draw per layer and call
        res = x.shape[-1] * up
        draw = noise_rng.randn(x.shape[0], 1, res, res) if noise_rng is not None else torch.randn(x.shape[0], 1, res, res)
        noise = draw * P[pfx + 'noise_strength']
    f = P[pfx + 'resample_filter']
    x = modulated_conv2d(x, P[pfx + 'weight'], styles, noise=noise, up=up, padding=1, f=f,
                         flip_weight=(up == 1), fp16_operands=fp16_operands, fp16_storage=fp16_storage)
    clamp = conv_clamp * gain if conv_clamp is not None else None
    y = bias_act(x, P[pfx + 'bias'], act='lrelu', gain=SQRT2 * gain, clamp=clamp)
    return y.half().float() if fp16_storage else y


def torgb_layer(P, pfx, x, w, conv_clamp=None, fp16_operands=False, fp16_storage=False):
    wt = P[pfx + 'weight']
    styles = fully_connected(w, P[pfx + 'affine.weight'], P[pfx + 'affine.bias'])
    styles = styles * (1.0 / math.sqrt(wt.shape[1] * wt.shape[2] * wt.shape[3]))
    x = modulated_conv2d(x, wt, styles, demodulate=False, fp16_operands=fp16_operands)
    y = bias_act(x, P[pfx + 'bias'], clamp=conv_clamp)
    return y.half().float() if fp16_storage else y           # (:443 of the reference: y.to(torch.float32) of a half tensor)


def synthesis_block(P, pfx, x, img, ws, first=False, noise_mode='const', conv_clamp=None, noise_rng=None, fp16_operands=False, fp16_storage=False):
    """ws: [N, num_conv+1, 512] (conv0?, conv1, torgb)."""
    wi = 0
    if first:
        x = P[pfx + 'const'].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
    else:
        if fp16_storage:
            x = x.half().float()                                   # (:436) x = x.to(dtype=torch.float16) at the block entry
        x = synthesis_layer(P, pfx + 'conv0.', x, ws[:, wi], up=2, noise_mode=noise_mode, conv_clamp=conv_clamp, noise_rng=noise_rng,
                            fp16_operands=fp16_operands, fp16_storage=fp16_storage)
        wi += 1
    x = synthesis_layer(P, pfx + 'conv1.', x, ws[:, wi], noise_mode=noise_mode, conv_clamp=conv_clamp, noise_rng=noise_rng,
                        fp16_operands=fp16_operands, fp16_storage=fp16_storage)
    wi += 1
    if img is not None:
        img = upsample2d(img, P[pfx + 'resample_filter'])
    y = torgb_layer(P, pfx + 'torgb.', x, ws[:, wi], conv_clamp=conv_clamp, fp16_operands=fp16_operands, fp16_storage=fp16_storage)
    img = y if img is None else img + y
    return x, img


def backbone_synthesis(P, ws, resolutions=(4, 8, 16, 32, 64, 128, 256), noise_mode='const',
                       prefix='backbone.synthesis.', noise_rng=None):
    ws = ws.float()
    x = img = None
    wi = 0
    for res in resolutions:
        first = (res == resolutions[0])
        nconv = 1 if first else 2
        x, img = synthesis_block(P, f'{prefix}b{res}.', x, img, ws[:, wi:wi + nconv + 1], first=first,
                                 noise_mode=noise_mode, noise_rng=noise_rng)
        wi += nconv
    return img


def superresolution_8xdc(P, rgb, x, ws, noise_mode='none', conv_clamp=256, prefix='superresolution.', fp16_operands=False, fp16_storage=False):
    """rgb [N,3,128,128], x [N,32,128,128] -> [N,3,512,512]; every layer driven by ws[:, -1]."""
    w3 = ws[:, -1:, :].repeat(1, 3, 1)
    if x.shape[-1] != 128:     # superresolution.py:282-286 (only reached by reduced-size test configs)
        x = F.interpolate(x, size=(128, 128), mode='bilinear', align_corners=False, antialias=True)
        rgb = F.interpolate(rgb, size=(128, 128), mode='bilinear', align_corners=False, antialias=True)
    kw = dict(noise_mode=noise_mode, conv_clamp=conv_clamp, fp16_operands=fp16_operands, fp16_storage=fp16_storage and fp16_operands)
    x, rgb = synthesis_block(P, prefix + 'block0.', x, rgb, w3, **kw)
    x, rgb = synthesis_block(P, prefix + 'block1.', x, rgb, w3, **kw)
    return rgb
