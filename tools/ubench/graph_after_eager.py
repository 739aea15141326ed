#!/usr/bin/env python3
"""Which captured op stops replaying correctly after a few thousand eager launches?  (round 4: the W projector's graph returned garbage
when replayed after eager PTI iterations.)  Each candidate body is captured alone, replayed, then 6000 tiny eager launches run, then it is
replayed again and compared with its eager result."""
import os, sys
import torch
import torch.nn.functional as F
if os.environ.get('ENV_AFTER_IMPORT'):                       # does the switch still take effect when it is set after `import torch` (before the first HIP call)?
    os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = 'cuda'
torch.manual_seed(0)
img = torch.randn(1, 3, 512, 512, device=dev)
feats = [torch.randn(1, c, s, s, device=dev) for c, s in ((64, 256), (128, 128), (256, 64), (512, 32), (512, 16))]
lins = [torch.rand(c, device=dev) / c for c in (64, 128, 256, 512, 512)]
tgt = torch.randn(1, sum(f[0].numel() for f in feats), device=dev)


def body_area():
    return F.interpolate((img + 1) * (255 / 2), size=(256, 256), mode='area').sum()


def body_norm_cat():
    out = []
    for f, l in zip(feats, lins):
        hw = f.shape[2] * f.shape[3]
        g = f / (torch.sqrt(torch.sum(f * f, dim=1, keepdim=True)) + 1e-10)
        out.append((g * torch.sqrt(l / hw).view(1, -1, 1, 1)).flatten(1))
    return (tgt - torch.cat(out, dim=1)).square().sum()


def body_cat_only():
    return torch.cat([f.flatten(1) for f in feats], dim=1).sum()


def body_sqsum():
    return (tgt - 0.5).square().sum()


def body_repeat():
    w = torch.randn(1, 1, 512, device=dev)
    return w.repeat([1, 14, 1]).sum()


def body_conv():
    from spi_amd.torch_utils.ops import conv2d_mfma
    x = feats[0]
    w = torch.ones(64, 64, 3, 3, device=dev) * 0.01
    return conv2d_mfma.conv2d(x, w, padding=1).sum()


def body_maxpool():
    return F.max_pool2d(feats[0], 2).sum()


def body_grad():
    x = img.clone().requires_grad_(True)
    y = F.interpolate((x + 1) * (255 / 2), size=(256, 256), mode='area')
    (y * y).sum().backward()
    return x.grad.abs().sum()


def body_sumdim1():
    return sum(torch.sum(f * f, dim=1, keepdim=True).sum() for f in feats)


def _mk(i):
    return lambda: torch.sum(feats[i] * feats[i], dim=1, keepdim=True).sum()


for _i in range(5):
    globals()[f'body_sumdim1_{_i}'] = _mk(_i)


def body_sumdim1_nokeep():
    return torch.sum(feats[4] * feats[4], dim=1).sum()


def body_norm_dim1():
    return torch.linalg.vector_norm(feats[4], dim=1, keepdim=True).sum()


def body_mse():
    return F.mse_loss(img, img * 0.5)


def body_mean_dims():
    return feats[0].mean(dim=(2, 3)).sum()


def body_divbc():
    return sum((f / (f[:, :1] * 0 + 2.0)).sum() for f in feats)


def body_sqrtview():
    return sum((f * torch.sqrt(l / 7.0).view(1, -1, 1, 1)).sum() for f, l in zip(feats, lins))


def body_scalar():
    return sum((torch.sqrt(f.abs()) + 1e-10).sum() for f in feats)


def body_sub_sq():
    c = torch.cat([f.flatten(1) for f in feats], dim=1)
    return (tgt - c).square().sum()


names = sys.argv[1:] or ['sumdim1_0', 'sumdim1_1', 'sumdim1_2', 'sumdim1_3', 'sumdim1_4', 'sumdim1_nokeep', 'norm_dim1', 'mse', 'mean_dims']
for name in names:
    fn = globals()['body_' + name]
    torch.manual_seed(1)
    ref = float(fn())
    torch.manual_seed(1)
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    torch.manual_seed(1)
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    first = float(out)
    t = torch.zeros(1024, device=dev)
    if os.environ.get('SHOW_POOLS'):
        segs = [(s_['address'], s_['total_size'], tuple(s_.get('segment_pool_id', (0, 0))), s_.get('stream')) for s_ in torch.cuda.memory_snapshot()]
        inside = [sg for sg in segs if sg[0] <= t.data_ptr() < sg[0] + sg[1]]
        print('   t at', hex(t.data_ptr()), 'lies in segment', [(hex(a), sz, pid, st) for a, sz, pid, st in inside], ' out at', hex(out.data_ptr()),
              [(hex(a), sz, pid) for a, sz, pid, st in segs if a <= out.data_ptr() < a + sz], ' private segments:', sum(1 for sg in segs if sg[2] != (0, 0)), flush=True)
    for _ in range(int(os.environ.get('NLAUNCH', '6000'))):
        t.add_(1.0)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    second = float(out)
    print(f'{name:10s} eager {ref:.6g}  replay {first:.6g}  replay after the eager launches {second:.6g}  {"OK" if second == first else "CHANGED"}', flush=True)
