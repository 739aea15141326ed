"""Full-size synthesis with fp32 and with fp16-operand super-resolution on the same inputs: size of the difference."""
import sys, torch
sys.path.insert(0, '.')
from spi_amd.configs import global_config
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.utils import camera_utils as cu
dev = 'cuda'
torch.manual_seed(0)
G = TriPlaneGenerator(**ffhq512_kwargs(depth_resolution=48, depth_resolution_importance=48)).eval().requires_grad_(False).to(dev)
G.neural_rendering_resolution = 128
c = cu.cal_canonical_c(0.2, 0.0).to(dev)
w = torch.randn(1, 14, 512, device=dev) * 0.5
m = 128 * 128
noise = (torch.rand(1, m, 48, 1, device=dev), torch.rand(m, 48, device=dev))
outs = {}
for f16 in (False, True):
    global_config.enable_fp16_blocks = f16
    with torch.no_grad():
        o = G.synthesis(w, c, noise_mode='const', render_noise=noise)
    outs[f16] = o
a, b = outs[False]['image'], outs[True]['image']
print('image range', a.min().item(), a.max().item(), 'raw range', outs[False]['image_raw'].min().item(), outs[False]['image_raw'].max().item())
print('max abs diff', (a - b).abs().max().item(), 'rel to max', ((a - b).abs().max() / a.abs().max()).item(), 'mse a', (a ** 2).mean().item(), 'mse diff', ((a - b) ** 2).mean().item())
tgt = torch.zeros_like(a)
print('l2 vs zero: fp32', ((a - tgt) ** 2).mean().item(), 'fp16', ((b - tgt) ** 2).mean().item())
