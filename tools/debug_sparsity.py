"""Print, for one rot / mirror-rot backward of the stage-2 loop, the non-zero share of the 16-pixel gradient segments every generator conv sees."""
import sys, os, tempfile, torch
sys.path.insert(0, '.')
from spi_amd.configs import hyperparameters as hp, paths_config
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach
from spi_amd.data.images_dataset import SyntheticDataset
from spi_amd.torch_utils.ops import conv2d_mfma as cm
dev = 'cuda'
tmp = tempfile.mkdtemp(prefix='spi_ds_')
for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
    setattr(paths_config, k, f'{tmp}/{k}/')
hp.LPIPS_value_threshold = -1.0
hp.pt_rot_lambda, hp.pt_mirror_rot_lambda, hp.pt_depth_lambda = 0.1, 0.05, 0.0
torch.manual_seed(0)
G = TriPlaneGenerator(**ffhq512_kwargs(depth_resolution=96, depth_resolution_importance=96)).eval().requires_grad_(False).to(dev)
G.neural_rendering_resolution = 128
coach = RotBboxCoach(None, False, G=G, synthetic=True)
d = SyntheticDataset(1)[0]
data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in d.items()}
ctx = coach.prepare_image(data)
w_pivot = torch.randn(1, 14, 512, device=dev) * 0.5
orig = cm.seg_flags
log = []
def spy(x):
    f = orig(x)
    fl = f.float().reshape(x.shape[0], x.shape[2], -1) if (x.shape[3] % 16 == 0) else None
    rows = (fl.amax(2).mean().item() if fl is not None else float('nan'))
    log.append((tuple(x.shape), f.float().mean().item(), rows))
    return f
cm.seg_flags = spy
coach.train_step(0, ctx, w_pivot)
torch.cuda.synchronize()
for s, m, r in log:
    print(f'dy {s}: {m * 100:5.1f} % of segments non-zero, {r * 100:5.1f} % of rows touched')
