"""Shared state of the stage-2 coaches (mirror of spi/training/coaches/base_coach.py:36-270).

Same responsibilities and names -- ``restart_training``, ``get_inversion`` / ``calc_inversions`` (stage-1 dispatch on
``first_inv_type``), ``save`` / ``load`` / ``post_process``, ``build_name`` -- with these MI355X-first changes, all
result-identical to the reference:
  * the generator is read from disk ONCE; ``restart_training`` restores G from the frozen ``original_G`` that
    already sits in HBM instead of unpickling both copies again per image (base_coach.py:53-60);
  * one fused Adam launch per step (training/optim.py);
  * ``use_wandb`` only gates disk logging (it never touched wandb in the reference either): target / w_inv / G1_inv images and
    orbit videos under ``experiments_output_dir/<image>/``, the L2 / LPIPS / ID table in ``metric_log.txt`` (:80-87, :141-198);
  * the perceptual-loss weights are read from the files named in ``paths_config`` and a missing file raises
    (criteria/weights.py); seeded stand-ins only with ``synthetic=True`` / ``--synthetic``.
"""
import abc
import os
import numpy as np
import torch

from ...configs import global_config, paths_config, hyperparameters
from ...criteria.lpips.lpips import LPIPS
from ...criteria import weights as pretrained
from ...utils import load_utils
from ...utils.camera_utils import cal_mirror_c
from ..optim import Adam
from ..projectors import w_plus_projector, mirror_projector, w_projector


def toogle_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def fix_seed():
    torch.manual_seed(0)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(0)
    np.random.seed(0)


class BaseCoach:
    def __init__(self, data_loader, use_wandb, G=None, lpips_loss=None, vgg16=None, rng=None, synthetic=None):
        self.use_wandb = use_wandb
        self.data_loader = data_loader
        self.w_pivots = {}
        self.image_counter = 0
        self.metric_dic = {}
        self.coach_name = 'Base_coach'
        self.device = torch.device(global_config.device)
        self.rng = rng
        self.synthetic = pretrained.want_synthetic(synthetic)
        if lpips_loss is None:
            lpips_loss = LPIPS(net_type='vgg', weights=pretrained.lpips_vgg16_weights(self.synthetic))
        self.lpips_loss = lpips_loss.to(self.device).eval()
        self.vgg16 = vgg16                                           # 'sg' feature extractor; built on first use (_sg_vgg16)
        self.original_G = (G if G is not None else load_utils.load_eg3d(device=self.device)).to(self.device)
        self.original_G.eval().requires_grad_(False)
        self.G = None
        self.optimizer = None
        self.restart_training()

    def _async_flag(self, cond):
        """Start copying a 0-dim device boolean to pinned host memory; returns a callable that waits for THAT copy only."""
        if cond.device.type != 'cuda':
            return lambda: bool(cond)
        if getattr(self, '_flag_host', None) is None:
            self._flag_host = torch.zeros(1, dtype=torch.uint8).pin_memory()
        self._flag_host.copy_(cond.detach().reshape(1).to(torch.uint8), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()

        def wait():
            ev.synchronize()
            return bool(self._flag_host[0])
        return wait

    def restart_training(self):
        if self.G is None:
            self.G = load_utils.build_generator(self.original_G.init_args, self.original_G.init_kwargs, None, self.device)
            self.G.rendering_kwargs = self.original_G.rendering_kwargs
            self.G.neural_rendering_resolution = self.original_G.neural_rendering_resolution
            self.G.eval()
            toogle_grad(self.G, True)
            with torch.no_grad():
                self.G.load_state_dict(self.original_G.state_dict())
            self.optimizer = self.configure_optimizers()
        else:
            with torch.no_grad():
                self.G.load_state_dict(self.original_G.state_dict())     # copy_ into the flat parameter buffer
            self.optimizer.reset_state()
        fix_seed()

    def configure_optimizers(self):
        return Adam(self.G.parameters(), lr=hyperparameters.pti_learning_rate)

    def get_inversion(self, image_name, image, camera, fg_mask=None):
        embedding_dir = f'{paths_config.embedding_base_dir}/{self.coach_name}/'
        os.makedirs(embedding_dir, exist_ok=True)
        w_pivot = None
        if hyperparameters.load_embedding_coach_name is not None:
            w_pivot = self.load_inversions(f'{paths_config.embedding_base_dir}/{hyperparameters.load_embedding_coach_name}/', image_name)
        if w_pivot is None:
            w_pivot = self.calc_inversions(image_name, image, camera, fg_mask)
        torch.save(w_pivot.detach().cpu(), f'{embedding_dir}/{image_name}.pt')
        w_pivot = w_pivot.to(self.device)
        if self.use_wandb:                                           # (:80-87)
            camera_m = cal_mirror_c(camera)
            w_inv = self.log_image_from_w(w_pivot, camera, self.G, f'{image_name}_w_inv')
            w_inv_m = self.log_image_from_w(w_pivot, camera_m, self.G, f'{image_name}_w_inv_m')
            if getattr(hyperparameters, 'log_video', True):
                self.log_video(w_pivot, self.G, os.path.join(paths_config.experiments_output_dir, f'{image_name}_w_inv.mp4'))
            self.cal_metric(w_inv, image, 'w_inv', fake_m=w_inv_m)
        return w_pivot

    def log_image_from_w(self, w, c, G, name):
        """spi/utils/log_utils.py:7-15: synthesise, write <experiments_output_dir>/<name>.jpg, return the image tensor."""
        from PIL import Image
        if len(w.size()) <= 2:
            w = w.unsqueeze(0)
        with torch.no_grad():
            img_tensor = G.synthesis(w, c, noise_mode='const')['image']
            img = (img_tensor[0].permute(1, 2, 0) * 127.5 + 128).clamp(0, 255).to(torch.uint8).cpu().numpy()
        Image.fromarray(img).save(os.path.join(paths_config.experiments_output_dir, name + '.jpg'))
        return img_tensor

    def log_target(self, image, name):
        """log_utils.log_image (:45-54) for a [-1,1] RGB tensor."""
        from PIL import Image
        t = image.detach().float().cpu()
        t = t[0] if t.ndim == 4 else t
        arr = ((t.permute(1, 2, 0).numpy() + 1) / 2).clip(0, 1) * 255
        Image.fromarray(arr.astype('uint8')).save(os.path.join(paths_config.experiments_output_dir, name + '.jpg'))

    def finish_image(self, image_name, image, camera, w_pivot):
        """End-of-image logging of the G_1 stage (rot_bbox_cx_coach.py:160-164, pti_coach.py:88-94)."""
        if self.use_wandb and hyperparameters.G_1_step > 0:
            camera_m = cal_mirror_c(camera=camera)
            g1 = self.log_image_from_w(w_pivot, camera, self.G, f'{image_name}_G1_inv')
            g1_m = self.log_image_from_w(w_pivot, camera_m, self.G, f'{image_name}_G1_inv_m')
            if getattr(hyperparameters, 'log_video', True):
                self.log_video(w_pivot, self.G, path=os.path.join(paths_config.experiments_output_dir, f'{image_name}_G1_inv.mp4'))
            self.cal_metric(g1, image, 'G1_inv', fake_m=g1_m)

    def _sg_vgg16(self):
        """The `sg` extractor (base_coach.py:51 loads NVIDIA's TorchScript vgg16.pt, which cannot run on the HIP kernels and does not
        exist offline): its contract -- squared feature distance == LPIPS-VGG -- rebuilt on the LPIPS weights (criteria/sg_vgg.py)."""
        if self.vgg16 is None:
            from ...criteria.sg_vgg import SgVgg16
            self.vgg16 = SgVgg16(weights=pretrained.lpips_vgg16_weights(self.synthetic)).to(self.device).eval()
        return self.vgg16

    def load_inversions(self, embedding_dir, image_name):
        if image_name in self.w_pivots:
            return self.w_pivots[image_name]
        path = f'{embedding_dir}/{image_name}.pt'
        if not os.path.isfile(path):
            print('[ERROR]: No existing w codes.')
            return None
        w = torch.load(path, map_location='cpu').to(self.device)
        self.w_pivots[image_name] = w
        return w

    def calc_inversions(self, image_name, image, camera, fg_mask=None):
        kind = hyperparameters.first_inv_type
        assert kind in ['sg', 'sgw+', 'mir']
        common = dict(device=self.device, w_avg_samples=600, num_steps=hyperparameters.first_inv_steps, verbose=self.use_wandb,
                      w_name=image_name, initial_w=None, rng=self.rng)
        if kind == 'sg':
            return w_projector.project(self.G, image, camera, self._sg_vgg16(), **common)
        if kind == 'sgw+':
            return w_plus_projector.project(self.G, image, camera, lpips_func=self.lpips_loss, **common)
        return mirror_projector.project(self.G, image, camera, lpips_func=self.lpips_loss, fg_mask=fg_mask, **common)

    @abc.abstractmethod
    def train(self):
        pass

    def cal_metric(self, fake, gt, name, fake_m=None):
        """L2 / LPIPS / ID of a synthesised view (and of the mirrored view against the flipped photo), base_coach.py:141-152."""
        if getattr(self, 'metric', None) is None:
            from ...utils.metric_utils import Metric
            self.metric = Metric(lpips_loss=self.lpips_loss, device=self.device)
        d = self.metric_dic.setdefault(name, {'l2': [], 'lpips': [], 'id': [], 'l2_m': [], 'lpips_m': [], 'id_m': []})
        for key, (a, b) in (('', (gt, fake)), ('_m', (torch.flip(gt, dims=[3]), fake_m))):
            if b is None:
                continue
            l2, lp, ids = self.metric.run(a, b)
            d['l2' + key].append(l2); d['lpips' + key].append(lp); d['id' + key].append(ids)

    def log_metric(self):
        """Append the per-image table and averages to <experiments_output_dir>/metric_log.txt (format of base_coach.py:154-198)."""
        hp = hyperparameters
        with open(os.path.join(paths_config.experiments_output_dir, 'metric_log.txt'), 'a') as f:
            f.write(f'Coach name: {self.coach_name}\nhyperparameters.use_encoder: {hp.use_encoder}\n'
                    f'hyperparameters.first_inv_type: {hp.first_inv_type}\nhyperparameters.first_inv_steps: {hp.first_inv_steps}\n'
                    f'hyperparameters.G_1_step: {hp.G_1_step}\nhyperparameters.G_2_step: {hp.G_2_step}\n\n')
            for key, cur in self.metric_dic.items():
                msg = f'Mode: {key}\n'
                cnt = len(cur['l2'])
                tot = [0.0] * 6
                cols = ('l2', 'lpips', 'id', 'l2_m', 'lpips_m', 'id_m')
                for i in range(cnt):
                    v = [cur[c][i] if i < len(cur[c]) else float('nan') for c in cols]
                    msg += (f'ID: {i} L2: {v[0]:.6f}; Lpips: {v[1]:.6f}; ID Sim: {v[2]:.6f}; L2 M: {v[3]:.6f}; Lpips M: {v[4]:.6f}; '
                            f'ID Sim M: {v[5]:.6f};\n')
                    tot = [t + x for t, x in zip(tot, v)]
                tot = [t / max(cnt, 1) for t in tot]
                msg += (f'Mode: {key} AVG\nL2: {tot[0]:.6f}; Lpips: {tot[1]:.6f}; ID Sim: {tot[2]:.6f}; L2 M: {tot[3]:.6f}; '
                        f'Lpips M: {tot[4]:.6f}; ID Sim M: {tot[5]:.6f};\n')
                f.write(msg + '\n')

    def save(self, w, c, G, path):
        torch.save({'w': w.detach().cpu(), 'c': c.detach().cpu(), 'G': {k: v.detach().cpu() for k, v in G.state_dict().items()}}, path)

    def load(self, path):
        ckpt = torch.load(path, map_location='cpu')
        self.G.load_state_dict(ckpt['G'])
        return ckpt['w'].to(self.device), ckpt['c'].to(self.device), self.G

    def log_image(self, w, c, G, path):
        from PIL import Image
        if len(w.size()) <= 2:
            w = w.unsqueeze(0)
        with torch.no_grad():
            img = G.synthesis(w, c, noise_mode='const')['image'][0].permute(1, 2, 0)
            img = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).cpu().numpy()
        Image.fromarray(img).save(path)

    def post_process(self, w, c, G, name):
        self.save(w, c, G, path=os.path.join(paths_config.checkpoints_dir, self.coach_name, f'{name}.pt'))
        self.log_image(w, c, G, path=os.path.join(paths_config.images_output_dir, self.coach_name, name + '.jpg'))
        self.log_image(w, cal_mirror_c(c), G, path=os.path.join(paths_config.mirror_images_output_dir, self.coach_name, name + '.jpg'))
        if getattr(hyperparameters, 'log_video', True):
            self.log_video(w, G, path=os.path.join(paths_config.video_output_dir, self.coach_name, f'{name}.mp4'))

    def log_video(self, w, G, path):
        """120-frame novel-view orbit of the inverted latent (base_coach.py:236-237 -> video_utils.gen_interp_video)."""
        from ...utils.video_utils import gen_interp_video
        os.makedirs(os.path.dirname(path), exist_ok=True)
        return gen_interp_video(G, {'w': w.detach().clone()}, mp4=path)

    def build_name(self):
        hp = hyperparameters
        self.coach_name += f'_{hp.first_inv_type}_{hp.first_inv_steps}_{hp.G_1_type}_{hp.G_1_step}'
        if hp.use_encoder:
            self.coach_name += '_wenc'
        if hp.use_G_avg:
            self.coach_name += '_wgavg'
        self.coach_name += f'_rot_{hp.pt_rot_lambda}_mirrorrot_{hp.pt_mirror_rot_lambda}_depth_{hp.pt_depth_lambda}_tv_{hp.pt_tv_lambda}'
        if hp.use_adapt_yaw_range:
            self.coach_name += '_wadyaw'
        if hp.description is not None:
            self.coach_name += '_' + hp.description
        print('[COACH]:', self.coach_name)
        for d in (paths_config.checkpoints_dir, paths_config.embedding_base_dir, paths_config.experiments_output_dir,
                  paths_config.images_output_dir, paths_config.mirror_images_output_dir):
            os.makedirs(os.path.join(d, self.coach_name), exist_ok=True)
