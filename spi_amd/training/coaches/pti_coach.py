"""PTI baseline stage 2 (mirror of spi/training/coaches/pti_coach.py:34-98): one synthesis, L2 + LPIPS, Adam,
early stop at LPIPS <= 0.05.  Locality regularisation is disabled in the reference's configuration
(hyperparameters.use_locality_regularization = False) and is not implemented."""
import os
import torch

from ...configs import paths_config, hyperparameters, global_config
from ...criteria.l2_loss import l2_loss
from ...utils.rng import DeviceRNG
from ...torch_utils import zero_arena
from .base_coach import BaseCoach


class SingleIDCoach(BaseCoach):
    def __init__(self, data_loader, use_wandb, **kw):
        super().__init__(data_loader, use_wandb, **kw)
        self.coach_name = 'PTI_coach'
        self.build_name()

    def calc_loss(self, generated_images, real_images, target_feats=None):
        loss = 0.0
        loss_lpips = None
        if hyperparameters.pt_l2_lambda > 0:
            loss = loss + l2_loss(generated_images, real_images) * hyperparameters.pt_l2_lambda
        if hyperparameters.pt_lpips_lambda > 0:
            loss_lpips = torch.squeeze(self.lpips_loss(generated_images, real_images, y_feats=target_feats))
            loss = loss + loss_lpips * hyperparameters.pt_lpips_lambda
        return loss, loss_lpips

    @zero_arena.closes_iteration
    def train_step(self, image, camera, w_pivot, target_feats=None, rng=None):
        rng = rng or self.rng or DeviceRNG(self.device)
        G = self.G
        zero_arena.begin(w_pivot.device, key='pti')
        m = G.neural_rendering_resolution ** 2
        rk = G.rendering_kwargs
        noise = (rng.rand(1, m, int(rk['depth_resolution']), 1), rng.rand(m, max(int(rk['depth_resolution_importance']), 1)))
        img = G.synthesis(w_pivot.detach(), camera, noise_mode='const', render_noise=noise)['image']
        loss, loss_lpips = self.calc_loss(img, image, target_feats)
        self.optimizer.zero_grad()
        stop_flag = self._async_flag(loss_lpips <= hyperparameters.LPIPS_value_threshold) if loss_lpips is not None else None
        loss.backward()                                        # enqueued before the flag is read: the GPU stays busy during the host wait
        zero_arena.finish()
        if stop_flag is not None and stop_flag():              # (:95-96) stops before the optimiser step; the extra gradients are discarded
            return True, dict(loss=loss.detach(), lpips=loss_lpips.detach())
        self.optimizer.step()
        return False, dict(loss=loss.detach(), lpips=loss_lpips.detach() if loss_lpips is not None else None)

    def train(self):
        paths_config.experiments_output_dir += f'{self.coach_name}'
        output_dir = paths_config.experiments_output_dir
        stats = []
        for idx, data in enumerate(self.data_loader):
            if self.image_counter >= hyperparameters.max_images_to_invert:
                break
            image_name = data['name'][0] if isinstance(data['name'], (list, tuple)) else data['name']
            image = data['img'].to(self.device).float()
            camera = torch.as_tensor(data['c']).to(self.device).float().reshape(-1, 25)
            mask = data['mask'].to(self.device)
            fg_mask = 1 - (mask.reshape(1, 1, *mask.shape[-2:]) == 0).float()
            paths_config.experiments_output_dir = os.path.join(output_dir, image_name)
            os.makedirs(paths_config.experiments_output_dir, exist_ok=True)
            if self.use_wandb:                                   # (:52-53)
                self.log_target(image, 'target_image')
            self.restart_training()
            embedding_loaded = hyperparameters.load_embedding_coach_name is not None and os.path.isfile(
                f"{paths_config.embedding_base_dir}/{hyperparameters.load_embedding_coach_name}/{image_name}.pt")
            w_pivot = self.get_inversion(image_name, image, camera, fg_mask=fg_mask)
            feats = self.lpips_loss.features(image)
            iters = 0
            log_images_counter = 0
            from ...torch_utils.misc import quiet_gc
            with quiet_gc():
                for i in range(hyperparameters.G_1_step):
                    stop, losses = self.train_step(image, camera, w_pivot, feats)
                    iters += 1
                    if stop:
                        break
                    if self.use_wandb and log_images_counter % global_config.log_snapshot == 0:        # (:81-82)
                        self.log_image_from_w(w_pivot, camera, self.G, f'{image_name}_G1_inv_{log_images_counter}')
                    global_config.training_step += 1
                    log_images_counter += 1
            self.image_counter += 1
            self.finish_image(image_name, image, camera, w_pivot)
            stats.append(dict(name=image_name, iters=iters, stage1_iters=0 if embedding_loaded else hyperparameters.first_inv_steps))
            self.post_process(w_pivot, camera, self.G, image_name)
        paths_config.experiments_output_dir = output_dir
        if self.use_wandb:                                        # (:98-99)
            self.log_metric()
        return stats
