"""Reload a finished inversion and render its novel-view video (mirror of spi/training/coaches/inference_coach.py:10-44).

``--G_1_type Inference --load_embedding_coach_name <coach>``: for every image of the dataset the checkpoint
``<checkpoints_dir>/<coach>/<name>.pt`` (``{'w', 'c', 'G'}``, written by ``BaseCoach.save``, base_coach.py:204-210) is
loaded into the coach's generator and ``log_video`` renders the 120-frame orbit into ``<video_output_dir>/<name>.mp4``.
A checkpoint written by the reference loads unchanged (same keys and shapes: tests/golden/manifest_full.json)."""
import os

from ...configs import paths_config, hyperparameters
from .base_coach import BaseCoach


class InferenceCoach(BaseCoach):
    def __init__(self, data_loader, use_wandb, **kw):
        super().__init__(data_loader, use_wandb, **kw)
        self.coach_name = 'InferenceCoach'
        self.build_name()

    def train(self):
        paths_config.experiments_output_dir += f'{self.coach_name}'
        output_dir = paths_config.experiments_output_dir
        done = []
        for idx, data in enumerate(self.data_loader):
            if self.image_counter >= hyperparameters.max_images_to_invert:
                break
            image_name = data['name'][0] if isinstance(data['name'], (list, tuple)) else data['name']
            paths_config.experiments_output_dir = os.path.join(output_dir, image_name)
            os.makedirs(paths_config.experiments_output_dir, exist_ok=True)
            ckpt_path = os.path.join(paths_config.checkpoints_dir, hyperparameters.load_embedding_coach_name, f'{image_name}.pt')
            w_pivot, camera, G = self.load(ckpt_path)
            self.log_video(w_pivot, G, os.path.join(paths_config.video_output_dir, f'{image_name}.mp4'))
            self.image_counter += 1
            done.append(dict(name=image_name, iters=0))
        paths_config.experiments_output_dir = output_dir
        return done
