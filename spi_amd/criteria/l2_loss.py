"""Mean-squared error (spi/criteria/l2_loss.py:3-8)."""
import torch


def l2_loss(real_images, generated_images):
    return torch.nn.functional.mse_loss(real_images, generated_images)
