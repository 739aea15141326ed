"""Super-resolution module of the FFHQ-512 EG3D generator on MI355X kernels.

Drop-in surface of ``SuperresolutionHybrid8XDC`` (eg3d/training/superresolution.py:264-290): two
StyleGAN2 synthesis blocks 128^2 -> 256^2 -> 512^2, every layer driven by the last W+ row,
``conv_clamp = 256`` whenever ``sr_num_fp16_res > 0`` (the clamp stays active in fp32).
Precision is fp32 (the parity configurations); the sibling SR variants of the reference
(8X, 4X, 2X, Deepfp32) are selected by other pickles and are not on the SPI path.
"""
import torch
from .networks_stylegan2 import SynthesisBlock


class SuperresolutionHybrid8XDC(torch.nn.Module):
    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None,
                 channel_base=None, channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        use_fp16 = sr_num_fp16_res > 0
        self.input_resolution = 128
        self.sr_antialias = sr_antialias
        clamp = 256 if use_fp16 else None
        self.block0 = SynthesisBlock(channels, 256, w_dim=512, resolution=256, img_channels=3, is_last=False, use_fp16=use_fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(256, 128, w_dim=512, resolution=512, img_channels=3, is_last=True, use_fp16=use_fp16,
                                     conv_clamp=clamp, **block_kwargs)

    def forward(self, rgb, x, ws, **block_kwargs):
        ws = ws[:, -1:, :].repeat(1, 3, 1)
        if x.shape[-1] != self.input_resolution:      # only reduced-size test configurations get here
            size = (self.input_resolution, self.input_resolution)
            x = torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
            rgb = torch.nn.functional.interpolate(rgb, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
        x, rgb = self.block0(x, rgb, ws, **block_kwargs)
        x, rgb = self.block1(x, rgb, ws, **block_kwargs)
        return rgb


# dotted names an EG3D pickle may carry in rendering_kwargs['superresolution_module'] (triplane.py:41)
SR_REGISTRY = {
    'training.superresolution.SuperresolutionHybrid8XDC': SuperresolutionHybrid8XDC,
    'spi_amd.training.superresolution.SuperresolutionHybrid8XDC': SuperresolutionHybrid8XDC,
}
