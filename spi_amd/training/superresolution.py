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

    @staticmethod
    @torch.no_grad()
    def needed_output_maps(region):
        """region [N,1,512,512] (non-zero = the caller reads this image pixel) -> {(OH, OW): int32 segment flags} for
        ``conv2d_mfma.needed_output``: per conv output resolution of the two blocks, the region dilated by what the layers between
        that conv and the image read (3x3 convs: 1 px; the 4x4 up-sampling FIR: [-1, +2] px; torgb / skip up-sampling: the FIR).
        The margins are generous (4 px at 512^2 and again 4 px at 256^2) -- a tile of the implicit GEMM is 128+ pixels wide anyway."""
        from ..torch_utils.ops import conv2d_mfma
        F = torch.nn.functional
        assert region.ndim == 4 and region.shape[1] == 1 and region.shape[-1] == 512 and region.shape[-2] == 512
        m = (region != 0).float()
        m512 = F.max_pool2d(m, 9, 1, 4)                                   # block1.conv1 / torgb outputs (512^2)
        m513 = F.max_pool2d(F.pad(m512, (0, 1, 0, 1)), 5, 1, 2)           # block1.conv0's transposed conv (513^2), read through the FIR
        m256 = F.max_pool2d(F.max_pool2d(m513[:, :, :512, :512], 2), 9, 1, 4)   # block0.conv1 / torgb outputs (256^2)
        return {(512, 512): conv2d_mfma.seg_flags(m512), (513, 513): conv2d_mfma.seg_flags(m513.contiguous()), (256, 256): conv2d_mfma.seg_flags(m256)}

    def forward(self, rgb, x, ws, **block_kwargs):
        ws = ws[:, -1:, :].repeat(1, 3, 1)
        if x.shape[-1] != self.input_resolution:      # only reduced-size test configurations get here
            size = (self.input_resolution, self.input_resolution)
            x = torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
            rgb = torch.nn.functional.interpolate(rgb, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
        # the six style layers of the two blocks in one launch (one more in the backward)
        from .networks_stylegan2 import multi_affine, multi_modulate
        l0, l1 = self.block0.affine_layers(), self.block1.affine_layers()
        st = multi_affine(ws, [(m, k) for k, m in enumerate(l0)] + [(m, k) for k, m in enumerate(l1)])
        wm = multi_modulate(l0 + l1, st)
        x, rgb = self.block0(x, rgb, ws, styles=(st[:len(l0)] if st is not None else None), w_mods=(wm[:len(l0)] if wm is not None else None), **block_kwargs)
        x, rgb = self.block1(x, rgb, ws, styles=(st[len(l0):] if st is not None else None), w_mods=(wm[len(l0):] if wm is not None else None), **block_kwargs)
        return rgb


# dotted names an EG3D pickle may carry in rendering_kwargs['superresolution_module'] (triplane.py:41)
SR_REGISTRY = {
    'training.superresolution.SuperresolutionHybrid8XDC': SuperresolutionHybrid8XDC,
    'spi_amd.training.superresolution.SuperresolutionHybrid8XDC': SuperresolutionHybrid8XDC,
}
