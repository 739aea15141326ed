"""TriPlaneGenerator / OSGDecoder on MI355X kernels (drop-in surface of eg3d/training/triplane.py).

``synthesis(ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
use_cached_backbone=False, **synthesis_kwargs) -> {'image', 'image_raw', 'image_depth'}`` and
``mapping`` / ``sample`` / ``sample_mixed`` / ``forward`` keep the reference's meaning (triplane.py:48-107);
``state_dict`` keys match (backbone.*, superresolution.*, decoder.net.{0,2}.*).

Extras that do not exist in the reference (all optional, result-identical):
  * ``render_noise=(xi, u)`` injects the renderer's two random draws (parity tests);
  * ``skip_superresolution=True`` returns without 'image' when only depth is consumed
    (the depth-regularisation branch, rot_bbox_cx_coach.py:136-138).
"""
import torch
from ..torch_utils import misc
from .networks_stylegan2 import Generator as StyleGAN2Backbone, FullyConnectedLayer
from .superresolution import SR_REGISTRY
from .volumetric_rendering.renderer import ImportanceRenderer
from .volumetric_rendering.ray_sampler import RaySampler


class OSGDecoder(torch.nn.Module):
    """32 -> 64 -> 33 MLP (softplus hidden).  Holds the parameters; the arithmetic runs inside the fused
    gather+decode kernel, so ``forward`` on explicit features is only provided for API completeness."""
    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        self.net = torch.nn.Sequential(
            FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=options['decoder_lr_mul']),
            torch.nn.Softplus(),
            FullyConnectedLayer(self.hidden_dim, 1 + options['decoder_output_dim'], lr_multiplier=options['decoder_lr_mul']))

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)
        n, m, c = x.shape
        x = self.net(x.reshape(n * m, c)).reshape(n, m, -1)
        return {'rgb': torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001, 'sigma': x[..., 0:1]}


class TriPlaneGenerator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={},
                 rendering_kwargs={}, sr_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.renderer = ImportanceRenderer()
        self.ray_sampler = RaySampler()
        self.backbone = StyleGAN2Backbone(z_dim, c_dim, w_dim, img_resolution=256, img_channels=32 * 3,
                                          mapping_kwargs=mapping_kwargs, **synthesis_kwargs)
        sr_name = rendering_kwargs['superresolution_module']
        if sr_name not in SR_REGISTRY:
            raise NotImplementedError(f'superresolution module {sr_name!r} is not on the SPI path')
        self.superresolution = SR_REGISTRY[sr_name](channels=32, img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res,
                                                    sr_antialias=rendering_kwargs['sr_antialias'], **sr_kwargs)
        self.decoder = OSGDecoder(32, {'decoder_lr_mul': rendering_kwargs.get('decoder_lr_mul', 1), 'decoder_output_dim': 32})
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self._last_planes = None
        self.init_args = (z_dim, c_dim, w_dim, img_resolution, img_channels)
        self.init_kwargs = dict(sr_num_fp16_res=sr_num_fp16_res, mapping_kwargs=mapping_kwargs, rendering_kwargs=rendering_kwargs,
                                sr_kwargs=sr_kwargs, **synthesis_kwargs)

    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs['c_gen_conditioning_zero']:
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0), truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    def _planes(self, ws, update_emas=False, **synthesis_kwargs):
        planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        return planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])

    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                  use_cached_backbone=False, render_noise=None, skip_superresolution=False, sr_region_fn=None, depth_only=False,
                  **synthesis_kwargs):
        """sr_region_fn (extension, optional): called as ``sr_region_fn(out)`` with ``{'image_raw', 'image_depth'}`` once the renderer is
        done; returns a ``[N,1,512,512]`` mask of the image pixels the caller will look at (None = all).  The super-resolution convs
        then skip the output tiles that no such pixel depends on -- ``out['image']`` is the full forward's inside the mask and unspecified outside.
        depth_only (extension): return ``{'image_depth'}`` alone; the renderer skips the colour half of the decoder and of the composite."""
        cam2world = c[:, :16].view(-1, 4, 4)
        intrinsics = c[:, 16:25].view(-1, 3, 3)
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        ray_o, ray_d = self.ray_sampler(cam2world, intrinsics, neural_rendering_resolution)
        n = ray_o.shape[0]
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            with misc.trace_range('synthesis/backbone'):
                planes = self._planes(ws, update_emas=update_emas, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        if planes.shape[0] == 1 and n > 1:
            # one latent, several cameras (SPI's mirror / rot / depth branches repeat the same w per view, e.g.
            # rot_bbox_cx_coach.py:92): the tri-planes do not depend on the camera, so the backbone runs ONCE and the views
            # share its output; autograd sums the per-view plane gradients before the single backbone backward pass.
            planes = planes.expand(n, -1, -1, -1, -1)
        with misc.trace_range('synthesis/renderer'):
            feat, depth, _ = self.renderer(planes, self.decoder, ray_o, ray_d, self.rendering_kwargs, noise=render_noise, depth_only=depth_only)
        r = self.neural_rendering_resolution
        if depth_only:
            return {'image_depth': depth.permute(0, 2, 1).reshape(n, 1, r, r)}
        feature_image = feat.permute(0, 2, 1).reshape(n, feat.shape[-1], r, r).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(n, 1, r, r)
        rgb_image = feature_image[:, :3]
        out = {'image_raw': rgb_image, 'image_depth': depth_image}
        if not skip_superresolution:
            sr_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'noise_mode'}
            region = sr_region_fn(out) if sr_region_fn is not None else None
            if region is not None and hasattr(self.superresolution, 'needed_output_maps'):
                from ..torch_utils.ops import conv2d_mfma
                with conv2d_mfma.needed_output(self.superresolution.needed_output_maps(region)), misc.trace_range('synthesis/superresolution'):
                    out['image'] = self.superresolution(rgb_image, feature_image, ws,
                                                        noise_mode=self.rendering_kwargs['superresolution_noise_mode'], **sr_kwargs)
            else:
                with misc.trace_range('synthesis/superresolution'):
                    out['image'] = self.superresolution(rgb_image, feature_image, ws,
                                                        noise_mode=self.rendering_kwargs['superresolution_noise_mode'], **sr_kwargs)
        return out

    def sample(self, coordinates, directions, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.sample_mixed(coordinates, directions, ws, update_emas=update_emas, **synthesis_kwargs)

    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        planes = self._planes(ws, update_emas=update_emas, **synthesis_kwargs)
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)


def ffhq512_kwargs(narrow=False, depth_resolution=48, depth_resolution_importance=48):
    """Constructor arguments of the ffhqrebalanced512-128 architecture (what the EG3D pickle's init_kwargs hold;
    SURVEY.md 8d).  ``narrow`` shrinks the backbone widths for tests (SR widths are fixed by the module)."""
    cb, cm = (2048, 32) if narrow else (32768, 512)
    rk = dict(superresolution_module='training.superresolution.SuperresolutionHybrid8XDC', sr_antialias=True,
              superresolution_noise_mode='none', c_gen_conditioning_zero=False, c_scale=1.0, clamp_mode='softplus',
              disparity_space_sampling=False, decoder_lr_mul=1.0, box_warp=1, ray_start=2.25, ray_end=3.3,
              depth_resolution=depth_resolution, depth_resolution_importance=depth_resolution_importance, white_back=False)
    return dict(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=2),
                channel_base=cb, channel_max=cm, fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None,
                sr_num_fp16_res=4, sr_kwargs=dict(channel_base=cb, channel_max=cm, fused_modconv_default='inference_only'),
                rendering_kwargs=rk)
