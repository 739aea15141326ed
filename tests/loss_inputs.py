"""Seeded inputs of the loss fixtures (golden/losses.npz): shared by tests/golden/make_golden.py (which feeds them to the REFERENCE's LPIPS /
BoxCXLoss) and by the tests (which feed the same tensors to the oracle and to the HIP path), so the 512^2 images need not be stored."""
import torch
import torch.nn.functional as F

_RAND, _RANDN = torch.rand, torch.randn          # bound at import: make_golden.py patches torch.rand while the reference runs


def noise_img(seed, n, res):
    """white noise in [-1, 1] (what SURVEY 8d's synthetic target is)"""
    return torch.rand(n, 3, res, res, generator=torch.Generator().manual_seed(seed)) * 2 - 1


def smooth_img(seed, n, lo_res, res=512):
    """low-pass images in [-1, 1]: crops with structure, as BoxCX sees them"""
    base = torch.rand(n, 3, lo_res, lo_res, generator=torch.Generator().manual_seed(seed))
    return F.interpolate(base, size=(res, res), mode='bicubic', align_corners=False).clamp(0, 1) * 2 - 1


def blob_mask(seed, n, res=512, thr=0.45):
    """a 0/1 visibility mask with large connected regions (the warp masks of the stage-2 branches look like this)"""
    base = torch.rand(n, 1, 16, 16, generator=torch.Generator().manual_seed(seed))
    return (F.interpolate(base, size=(res, res), mode='bilinear', align_corners=False) > thr).float()


def landmarks(seed, n):
    """the synthetic 68-point template, jittered per batch element; element 2 shifted, element 3's eyes pushed to the left border (its eye
    boxes reach outside the 256^2 frame: roi_align's out-of-range rule)"""
    from spi_amd.data.images_dataset import synthetic_landmarks
    lm = synthetic_landmarks()[None].repeat(n, 1, 1) + torch.randn(n, 68, 2, generator=torch.Generator().manual_seed(seed)) * 1.5
    if n > 2:
        lm[2] += torch.tensor([38.0, 52.0])
    if n > 3:
        lm[3, 36:48] -= torch.tensor([70.0, 0.0])
    return lm


def lpips_cases():
    """tag -> (x, y, mask or None): x is the differentiated input (multiplied by the mask before the loss when there is one)"""
    m4 = blob_mask(913, 4)
    return dict(lp1=(noise_img(901, 1, 512), noise_img(902, 1, 512), None),                       # stage-1 / main-view usage (incl. bilinear 512 -> 256)
                lp4=(smooth_img(903, 4, 24), smooth_img(904, 4, 20) * m4, m4),                    # the rot branch: batch 4, masked
                lp256=(noise_img(905, 2, 256), noise_img(906, 2, 256), None),                     # no resize branch
                lp64=(noise_img(907, 1, 64), noise_img(908, 1, 64), None))                        # deepest tap 4 x 4


def boxcx_cases():
    """tag -> (x, mask or None, y, lm): x is the differentiated input, multiplied by the mask before the loss when there is one -- the way the
    mirror-rot branch feeds it (`flip_gen_image * flip_warp_mask`, rot_bbox_cx_coach.py:127).  Inside a masked-out region every feature
    vector is identical, so the contextual loss's max / min run into EXACT ties there; which tied index receives the gradient differs
    between implementations (torch CPU: the first; the HIP kernels: their reduction order), but tied positions have fully masked receptive
    fields, so the difference is multiplied by the mask's zeros on its way to x -- differentiating through the mask is the meaningful test."""
    from spi_amd.data.images_dataset import synthetic_landmarks
    m4 = blob_mask(913, 4)
    xb, yb = smooth_img(909, 4, 40), smooth_img(910, 4, 40)
    yb = 0.7 * yb + 0.3 * xb                                        # correlated, like a render against its warped target
    return dict(bx4=(xb, m4, yb * m4, landmarks(911, 4)), bx1=(xb[:1].clone(), None, yb[:1].clone(), synthetic_landmarks()[None]))


def grad_sub(g):
    """the stored part of an input gradient: stride 4 at 512^2, stride 2 at 256^2, whole below"""
    s = {512: 4, 256: 2}.get(g.shape[-1], 1)
    return g[..., ::s, ::s]


def indexed_draw(j, shape, kind='rand', base=77000):
    """draw number j of a loop run under a counter-seeded stream: every draw has its own generator, so a test can re-create the stream from
    the list of shapes alone (the stage-2 fixtures would otherwise carry 1.5 MB of uniform noise per rendered image)"""
    g = torch.Generator().manual_seed(base + j)
    return (_RAND if kind == 'rand' else _RANDN)(*shape, generator=g)


def indexed_draws(shapes, kinds=None):
    return [indexed_draw(j, tuple(s), (kinds[j] if kinds else 'rand')) for j, s in enumerate(shapes)]


STAGE2_KEYS = ('backbone.synthesis.b64.conv1.weight', 'backbone.synthesis.b16.conv0.weight', 'superresolution.block1.conv1.weight',
               'superresolution.block0.conv0.affine.weight', 'decoder.net.0.weight', 'decoder.net.2.weight', 'backbone.synthesis.b8.torgb.bias',
               'backbone.synthesis.b32.conv1.noise_strength', 'superresolution.block1.torgb.weight', 'backbone.synthesis.b4.const')


def stage2_sub(k, t):
    """the stored part of a logged tensor (golden/trajectory_stage2.npz, trajectory_pti.npz keep the one large tensor at every 4th output channel)"""
    return t[::4] if k == 'superresolution.block1.conv1.weight' else t


class ReplayDraws:
    """feeds a recorded / re-created draw list to the oracle loop in order (shape-checked); `.log` = what was consumed"""
    def __init__(self, draws):
        self.d, self.pos, self.log = list(draws), 0, []

    def rand(self, *shape):
        t = self.d[self.pos]
        assert tuple(t.shape) == tuple(shape), (self.pos, tuple(t.shape), shape)
        self.pos += 1
        self.log.append(t)
        return t
    randn = rand


def golden_draws(g):
    """the random stream the reference's coach consumed when golden/trajectory_{stage2,pti}.npz was made, re-created from its shape list"""
    import json
    shapes = json.loads(str(g.z['draw_shapes'][0]))
    kinds = json.loads(str(g.z['draw_kinds'][0]))
    draws = indexed_draws(shapes, kinds)
    chk = torch.stack([d.double().sum() for d in draws])
    assert torch.equal(chk, g['draw_checksum']), 'the counter-seeded draw stream does not reproduce on this torch build'
    return draws


def oracle_stage2_run(P0, pnames, data, w_pivot, draws, n_iters, threshold, pti_only, W16, W19, opts, keys=STAGE2_KEYS):
    """the oracle's stage-2 / PTI loop (oracle/loops_ref.stage2_iteration) on a given draw stream -> per-iteration dicts with the loss values, the
    gradients Adam consumed and the parameters after the step; stops like the reference's loop"""
    from oracle import loops_ref as olp, losses_ref as olo
    st = olp.Stage2State(P0, pnames)
    rd = ReplayDraws(draws)
    mask = data['mask'].reshape(1, 1, 512, 512)
    od = dict(img=data['img'].reshape(1, 3, 512, 512), c=torch.as_tensor(data['c']).reshape(1, 25), lm=data['lm'].reshape(1, 68, 2),
              face_mask=olp.face_mask_from_parsing(mask).float())
    hp = dict(olp.HP, LPIPS_value_threshold=threshold)
    res = []
    for i in range(n_iters):
        o = olp.stage2_iteration(st, i, od, w_pivot, opts, lambda a, b: olo.lpips(W16, a, b), lambda a, b, lm: olo.box_cx_loss(W19, a, b, lm),
                                 hp=hp, nrr=128, draws=rd, pti_only=pti_only)
        o['grads'] = {k: (st.P[k].grad.detach().clone() if st.P[k].grad is not None else None) for k in keys}
        o['params'] = {k: st.P[k].detach().clone() for k in keys}
        res.append(o)
        if o.get('stopped'):
            break
    return res, rd, st
