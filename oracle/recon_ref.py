"""Oracle (CPU, numpy / plain torch) for the crop + camera producer of the dataset preparation.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates, relative to /root/reference:
  image alignment          preprocess/extract_3dmm.py:15-102   (POS least squares, extract_5p, resize_n_crop_img, align_img)
  3DMM regressor           third_part/Deep3DFaceRecon_pytorch/models/networks.py:69-104 (ReconNetWrapper: ResNet-50 v1.5 trunk :205-370 with the
                           stride on the 3x3 conv, global average pool, seven 1x1 heads -> 257 coefficients), models/bfm.py:252-273 (split_coeff)
  camera from coefficients preprocess/extract_camera.py:14-47 (compute_rotation), :83-138 (cal_camera), preprocess/process_camera.py:9-62
  mirror camera            preprocess/extract_camera.py:161-176
Pinning (tests/golden/make_golden.py): section `recon` -- the NETWORK against the imported reference class `ReconNetWrapper` (fp32 round-off)
and `process_camera` against the imported preprocess/process_camera.py (bit-exact); section `preprocess` (round 4) -- `pos`, `extract_5p`,
`align_img` (both rescale factors), `compute_rotation`, `cal_camera`, `mirror_camera` against the reference's own preprocess/extract_3dmm.py and
extract_camera.py functions, bit-exact (`golden/preprocess.npz`).  Those modules import `face_alignment`, `cv2`, `skimage`, `kornia`,
`torchvision` at module level and instantiate the landmark detector; none of that is used by the functions above, so the imports resolve to
empty placeholder modules for the run (the technique sections `bisenet` / `recon` already use for `torchvision` / `kornia`).
The landmark detector itself (`face_alignment`, preprocess/extract_landmark.py:11-24) is a third-party package that is not in the reference
tree: landmarks are an input here ("parity unpinned" at that edge).
"""
import numpy as np
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
HEADS = (80, 64, 80, 3, 27, 2, 1)            # id, exp, tex, angle, gamma, (tx, ty), tz  (networks.py:84-92)


# ---- alignment (numpy / PIL) ---------------------------------------------------------------------------------------------------------

def pos(xp, x):
    """Least-squares similarity of 3D points x [3,n] onto 2D points xp [2,n] -> (t [2,1], s)   (extract_3dmm.py:15-39)."""
    npts = xp.shape[1]
    A = np.zeros([2 * npts, 8])
    A[0:2 * npts - 1:2, 0:3] = x.transpose()
    A[0:2 * npts - 1:2, 3] = 1
    A[1:2 * npts:2, 4:7] = x.transpose()
    A[1:2 * npts:2, 7] = 1
    b = np.reshape(xp.transpose(), [2 * npts, 1])
    k, _, _, _ = np.linalg.lstsq(A, b, rcond=None)               # (the reference's bare call under numpy >= 2)
    R1, R2 = k[0:3], k[4:7]
    s = (np.linalg.norm(R1) + np.linalg.norm(R2)) / 2
    t = np.stack([k[3], k[7]], axis=0)
    return t, s


def extract_5p(lm):
    idx = np.array([31, 37, 40, 43, 46, 49, 55]) - 1
    lm5p = np.stack([lm[idx[0], :], np.mean(lm[idx[[1, 2]], :], 0), np.mean(lm[idx[[3, 4]], :], 0), lm[idx[5], :], lm[idx[6], :]], axis=0)
    return lm5p[[1, 2, 0, 3, 4], :]


def lm3d_from_68(lm3d_68):
    """util/load_mats.py:105-116: the five standard 3D landmarks from the 68 of similarity_Lm3D_all.mat."""
    return extract_5p(lm3d_68)


def resize_n_crop_img(img, lm, t, s, target_size=1024.):
    from PIL import Image
    w0, h0 = img.size
    t = [float(np.ravel(t)[0]), float(np.ravel(t)[1])]
    w = (w0 * s).astype(np.int32)
    h = (h0 * s).astype(np.int32)
    left = (w / 2 - target_size / 2 + float((t[0] - w0 / 2) * s)).astype(np.int32)
    right = left + target_size
    up = (h / 2 - target_size / 2 + float((h0 / 2 - t[1]) * s)).astype(np.int32)
    below = up + target_size
    img = img.resize((w, h), resample=Image.LANCZOS)
    img = img.crop((left, up, right, below))
    lm = np.stack([lm[:, 0] - t[0] + w0 / 2, lm[:, 1] - t[1] + h0 / 2], axis=1) * s
    lm = lm - np.reshape(np.array([(w / 2 - target_size / 2), (h / 2 - target_size / 2)]), [1, 2])
    return img, lm


def align_img(img, lm, lm3d, target_size=1024., rescale_factor=466.285):
    """-> (trans_params [5], 224^2 PIL image, landmarks at 224 scale, 1024^2 PIL image)   (extract_3dmm.py:67-102)."""
    from PIL import Image
    w0, h0 = img.size
    lm5p = extract_5p(lm) if lm.shape[0] != 5 else lm
    t, s = pos(lm5p.transpose(), lm3d.transpose())
    s = rescale_factor / s
    img_new, lm_new = resize_n_crop_img(img, lm, t, s, target_size=target_size)
    trans_params = np.array([w0, h0, float(s), float(np.ravel(t)[0]), float(np.ravel(t)[1])])      # (the reference's ragged np.array([.., t[0], t[1]]) is an error under numpy >= 1.24)
    lm_new = lm_new * (224 / 1024.0)
    return trans_params, img_new.resize((224, 224), resample=Image.LANCZOS), lm_new, img_new


# ---- network -------------------------------------------------------------------------------------------------------------------------

def synthetic_state_dict(manifest, seed=0):
    """Seeded weights keyed by name (the trained epoch_20.pth does not exist offline): He-style conv weights, BatchNorm statistics away from
    the identity so that folding errors would show, heads with small weights (the reference zero-initialises them, which would pin nothing)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in manifest.items():
        leaf = k.split('.')[-1]
        if leaf == 'num_batches_tracked':
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif leaf == 'running_var':
            sd[k] = 0.5 + torch.rand(shape, generator=g)
        elif leaf == 'running_mean':
            sd[k] = 0.2 * torch.randn(shape, generator=g)
        elif leaf == 'bias':
            sd[k] = 0.1 * torch.randn(shape, generator=g)
        elif leaf == 'weight' and len(shape) == 1:
            sd[k] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.5 if k.startswith('final_layers') else (0.7 if 'conv3' in k else 2.0 ** 0.5)       # residual branches tamed: 16 blocks deep
            sd[k] = torch.randn(shape, generator=g) * gain / fan_in ** 0.5
    return sd


def _bn(P, pfx, x):
    return F.batch_norm(x, P[pfx + 'running_mean'], P[pfx + 'running_var'], P[pfx + 'weight'], P[pfx + 'bias'], False, 0.0, 1e-5)


def _bottleneck(P, pfx, x, stride):
    out = F.relu(_bn(P, pfx + 'bn1.', F.conv2d(x, P[pfx + 'conv1.weight'])))
    out = F.relu(_bn(P, pfx + 'bn2.', F.conv2d(out, P[pfx + 'conv2.weight'], None, stride, 1)))
    out = _bn(P, pfx + 'bn3.', F.conv2d(out, P[pfx + 'conv3.weight']))
    sc = x
    if pfx + 'downsample.0.weight' in P:
        sc = _bn(P, pfx + 'downsample.1.', F.conv2d(x, P[pfx + 'downsample.0.weight'], None, stride))
    return F.relu(out + sc)


def recon_net(P, x):
    """x [B,3,224,224] in [0,1] -> coefficients [B,257]."""
    b = 'backbone.'
    x = F.relu(_bn(P, b + 'bn1.', F.conv2d(x, P[b + 'conv1.weight'], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(LAYERS, start=1):
        for bi in range(n):
            x = _bottleneck(P, f'{b}layer{li}.{bi}.', x, 2 if (bi == 0 and li > 1) else 1)
    x = F.adaptive_avg_pool2d(x, (1, 1))
    out = [F.conv2d(x, P[f'final_layers.{i}.weight'], P[f'final_layers.{i}.bias']) for i in range(len(HEADS))]
    return torch.flatten(torch.cat(out, dim=1), 1)


def split_coeff(c):
    return {'id': c[:, :80], 'exp': c[:, 80:144], 'tex': c[:, 144:224], 'angle': c[:, 224:227], 'gamma': c[:, 227:254], 'trans': c[:, 254:]}


# ---- camera --------------------------------------------------------------------------------------------------------------------------

def compute_rotation(angles):
    """angles [B,3] (torch, radians) -> the FIRST sample's rotation, transposed   (extract_camera.py:14-47)."""
    b = angles.shape[0]
    ones, zeros = torch.ones([b, 1]), torch.zeros([b, 1])
    x, y, z = angles[:, :1], angles[:, 1:2], angles[:, 2:]
    rx = torch.cat([ones, zeros, zeros, zeros, torch.cos(x), -torch.sin(x), zeros, torch.sin(x), torch.cos(x)], dim=1).reshape([b, 3, 3])
    ry = torch.cat([torch.cos(y), zeros, torch.sin(y), zeros, ones, zeros, -torch.sin(y), zeros, torch.cos(y)], dim=1).reshape([b, 3, 3])
    rz = torch.cat([torch.cos(z), -torch.sin(z), zeros, torch.sin(z), torch.cos(z), zeros, zeros, zeros, ones], dim=1).reshape([b, 3, 3])
    return (rz @ ry @ rx).permute(0, 2, 1)[0]


def cal_camera(angle, trans0):
    """angle [1,3] torch, trans0 [3] torch (MODIFIED in place like the reference: trans[2] += -10) -> dict(intrinsics, pose, angle)   (:83-138)."""
    R = compute_rotation(angle).numpy()
    trans0[2] += -10
    c = -np.dot(R, trans0.numpy())
    pose = np.eye(4)
    pose[:3, :3] = R
    c *= 0.27
    c[1] += 0.006
    c[2] += 0.161
    pose[0, 3], pose[1, 3], pose[2, 3] = c[0], c[1], c[2]
    focal, w, h = 2985.29, 1024, 1024
    K = np.eye(3)
    K[0][0] = focal
    K[1][1] = focal
    K[0][2] = w / 2.0
    K[1][2] = h / 2.0
    rot = np.eye(3)
    rot[1, 1] = -1
    rot[2, 2] = -1
    pose[:3, :3] = np.dot(pose[:3, :3], rot)
    return {'intrinsics': K.tolist(), 'pose': pose.tolist(), 'angle': (angle * torch.tensor([1, -1, 1])).flatten().tolist()}


def process_camera(pose, intrinsics):
    """process_camera.py:9-62 (mode 'orig'): camera centre scaled to radius 2.7, normalised intrinsics -> 25 floats."""
    pose = np.array(pose).copy()
    pose[:3, 3] = pose[:3, 3] / np.linalg.norm(pose[:3, 3]) * 2.7
    K = np.array(intrinsics).copy()
    K[0, 0] = 2985.29 / 700
    K[1, 1] = 2985.29 / 700
    K[0, 2] = 1 / 2
    K[1, 2] = 1 / 2
    return np.concatenate([pose.reshape(-1), K.reshape(-1)])


def mirror_camera(c):
    """extract_camera.py:161-176."""
    pose, K = c[:16].reshape(4, 4).copy(), c[16:].reshape(3, 3)
    for i, j in ((0, 1), (0, 2), (1, 0), (2, 0), (0, 3)):
        pose[i, j] *= -1
    return np.concatenate([pose.reshape(-1), K.reshape(-1)])
