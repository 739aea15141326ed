"""Dataset preparation driver (mirror of preprocess/run_total.py:16-91): ``--input_root`` photos -> ``--output_root``/{input, crop, c, lm, mask}/
<name>/target.*, the layout ``PTIDataset`` reads (spi/data/images_dataset.py:102-147).

Of its three producers only the parsing mask is arithmetic this repository owns (BiSeNet on the MI355X conv kernels,
preprocess/extract_mask.py).  The other two wrap third-party trained networks that are neither in the reference tree nor installable
offline -- `face_alignment` (68 landmarks, preprocess/extract_landmark.py:11-24) and Deep3DFaceRecon + BFM (crop + camera,
preprocess/extract_camera.py:52-186) -- so they are INJECTED: ``landmark_fn(PIL image 256^2) -> float32 [68, 2]`` and
``camera_fn(image_path, crop_outdir, c_outdir, mode) -> None`` (must write ``target.<mode>`` into crop_outdir and ``target.npy`` (25 floats)
into c_outdir).  Without them the driver stops with an explanation instead of producing a partial dataset.
"""
import argparse
import glob
import os
import shutil

import numpy as np


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Training')
    parser.add_argument('--input_root', type=str, default='./test/images/')
    parser.add_argument('--output_root', type=str, default='./test/dataset/')
    parser.add_argument('--mode', type=str, default='jpg')
    return parser.parse_args(argv)


def extract_landmark(input_dir, output_dir, mode='png', landmark_fn=None):
    """preprocess/extract_landmark.py:27-40: RGB, resized to 256^2, one [68,2] .npy per image."""
    from PIL import Image
    if landmark_fn is None:
        raise RuntimeError('no landmark detector: the reference uses the third-party `face_alignment` package (extract_landmark.py:11); '
                           'pass landmark_fn(image_256) -> [68, 2]')
    os.makedirs(output_dir, exist_ok=True)
    for image_path in sorted(glob.glob(f'{input_dir}/*.{mode}')):
        image = Image.open(image_path).convert('RGB').resize((256, 256))
        lm = np.asarray(landmark_fn(image), dtype=np.float32)
        assert lm.shape == (68, 2), lm.shape
        np.save(os.path.join(output_dir, os.path.basename(image_path).split('.')[0] + '.npy'), lm)


def run(input_root, output_root, mode='jpg', camera_fn=None, landmark_fn=None, bisenet=None, device=None):
    from .extract_mask import extract_mask
    if camera_fn is None:
        raise RuntimeError('no crop / camera extractor: the reference uses Deep3DFaceRecon + BFM checkpoints (extract_camera.py:52-60); '
                           'pass camera_fn(image_path, crop_outdir, c_outdir, mode)')
    dirs = {k: os.path.join(output_root, k) for k in ('input', 'c', 'crop', 'lm', 'mask')}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    done = []
    for image_path in sorted(glob.glob(f'{input_root}/*.{mode}')):
        name = os.path.basename(image_path).split('.')[0]
        frame_root = os.path.join(dirs['input'], name)
        os.makedirs(frame_root, exist_ok=True)
        if not os.path.exists(os.path.join(dirs['crop'], name, f'target.{mode}')):
            shutil.copy(image_path, os.path.join(frame_root, f'target.{mode}'))
        crop_dir, c_dir = os.path.join(dirs['crop'], name), os.path.join(dirs['c'], name)
        os.makedirs(crop_dir, exist_ok=True)
        os.makedirs(c_dir, exist_ok=True)
        for f in sorted(glob.glob(f'{frame_root}/*.{mode}')):
            camera_fn(f, crop_dir, c_dir, mode)
        extract_landmark(crop_dir, os.path.join(dirs['lm'], name), mode=mode, landmark_fn=landmark_fn)
        extract_mask(crop_dir, os.path.join(dirs['mask'], name), mode=mode, bisenet=bisenet, device=device)
        done.append(name)
    return done


def main(argv=None):
    args = parse_args(argv)
    return run(args.input_root, args.output_root, args.mode)


if __name__ == '__main__':
    main()
