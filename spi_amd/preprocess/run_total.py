"""Dataset preparation driver (mirror of preprocess/run_total.py:16-91): ``--input_root`` photos -> ``--output_root``/{input, crop, c, lm, mask}/
<name>/target.*, the layout ``PTIDataset`` reads (spi/data/images_dataset.py:102-147).

Producers: the parsing mask (BiSeNet, preprocess/extract_mask.py) and the crop + camera (alignment, the Deep3DFaceRecon ResNet-50
regressor and the camera arithmetic, preprocess/extract_camera.py) run on the MI355X conv kernels of this package; their trained weights
(`bisenet.pth`, `checkpoints/model_name/epoch_20.pth`, the BFM landmark file) are third-party files the user supplies, as for the reference.
The 68-landmark detector is the third-party `face_alignment` package (preprocess/extract_landmark.py:11-24), neither in the reference tree
nor installable offline: ``landmark_fn(PIL image) -> float32 [68, 2]`` is injected (with the package installed the reference's own call is
built).  ``camera_fn(image_path, crop_outdir, c_outdir, mode)`` may replace the whole crop + camera step.  A missing file or detector stops
the driver with an explanation instead of producing a partial dataset.
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


def run(input_root, output_root, mode='jpg', camera_fn=None, landmark_fn=None, bisenet=None, device=None, model_paths=None, recon_state_dict=None,
        lm3d_std=None):
    from .extract_landmark import extract_landmark
    from .extract_mask import extract_mask
    if camera_fn is None:                                         # preprocess/run_total.py:40: one extractor for the whole run
        from .extract_camera import CameraExtractor
        extractor = CameraExtractor(None, None, None, model_paths=model_paths, landmark_fn=landmark_fn, device=device or 'cuda',
                                    state_dict=recon_state_dict, lm3d_std=lm3d_std)

        def camera_fn(path, crop_dir, c_dir, md):
            extractor.set_path(crop_outdir=crop_dir, c_outdir=c_dir, mode=md)
            extractor.extract(path)
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
