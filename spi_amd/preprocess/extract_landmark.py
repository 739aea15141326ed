"""68 facial landmarks per crop (mirror of preprocess/extract_landmark.py:15-40).

The reference's detector is the third-party ``face_alignment`` package (an S3FD face detector + a 4-stack hourglass network with downloaded
weights, extract_landmark.py:11); neither its source nor its weights are in the reference tree.  ``face_alignment_detector()`` returns, in this
order: the pip package if it is importable (exactly the reference's callable); else this package's own restatement of the same two networks
on the MI355X conv kernels (``third_part/face_alignment``: S3FD + 2D-FAN-4, state-dict compatible with the package's downloaded weights, which
``paths_config.SFD_PATH`` / ``FAN_PATH`` must point to).  Any ``landmark_fn(PIL RGB image) -> float [68, 2]`` can be injected instead."""
import glob
import os

import numpy as np

_detector = None


def face_alignment_detector():
    """The reference's detector, if the package is importable (it is not a dependency of this package)."""
    try:
        import face_alignment
        kind = getattr(face_alignment.LandmarksType, 'TWO_D', None) or face_alignment.LandmarksType._2D
        det = face_alignment.FaceAlignment(kind)
        which = 'pip package face_alignment %s' % getattr(face_alignment, '__version__', '?')
    except ImportError:
        from ..third_part import face_alignment as own            # S3FD + 2D-FAN-4 on the HIP convs; raises FileNotFoundError without the weight files
        from ..configs import global_config
        det = own.FaceAlignment(own.LandmarksType._2D, device=global_config.device)
        which = 'spi_amd.third_part.face_alignment (S3FD + 2D-FAN-4 on the HIP convs; parity with the pip package unpinned)'
    import sys
    print(f'[spi_amd] landmark detector: {which}', file=sys.stderr)      # two machines must not differ silently in which detector ran

    def fn(image):
        lm = det.get_landmarks_from_image(np.array(image))
        assert lm is not None, 'No face detect error!'
        return lm[0]
    return fn


def get_landmark(image, landmark_fn=None):
    """PIL image -> [68, 2]"""
    global _detector
    if landmark_fn is None:
        if _detector is None:
            _detector = face_alignment_detector()
        landmark_fn = _detector
    lm = np.asarray(landmark_fn(image), dtype=np.float32)
    assert lm.shape == (68, 2), lm.shape
    return lm


def extract_landmark(input_dir, output_dir, mode='png', landmark_fn=None):
    """RGB, resized to 256^2, one [68,2] .npy per image (:27-40)"""
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    for image_path in sorted(glob.glob(f'{input_dir}/*.{mode}')):
        image = Image.open(image_path).convert('RGB').resize((256, 256))
        np.save(os.path.join(output_dir, os.path.basename(image_path).split('.')[0] + '.npy'), get_landmark(image, landmark_fn))
