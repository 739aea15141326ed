"""Face-parsing label -> face mask (mirror of spi/utils/mask_utils.py:4-9)."""
import torch

FACE_LABELS = (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13)


def calculate_face_mask(mask):
    """1 where the BiSeNet label is skin / brows / eyes / glasses-free face parts, same dtype as the input."""
    face = torch.zeros_like(mask)
    for att in FACE_LABELS:
        face += (mask == att)
    return face
