"""HIP volumetric renderer (drop-in surface of eg3d/training/volumetric_rendering)."""
