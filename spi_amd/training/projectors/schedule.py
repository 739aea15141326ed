"""Stage-1 learning-rate / W-noise schedule shared by the three projectors
(spi/training/projectors/mirror_projector.py:84-91; identical in w_projector / w_plus_projector)."""
import math


def stage1_schedule(step, num_steps, w_std, initial_learning_rate=0.01, initial_noise_factor=0.05, lr_rampdown_length=0.25,
                    lr_rampup_length=0.05, noise_ramp_length=0.75):
    """-> (lr, w_noise_scale): cosine ramp-down over the last 25 %, linear ramp-up over the first 5 %."""
    t = step / num_steps
    w_noise_scale = w_std * initial_noise_factor * max(0.0, 1.0 - t / noise_ramp_length) ** 2
    ramp = min(1.0, (1.0 - t) / lr_rampdown_length)
    ramp = 0.5 - 0.5 * math.cos(ramp * math.pi)
    ramp = ramp * min(1.0, t / lr_rampup_length)
    return initial_learning_rate * ramp, w_noise_scale
