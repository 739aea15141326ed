"""Soak run of the real CLI at the benchmark size (full-width generator, 512^2, 96+96): 120 stage-1 steps (replayed from the HIP graph) +\n120 RotBbox iterations incl. checkpoint / embedding / image output; prints the run statistics line."""
import json, os, sys, time, io, contextlib
sys.path.insert(0, '.')
from spi_amd import run_inversion
from spi_amd.configs import hyperparameters as hp
hp.LPIPS_value_threshold = -1.0
if os.environ.get('PHASES'):                                  # wall time of the phases of one image (device-synchronised): stage 1, stage 2, outputs
    import torch
    from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach

    def timed(name):
        inner = getattr(RotBboxCoach, name)

        def wrapper(self, *a, **k):
            torch.cuda.synchronize(); t = time.time()
            out = inner(self, *a, **k)
            torch.cuda.synchronize()
            print(f'[phase] {name}: {time.time() - t:.2f} s', file=sys.stderr, flush=True)
            return out
        setattr(RotBboxCoach, name, wrapper)
    for n_ in ('prepare_image', 'restart_training', 'get_inversion', 'optimise_image', 'finish_image', 'post_process'):
        timed(n_)
t0 = time.time()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    run_inversion.run(['--output_root', '/tmp/soak_out/', '--synthetic', '1', '--not_use_wandb', '--first_inv_type', 'mir', '--first_inv_steps', os.environ.get('STEPS1', '120'),
                       '--G_1_type', 'RotBbox', '--G_1_step', os.environ.get('STEPS2', '120'), '--depth_resolution', '96', '--depth_resolution_importance', '96',
                       '--pt_rot_lambda', '0.1', '--pt_mirror_rot_lambda', '0.05', '--pt_depth_lambda', '1.0'])      # the README's RotBbox command (BASELINE configs[1])
line = [l for l in buf.getvalue().splitlines() if l.startswith('{')][-1]
print('wall', round(time.time() - t0, 1), 's', line[:600])
