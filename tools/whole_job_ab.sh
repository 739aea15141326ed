#!/bin/bash
# whole-job CLI on 2 synthetic images (the bench's `whole_job` leg) with per-phase timestamps: where an image's set-up time goes
SPI_TIME_IMAGE=1 SPI_POOL_GIB=8 python -c "
from spi_amd.configs import hyperparameters as hp; hp.LPIPS_value_threshold = -1.0
from spi_amd import run_inversion
import tempfile
out = tempfile.mkdtemp(prefix='spi_wj_') + '/'
run_inversion.run(['--output_root', out, '--synthetic', '2', '--not_use_wandb', '--first_inv_type', 'mir', '--first_inv_steps', '120', '--G_1_type', 'RotBbox', '--G_1_step', '120', '--pt_rot_lambda', '0.1', '--pt_mirror_rot_lambda', '0.05', '--pt_depth_lambda', '1.0', '--depth_resolution', '96', '--depth_resolution_importance', '96'])
" 2>/dev/null | grep "^\[time\]"
