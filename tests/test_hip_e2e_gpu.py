"""End-to-end run of the reference CLI on synthetic inputs (tiny step counts): stage 1 -> stage 2 -> checkpoint, images,
novel-view video, metrics; then `--G_1_type Inference` reloads the checkpoint."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_global_configs():
    """run_inversion writes the reference's module-level config objects; put them back for the tests that follow."""
    from spi_amd.configs import hyperparameters, paths_config, global_config
    mods = (hyperparameters, paths_config, global_config)
    saved = [{k: v for k, v in vars(m).items() if not k.startswith('__')} for m in mods]
    yield
    for m, sv in zip(mods, saved):
        for k in [k for k in vars(m) if not k.startswith('__') and k not in sv]:
            delattr(m, k)
        for k, v in sv.items():
            setattr(m, k, v)


def test_cli_pti_then_inference_and_metrics(tmp_path, capsys):
    from spi_amd import run_inversion
    from spi_amd.configs import hyperparameters as hp, paths_config
    out = str(tmp_path) + '/'
    hp.LPIPS_value_threshold = -1.0          # seeded random LPIPS weights give tiny distances: never early-stop here
    common = ['--output_root', out, '--synthetic', '1', '--not_use_wandb', '--depth_resolution', '12', '--depth_resolution_importance', '12']
    run_inversion.run(common + ['--first_inv_type', 'sgw+', '--first_inv_steps', '2', '--G_1_type', 'pti', '--G_1_step', '2'])
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')][-1]
    stats = json.loads(line)
    assert stats['images'] == 1 and stats['iterations'] == 4 and stats['n_gpus'] == 1
    coach = 'PTI_coach_sgw+_2_pti_2_rot_0_mirrorrot_0_depth_0_tv_0'
    ck = os.path.join(out, 'checkpoints', coach)
    names = [f[:-3] for f in os.listdir(ck) if f.endswith('.pt')]
    assert len(names) == 1
    ckpt = torch.load(os.path.join(ck, names[0] + '.pt'), map_location='cpu')
    assert set(ckpt) == {'w', 'c', 'G'} and ckpt['w'].shape == (1, 14, 512) and ckpt['c'].shape == (1, 25)
    assert os.path.isfile(os.path.join(out, 'image', coach, names[0] + '.jpg')) and os.path.isfile(os.path.join(out, 'image_m', coach, names[0] + '.jpg'))
    frames = os.path.join(out, 'video', coach, names[0] + '_frames')
    assert len(os.listdir(frames)) == 120
    emb = torch.load(os.path.join(out, 'embedding', coach, names[0] + '.pt'), map_location='cpu')
    assert emb.shape == (1, 14, 512)
    # reload through the Inference coach
    run_inversion.run(common + ['--G_1_type', 'Inference', '--load_embedding_coach_name', coach])
    assert len(os.listdir(os.path.join(out, 'video', names[0] + '_frames'))) == 120
    # metrics on the reloaded generator
    from spi_amd.training.coaches.pti_coach import SingleIDCoach
    from spi_amd.utils import load_utils
    from spi_amd.data.images_dataset import SyntheticDataset
    G = load_utils.load_eg3d(device='cuda:0', synthetic=True)
    c = SingleIDCoach(None, False, G=G)
    w, cam, Gl = c.load(os.path.join(ck, names[0] + '.pt'))
    d = SyntheticDataset(1)[0]
    gt = d['img'][None].cuda()
    with torch.no_grad():
        fake = Gl.synthesis(w, cam, noise_mode='const')['image']
    c.cal_metric(fake, gt, 'final', fake_m=torch.flip(fake, dims=[3]))
    paths_config.experiments_output_dir = str(tmp_path)
    c.log_metric()
    txt = open(os.path.join(str(tmp_path), 'metric_log.txt')).read()
    assert 'Mode: final AVG' in txt and 'Lpips M:' in txt and 'ID Sim: nan' in txt
