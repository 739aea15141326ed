"""Command line of the inversion (mirror of spi/run_inversion.py:16-128: same flags, defaults, side effects on
``hyperparameters`` / ``paths_config`` and output-directory layout).

Additions that do not exist in the reference: ``--synthetic N`` (seeded random-init generator + N synthetic
inputs), ``--depth_resolution`` / ``--depth_resolution_importance`` (renderer overrides; the
reference always runs what the pickle says), ``--network_pkl``.  Under ``torchrun`` each rank takes its block
of the image list (the reference's ``--dataset_block`` done in-process) and rank 0 prints the all-reduced
throughput statistics; the device follows LOCAL_RANK instead of the hard-wired CUDA_VISIBLE_DEVICES='0'
(run_inversion.py:109).
"""
import argparse
import json
import os
import time

import torch

from .configs import global_config, hyperparameters, paths_config


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Training')
    parser.add_argument('--data_root', type=str, default='/apdcephfs_cq2/share_1290939/feiiyin/dataset/CelebAHQ_test_camera/')
    parser.add_argument('--data_mode', type=str, default='png')
    parser.add_argument('--output_root', type=str, default=None)
    parser.add_argument('--use_encoder', action='store_true', default=False)
    parser.add_argument('--use_G_avg', action='store_true', default=False)
    parser.add_argument('--use_adapt_yaw_range', action='store_true', default=False)
    parser.add_argument('--not_use_wandb', action='store_true', default=False)
    parser.add_argument('--first_inv_type', type=str, default='pti')
    parser.add_argument('--first_inv_steps', type=int, default=500)
    parser.add_argument('--G_1_step', type=int, default=500)
    parser.add_argument('--G_1_type', type=str, default='space')
    parser.add_argument('--G_2_step', type=int, default=500)
    parser.add_argument('--load_embedding_coach_name', type=str, default=None)
    parser.add_argument('--pt_rot_lambda', type=float, default=0)
    parser.add_argument('--pt_mirror_rot_lambda', type=float, default=0)
    parser.add_argument('--pt_depth_lambda', type=float, default=0)
    parser.add_argument('--pt_tv_lambda', type=float, default=0)
    parser.add_argument('--description', type=str, default=None)
    parser.add_argument('--dataset_block', type=str, default=None, help='1/20')
    parser.add_argument('--select_range', type=int, default=None, help='100')
    parser.add_argument('--filter_index', type=str, default=None, help='1,2,3')
    # ---- additions ----
    parser.add_argument('--synthetic', type=int, default=0, help='use N synthetic inputs and a seeded random-init generator')
    parser.add_argument('--network_pkl', type=str, default=None)
    parser.add_argument('--depth_resolution', type=int, default=None)
    parser.add_argument('--depth_resolution_importance', type=int, default=None)
    parser.add_argument('--conv_precision', choices=('f32', 'bf16x6', 'bf16x3'), default=None,
                        help='dense-conv arithmetic: exact fp32 MFMA (f32) or fp32 operands split into 3 / 2 bf16 pieces on the bf16 matrix cores; '
                             'not given = keep global_config.conv_precision (f32 unless SPI_CONV_PRECISION says otherwise)')
    parser.add_argument('--sr_fp16', action='store_true', help='fp16 MFMA in the super-resolution blocks (BASELINE config 5); default fp32')
    args = parser.parse_args(argv)
    global_config.enable_fp16_blocks = bool(args.sr_fp16)
    if args.conv_precision is not None:                           # only an explicit flag overrides the SPI_CONV_PRECISION environment default
        global_config.conv_precision = {'f32': 0, 'bf16x6': 3, 'bf16x3': 2}[args.conv_precision]
    global_config.synthetic_weights = args.synthetic > 0          # seeded perceptual-loss weights only in synthetic mode (criteria/weights.py)

    for k in ('use_encoder', 'use_G_avg', 'first_inv_type', 'first_inv_steps', 'G_1_step', 'G_1_type', 'G_2_step',
              'load_embedding_coach_name', 'use_adapt_yaw_range', 'description', 'pt_rot_lambda', 'pt_mirror_rot_lambda',
              'pt_depth_lambda', 'pt_tv_lambda', 'depth_resolution', 'depth_resolution_importance'):
        setattr(hyperparameters, k, getattr(args, k))
    if args.network_pkl is not None:
        paths_config.EG3D_PATH = args.network_pkl
    if args.output_root is not None:
        paths_config.root = args.output_root
        paths_config.checkpoints_dir = paths_config.root + 'checkpoints/'
        paths_config.embedding_base_dir = paths_config.root + 'embedding/'
        paths_config.experiments_output_dir = paths_config.root + 'experiments/'
        paths_config.images_output_dir = paths_config.root + 'image/'
        paths_config.mirror_images_output_dir = paths_config.root + 'image_m/'
        paths_config.video_output_dir = paths_config.root + 'video/'
        for d in (paths_config.checkpoints_dir, paths_config.embedding_base_dir, paths_config.experiments_output_dir,
                  paths_config.images_output_dir, paths_config.mirror_images_output_dir, paths_config.video_output_dir):
            os.makedirs(d, exist_ok=True)
    return args


def build_dataset(args, rank=0, world_size=1):
    from .data.images_dataset import PTIDataset, SyntheticDataset
    from . import dist as sdist
    if args.synthetic > 0:
        dataset = SyntheticDataset(args.synthetic)
    else:
        root = args.data_root
        filt = args.filter_index.split(',') if args.filter_index is not None else None
        dataset = PTIDataset(source_root=os.path.join(root, 'crop'), c_root=os.path.join(root, 'c'), w_root=None,
                             mask_root=os.path.join(root, 'mask'), lm_root=os.path.join(root, 'lm'), target_name='target',
                             mode=args.data_mode, dataset_block=args.dataset_block, select_range=args.select_range, filter_index=filt)
    if world_size > 1:
        dataset = torch.utils.data.Subset(dataset, sdist.shard_indices(len(dataset), rank, world_size))
    loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False)
    return dataset, loader


def run(argv=None):
    from . import dist as sdist
    from .utils import load_utils
    args = parse_args(argv)
    use_wandb = not args.not_use_wandb
    rank, world, local = sdist.init_from_env()
    dev_index = sdist.device_index(local)                         # HIP_VISIBLE_DEVICES honoured (the reference pins CUDA_VISIBLE_DEVICES='0', :109)
    global_config.device = f'cuda:{dev_index}'
    torch.cuda.set_device(dev_index)
    sdist.pin_rank_affinity(local, sdist.local_world_size(world))      # own host cores per rank, next to its GPU
    sdist.reserve_allocator_pool(global_config.device)           # the measured path of bench.py: pool reserved once; the loops freeze the GC (quiet_gc)
    _, loader = build_dataset(args, rank, world)
    G = load_utils.load_eg3d(device=global_config.device, synthetic=args.synthetic > 0)
    from .training.coaches.pti_coach import SingleIDCoach
    from .training.coaches.rot_bbox_cx_coach import RotBboxCoach
    if args.G_1_type == 'pti':
        coach = SingleIDCoach(loader, use_wandb, G=G)
    elif args.G_1_type == 'RotBbox':
        coach = RotBboxCoach(loader, use_wandb, G=G)
    elif args.G_1_type == 'Inference':
        from .training.coaches.inference_coach import InferenceCoach
        coach = InferenceCoach(loader, use_wandb, G=G)
    else:
        raise NotImplementedError(args.G_1_type)
    sdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = coach.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iters = sum(s['iters'] + s.get('stage1_iters', 0) for s in stats)       # stage-1 steps count only when calc_inversions actually ran
    tot = sdist.reduce_stats([iters, len(stats)], device=global_config.device)
    tmax = sdist.reduce_stats([dt], device=global_config.device, op='max')[0]
    if rank == 0:
        import spi_amd
        gs = spi_amd.hip_graphs_status()
        print(json.dumps(dict(images=int(tot[1]), iterations=int(tot[0]), seconds=tmax, iters_per_sec=tot[0] / max(tmax, 1e-9), n_gpus=world,
                              hip_graphs=('replayed' if (gs['env'] and gs['self_test']) else 'off: eager iterations'), hip_graphs_status=gs,
                              per_image=[{k: s[k] for k in ('name', 'iters', 'stage1_iters', 'seconds_loop', 'seconds_outputs', 'stage2_graph_captures') if k in s}
                                         for s in stats])))
    return global_config.run_name


if __name__ == '__main__':
    run()
