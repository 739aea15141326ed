"""Run-wide settings (mirror of spi/configs/global_config.py).  ``device`` follows LOCAL_RANK instead of
being pinned to 'cuda:0' so that one process per GPU works under torchrun."""
import os

cuda_visible_devices = os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES', '0'))
device = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', 0))

training_step = 1
log_snapshot = 500
pivotal_training_steps = 0
model_snapshot_interval = 400
run_name = ''

# perceptual-loss weights: False = read the pretrained files named in paths_config and fail if one is missing (what a real run
# needs); True = seeded stand-ins (offline benchmark / tests; set by `--synthetic`).  See criteria/weights.py.
synthetic_weights = False

# fp16 MFMA in the generator blocks that the checkpoint marks `use_fp16` (the super-resolution blocks: sr_num_fp16_res = 4).
# Off by default: fp32 everywhere reproduces the reference's CPU path, the parity target.  `--sr_fp16` / BASELINE config 5.
enable_fp16_blocks = False
# with enable_fp16_blocks: the blocks' activations are fp16 TENSORS in HBM (the reference's use_fp16 path), not only fp16 MFMA operands (round 5)
fp16_storage = os.environ.get('SPI_FP16_STORAGE', '1') != '0'

# stage 2: run the rot / mirror-rot / depth branches on their own HIP streams beside the main backward (rot_bbox_cx_coach.py).
# Measured neutral on one MI355X (169.5 vs 169.3 ms per super-cycle: the big kernels of every chain fill the chip on their own
# and the host enqueues the chains one after the other anyway), so it stays off.
concurrent_branches = False

# arithmetic of the dense convolutions (spi_conv_desc.compute_f16 of every conv that does not ask for fp16):
#   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TF peak) -- the default.  Bit-for-bit fp32 FMA chains ONLY together with
#      conv_winograd = False (SPI_CONV_WINOGRAD=0): with the Winograd default below, the forward / dgrad / weight-gradient passes of the 3x3 layers
#      are fp32 minimal filtering, which rounds differently from the direct sums (a few fp32 ulps; tested to 1e-5 of the tensor maximum)
#   3  fp32 operands split into THREE bf16 pieces, the six significant piece products on the bf16 matrix cores with fp32 accumulation
#      (error ~2^-23 per product: fp32-level; 2.7x less matrix-pipe time).  With conv_winograd on, the 3x3 forward / dgrad / weight-gradient passes
#      of this mode are served by the fp32 Winograd kernels (at least as precise, and faster there); SPI_CONV_WINOGRAD=0 forces the split
#      kernels everywhere
#   2  two pieces, three products (error ~2^-16 per product; 5.3x less matrix-pipe time)
# `--conv_precision {f32,bf16x6,bf16x3}` / `bench.py --conv-precision`.
conv_precision = {'f32': 0, 'bf16x6': 3, 'bf16x3': 2}[os.environ.get('SPI_CONV_PRECISION', 'f32')]   # env override: run any test / tool in a split mode

# data-driven skipping of exactly-zero gradients / unneeded super-resolution tiles in the masked pseudo-view branches.
# Results are equal either way (tested); False = dense bound: every ray, gradient segment and SR tile is processed (`bench.py --dense`).
exploit_sparsity = True

# Winograd F(2x2, 3x3) for the 3x3 / stride-1 forward and data-gradient passes of the large layers and F(3x3, 2x2) for their weight
# gradients (winograd.hip): fp32 operands and accumulation, 2.25x fewer MFMAs; results differ from the direct sums by a few fp32 roundings.  Off = implicit GEMM everywhere.
conv_winograd = os.environ.get('SPI_CONV_WINOGRAD', '1') != '0'
# conv_winograd_f4 (round 6; SPI_CONV_WINO_F4=0 switches it off): the >= 256^2 layers' Winograd forward / dgrad use F(4x4, 3x3) -- 36 multiplications per
#      4x4 output tile and channel pair, 1.78x fewer than F(2x2, 3x3) -- where its 16 x 32-pixel blocks fill the chip; fp32 operands and accumulation, the
#      result within 4e-5 of the tensor's range of the direct sum (F(2x2): 2e-6).  False: F(2x2, 3x3) everywhere (`bench.py` reports that rate beside `value`).
conv_winograd_f4 = os.environ.get('SPI_CONV_WINO_F4', '1') != '0'

# fp16 activation tensors (configs[4]): the 3x3 / stride-1 forward and data-gradient passes whose output channels come in blocks of 128 run the
# direct fp16 kernel (hconv.hip: the input patch staged once for all nine taps, weights pre-converted to their LDS image).  Off = implicit GEMM.
conv_direct_fp16 = os.environ.get('SPI_CONV_DIRECT_FP16', '1') != '0'

# frozen-weight modulated convs (stage 1): the <dz, z> dot product of the style gradient comes out of the layer-tail backward pass (spi_tail_bwd_dot_t)
# instead of a second pass over dz and y.  Off = the separate spi_chan_dot launch.
fuse_tail_dot = os.environ.get('SPI_FUSE_TAIL_DOT', '1') != '0'

# stage 1: capture the projector step in a HIP graph after an eager warm-up step and replay it (projectors/common.py).  The step is
# GPU-bound either way; the graph takes the ~10 ms of host enqueue work per step off the CPU.  Off: every step is enqueued eagerly.
stage1_hip_graph = os.environ.get('SPI_STAGE1_GRAPH', '1') != '0'

# stage 2: the same for the RotBbox iteration (rot_bbox_cx_coach.py): one graph for the plain iteration, one for the iteration with the
# rot / mirror-rot / depth branches.  ON by default since round 4 (SPI_STAGE2_GRAPH=0 switches it off).  It replays correctly at full size since round 3 (BoxCX uses amin / amax:
# the index scatter of torch.min's backward faulted under replay; test_stage2_hip_graph_replay_equals_eager_iterations_full_size) and the
# replays are PIPELINED: the early-stop comparison is ORed into a sticky device byte, the Adam launch is predicated on it, and the host
# reads the byte of iteration i - 2 when it launches iteration i.  Measured: 35.8 vs 35.4 it/s eager in steady state, 38.8 vs 39.3 over a
# whole 1 500-iteration job with its two captures (the first version, which read the flag after every replay, ran 33.0 vs 35.0) -- the
# iteration is GPU-bound: the graph buys HOST time (one graph launch instead of ~25 ms of single-thread enqueue work per iteration), which is
# what eight ranks sharing one host's cores need (dist.pin_rank_affinity), and a little throughput (round 4: 45.46 vs 45.31 it/s over 300 steps).
# Draw sources that are not the device generator (tests replaying recorded draws) and concurrent_branches keep the eager iteration.
stage2_hip_graph = os.environ.get('SPI_STAGE2_GRAPH', '1') != '0'

# Round 5: graphs bake in device ADDRESSES, not values -- the per-image inputs of both loops (target image, masks, cameras, LPIPS features of the
# target, pivot latent, the frozen generator's tri-planes, the projector's w / noise maps / Adam state) live in persistent per-coach buffers that
# the next image overwrites in place, so image k >= 2 replays image 1's graphs: no eager warm-up iterations, no captures, no generator deep copy
# per image (the whole job used to run 8 % below the benchmark rate).  SPI_REUSE_GRAPHS=0: per-image captures as in round 4.
reuse_graphs_across_images = os.environ.get('SPI_REUSE_GRAPHS', '1') != '0'

# host side: freeze Python's garbage collector state around the optimisation loops (torch_utils/misc.quiet_gc): a full collection over the
# whole heap costs 50-80 ms = two or three iterations whenever it strikes inside a loop.
freeze_gc_in_loops = os.environ.get('SPI_GC_FREEZE', '1') != '0'
