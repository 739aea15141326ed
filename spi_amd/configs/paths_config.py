"""Checkpoint and output locations (mirror of spi/configs/paths_config.py)."""
EG3D_PATH = 'checkpoints/ffhqrebalanced512-128.pkl'
IDLOSS_PATH = 'checkpoints/model_ir_se50.pth'
LPIPS_PATH = ''
BISENET_PATH = 'checkpoints/bisenet.pth'
VGG_PATH = 'checkpoints/vgg16.pt'

root = 'test/output/'
checkpoints_dir = root + 'checkpoints/'
embedding_base_dir = root + 'embedding/'
experiments_output_dir = root + 'experiments/'
images_output_dir = root + 'image/'
mirror_images_output_dir = root + 'image_m/'
video_output_dir = root + 'video/'
