"""Checkpoint and output locations (mirror of spi/configs/paths_config.py)."""
EG3D_PATH = 'checkpoints/ffhqrebalanced512-128.pkl'
IDLOSS_PATH = 'checkpoints/model_ir_se50.pth'
LPIPS_PATH = 'checkpoints/lpips_vgg_v0.1.pth'      # richzhang/PerceptualSimilarity lpips/weights/v0.1/vgg.pth ('' in the reference: it downloads the file)
VGG16_PATH = 'checkpoints/vgg16-397923af.pth'      # torchvision vgg16 state_dict (the reference: torchvision.models.vgg16(True))
VGG19_PATH = 'checkpoints/vgg19-dcbb9e9c.pth'      # torchvision vgg19 state_dict (the reference: torchvision vgg19(pretrained=True))
BISENET_PATH = 'checkpoints/bisenet.pth'
VGG_PATH = 'checkpoints/vgg16.pt'
SFD_PATH = 'checkpoints/s3fd-619a316812.pth'       # face_alignment's S3FD detector weights (the pip package downloads them; extract_landmark.py:10)
FAN_PATH = 'checkpoints/2DFAN4-11f355bf06.pth.tar'  # face_alignment's 2D-FAN-4 weights

root = 'test/output/'
checkpoints_dir = root + 'checkpoints/'
embedding_base_dir = root + 'embedding/'
experiments_output_dir = root + 'experiments/'
images_output_dir = root + 'image/'
mirror_images_output_dir = root + 'image_m/'
video_output_dir = root + 'video/'
