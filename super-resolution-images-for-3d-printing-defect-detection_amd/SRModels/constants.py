"""Patch / stride / scale constants of the reference experiments (reference: SRModels/constants.py:1-15).
Values are the reference's; the X4 block is the BASELINE.json configuration this build is measured on."""
RANDOM_SEED = 42

SRCNN_PATCH_SIZE, SRCNN_STRIDE = 24, 12
EDSR_PATCH_SIZE, EDSR_STRIDE, EDSR_SCALE_FACTOR = 24, 12, 2
ESRGAN_PATCH_SIZE, ESRGAN_STRIDE, ESRGAN_SCALE_FACTOR = 24, 12, 2
VGG_PATCH_SIZE, VGG_STRIDE = 96, 48

# BASELINE.json configs[2]: 4x ESRGAN on 512x512 LR tiles, super_resolve_image defaults (ESRGAN_model.py:858)
X4_SCALE_FACTOR = 4
X4_TILE = 512
X4_PATCH_SIZE_LR, X4_STRIDE = 48, 24
