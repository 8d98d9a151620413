"""KITTI-masks encoder and training-loop body with the reference's module layout
(/root/reference/kitti_masks/{model,solver}.py): the five Conv2d + ReLU stages on the HIP conv stack (cl_ica_amd/conv.py,
``clica_conv_*``: forward, data gradients incl. the input-image gradient, weight gradients; ``CLICA_CONV=miopen`` keeps nn.Conv2d for A/B),
final Linear, Softclip head and the Lp-InfoNCE loss on the HIP kernels."""
