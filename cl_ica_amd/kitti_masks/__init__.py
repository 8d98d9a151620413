"""KITTI-masks encoder and training-loop body with the reference's module layout
(/root/reference/kitti_masks/{model,solver}.py): conv stack on PyTorch-ROCm (MIOpen), as BASELINE.json config 5 prescribes;
final Linear, Softclip head and the Lp-InfoNCE loss on the HIP kernels."""
