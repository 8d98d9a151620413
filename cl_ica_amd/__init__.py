"""cl_ica_amd -- MI355X-native (gfx950) implementation of cl-ica's contrastive training hot path.

Drop-in module names mirror the reference (`losses`, `encoders`, `layers`, `spaces`,
`latent_spaces`, `invertible_network_utils`); all arithmetic runs in hand-written HIP kernels
behind the C ABI in include/clica.h (cl_ica_amd/lib/libclica_hip.so, loaded with ctypes).
"""
__version__ = "0.1.0"

from .graphed import capture_train_step      # noqa: E402,F401  (HIP-graph replay of the reference's unchanged train_step closure)

