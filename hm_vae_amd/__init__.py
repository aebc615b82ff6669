"""hm_vae_amd — MI355X-native (gfx950) implementation of the hm-vae training hot path.

Module names mirror the reference's flat files (skeleton, my_tools, fk_layer, seq_two_hier_sa_vae,
trainer_motion_vae, utils_common) so that `from hm_vae_amd.trainer_motion_vae import Trainer` is the drop-in for
`from trainer_motion_vae import Trainer`.  Compute runs in hand-written HIP kernels (csrc/) behind the C ABI in
include/hmvae.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
