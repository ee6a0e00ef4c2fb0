"""tokensgen_amd — MI355X-native (gfx950) implementation of the TokensGen denoising hot path.

Host side mirrors the reference's operator API (`longvgen.models`, `.schedulers`, `.fifo_sampling`);
all arithmetic runs in hand-written HIP kernels behind the C ABI in include/tokensgen_hip.h
(libtokensgen_hip.so).  There is no CPU or eager-PyTorch fallback: using an op without the built
library or without a GPU raises.
"""
__version__ = "0.1.0"
