"""HIP-backed mirror of the reference's ``backbones`` package (adapter blocks, CNN encoder, decode heads)."""
