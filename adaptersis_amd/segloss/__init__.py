"""HIP-backed mirror of the reference's ``segloss`` losses that the training scripts use."""
