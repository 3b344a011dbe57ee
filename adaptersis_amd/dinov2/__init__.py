"""HIP-backed mirror of the DINOv2 pieces on the AdapterSIS hot path (`dinov2/layers`, `dinov2/models`)."""
