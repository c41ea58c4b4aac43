"""Operator layer of the fused DiffusionTransitionHead (host side of the HIP kernels)."""
