"""Shared helpers for the parity tests (loading golden cases, tolerances)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEAD_CASES = ["tiny_l2", "tiny_l1_odd", "tiny_l4", "s8_h64", "clamp", "ou_dims", "lv_dims",
              "h80_l2", "h130_l1_s10", "h96_l3",  # these three: hidden_dim > 64 / state_dim > 9 (generic kernels)
              "s5_h32_l1", "s9_h64_l2"]           # 16 < emission rows <= 64: wide variant of the register-resident kernels
# batches above the dispatcher's thresholds (forward > 256 paths, backward > 640): the multi-path MFMA kernels under DEFAULT dispatch
MP_CASES = ["mp_b300_s2", "mp_b700_s1"]
W_NAMES = ["W_ih_l0", "W_hh_l0", "b_ih_l0", "b_hh_l0", "W_ih_stack", "W_hh_stack",
           "b_ih_stack", "b_hh_stack", "out_weight", "out_bias"]
G_NAMES = ["x0", "context", "sde_parameters"] + W_NAMES


def load_head_case(name):
    d = dict(np.load(os.path.join(GOLDEN, f"head_{name}.npz")))
    B, T, S, C, P, H, L = (int(v) for v in d["dims"])
    d["B"], d["T"], d["S"], d["C"], d["P"], d["H"], d["L"] = B, T, S, C, P, H, L
    for n in W_NAMES[4:8]:
        if L == 1:  # reference passes empty [0, 3H, H] stacks for a single layer (head.py:137-146)
            d["w_" + n] = d["w_" + n].reshape((0,) + d["w_" + n].shape[1:])
    return d


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
