"""helpers shared by the -m gpu tests: move synth batches to the GPU, compare with the oracle"""
import numpy as np
import torch

import oracle_lib as ol


def to_dev(a, dev):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def batch_to_dev(b, dev, weights=None):
    w = weights if weights is not None else b.weights
    return dict(ctrl=to_dev(b.ctrl, dev), guide_off=to_dev(b.guide_off, dev), guide_pv=to_dev(b.guide_pv, dev),
                guide_unk=to_dev(b.guide_unk, dev), obs_off=to_dev(b.obs_off, dev), obs=to_dev(b.obs, dev),
                weights=to_dev(w, dev))


def rel_err_per_traj(a, ref):
    B = ref.shape[0]
    return np.abs(a - ref).reshape(B, -1).max(1) / np.abs(ref).reshape(B, -1).max(1)


class emulation:
    """context manager: oracle in device-emulation mode for N control points"""

    def __init__(self, N):
        self.g, self.ppl = ol.emulation_shape(N)

    def __enter__(self):
        ol.set_emulation(self.g, self.ppl)

    def __exit__(self, *a):
        ol.set_emulation(0)
