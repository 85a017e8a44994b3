"""Batch sharding for the one-process-per-GPU layout: trajectories are
independent, so rank r owns the contiguous slice shard_range(B, r, world_size) and the only
collective is the broadcast of the packed voxel snapshot."""
import numpy as np

from .synth import Batch


def shard_range(B: int, rank: int, world_size: int):
    """contiguous, disjoint, covering; sizes differ by at most one"""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_batch(b: Batch, lo: int, hi: int) -> Batch:
    """trajectories [lo, hi) with their CSR guide / obstacle arrays re-based to zero"""
    N = b.N
    g0, g1 = int(b.guide_off[lo * N]), int(b.guide_off[hi * N])
    goff = (b.guide_off[lo * N:hi * N + 1] - g0).astype(np.int32)
    obs_off = obs = None
    if b.obs_off is not None:
        o0, o1 = int(b.obs_off[lo]), int(b.obs_off[hi])
        obs_off = (b.obs_off[lo:hi + 1] - o0).astype(np.int32)
        obs = np.ascontiguousarray(b.obs[o0:o1])
    elif b.obs is not None:
        obs = b.obs
    w = None if b.weights is None else np.ascontiguousarray(b.weights[lo:hi])
    return Batch(np.ascontiguousarray(b.ctrl[lo:hi]), goff, np.ascontiguousarray(b.guide_pv[g0:g1]),
                 np.ascontiguousarray(b.guide_unk[g0:g1]), obs_off, obs, w, dict(b.meta))


def packed_words(nx: int, ny: int, nz: int) -> int:
    return 3 * nx * ny * ((nz + 31) // 32)


def pack_grid_reference(voxels: np.ndarray) -> np.ndarray:
    """numpy statement of the snapshot format vigo_pack_grid produces (three bit planes, z
    fastest, bit k of word w = voxel z = 32 w + k) — used by CPU tests of the broadcast path."""
    nx, ny, nz = voxels.shape
    nzw = (nz + 31) // 32
    pad = np.zeros((nx, ny, nzw * 32), dtype=np.uint8)
    pad[:, :, :nz] = voxels
    planes = []
    for bit in range(3):
        bits = ((pad >> bit) & 1).reshape(nx, ny, nzw, 32).astype(np.uint64)
        words = (bits << np.arange(32, dtype=np.uint64)).sum(axis=3).astype(np.uint32)
        planes.append(words.reshape(-1))
    return np.concatenate(planes).view(np.int32)
