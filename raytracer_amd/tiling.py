"""Row-band sharding of one frame across GPUs.

The reference treats image rows as independent rayon tasks (camera.rs:52-54); here rank
`part` of `n_parts` renders every n_parts-th band of `band_rows` rows (interleaved, so that
sky-vs-glass cost differences spread over the ranks), stores them compactly in local row
order, and one gather rebuilds the frame on the root.  The mapping below is exactly the one
rt_hip_render_rows implements (include/rt_hip.h):

    image_row(k) = ((k // band_rows) * n_parts + part) * band_rows + k % band_rows

Pure index arithmetic -- used by bench.py, the multi-GPU helper and the CPU (gloo) tests.
"""
import numpy as np

DEFAULT_BAND_ROWS = 16  # two 8-row wave tiles; keeps a wave's 8x8 pixel tile inside one band


def n_local_rows(vsize, part, n_parts, band_rows=DEFAULT_BAND_ROWS):
    """Local row slots rank `part` needs (whole bands; slots past the image bottom stay unwritten)."""
    n_bands = -(-vsize // band_rows)
    if part >= n_bands:
        return 0
    return -(-(n_bands - part) // n_parts) * band_rows


def padded_local_rows(vsize, n_parts, band_rows=DEFAULT_BAND_ROWS):
    """Row slots of the largest shard: every rank allocates this many so the gather is uniform."""
    return n_local_rows(vsize, 0, n_parts, band_rows)


def image_rows(vsize, part, n_parts, band_rows=DEFAULT_BAND_ROWS, n_slots=None):
    """image row of each local slot (int64 array, -1 where the slot falls outside the image)."""
    n = n_local_rows(vsize, part, n_parts, band_rows) if n_slots is None else n_slots
    k = np.arange(n, dtype=np.int64)
    y = ((k // band_rows) * n_parts + part) * band_rows + k % band_rows
    return np.where(y < vsize, y, -1)


def gather_permutation(vsize, n_parts, band_rows=DEFAULT_BAND_ROWS):
    """perm[y] = index into the concatenation of all ranks' padded shards that holds image row y."""
    slots = padded_local_rows(vsize, n_parts, band_rows)
    perm = np.full(vsize, -1, dtype=np.int64)
    for part in range(n_parts):
        rows = image_rows(vsize, part, n_parts, band_rows, n_slots=slots)
        ok = rows >= 0
        perm[rows[ok]] = part * slots + np.nonzero(ok)[0]
    assert (perm >= 0).all()
    return perm


def assemble(shards, vsize, n_parts, band_rows=DEFAULT_BAND_ROWS):
    """Rebuild the frame [vsize, w, 3] from the ranks' padded shards (numpy or torch, same code)."""
    perm = gather_permutation(vsize, n_parts, band_rows)
    if type(shards[0]).__module__.startswith("torch"):
        import torch
        allrows = torch.cat(list(shards), dim=0)
        return allrows.index_select(0, torch.as_tensor(perm, device=allrows.device))
    return np.concatenate(list(shards), axis=0)[perm]
