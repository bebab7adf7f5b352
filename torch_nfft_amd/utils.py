"""Point pre-processing: centring and scaling point sets into the torus (reference: ``torch_nfft/utils.py``).

The batched variants of the reference need the third-party ``torch_scatter``; here they use
``Tensor.scatter_reduce_`` (amin / amax), so batches work out of the box."""
import torch


def _resolve(source_batch, target_batch, batch):
    if batch is not None:
        return batch, batch
    return source_batch, target_batch


def _segment_reduce(values, index, size, mode):
    """Per-point-set min / max of ``values`` [n, ...] over the sorted set index ``index`` [n]."""
    out = values.new_empty((size,) + tuple(values.shape[1:]))
    idx = index.reshape((-1,) + (1,) * (values.dim() - 1)).expand_as(values)
    return out.scatter_reduce_(0, idx, values, mode, include_self=False)


def compute_points_center(sources, targets=None, source_batch=None, target_batch=None, /, batch=None):
    """Centre of the bounding box of every point set (utils.py:4-28)."""
    source_batch, target_batch = _resolve(source_batch, target_batch, batch)
    if source_batch is None:
        lo, hi = sources.min(dim=0).values, sources.max(dim=0).values
        if targets is not None:
            lo = torch.minimum(lo, targets.min(dim=0).values)
            hi = torch.maximum(hi, targets.max(dim=0).values)
    else:
        B = int(source_batch[-1].item()) + 1
        lo = _segment_reduce(sources, source_batch, B, "amin")
        hi = _segment_reduce(sources, source_batch, B, "amax")
        if targets is not None:
            lo = torch.minimum(lo, _segment_reduce(targets, target_batch, B, "amin"))
            hi = torch.maximum(hi, _segment_reduce(targets, target_batch, B, "amax"))
    return 0.5 * (lo + hi)


def shift_points_by_center(sources, targets=None, source_batch=None, target_batch=None, /, batch=None):
    """Translate every point set so that its bounding box is centred at the origin (utils.py:31-43)."""
    source_batch, target_batch = _resolve(source_batch, target_batch, batch)
    center = compute_points_center(sources, targets, source_batch, target_batch)
    sources = sources - (center if source_batch is None else center[source_batch])
    if targets is not None:
        targets = targets - (center if target_batch is None else center[target_batch])
    return sources, targets


def _point_norms(points, norm):
    if norm == "euclidean":
        return torch.sum(points ** 2, dim=1)  # squared; the caller takes the root
    if norm == "infinity":
        return points.abs().max(dim=1).values
    raise ValueError(f"scale_points_by_norm received unknown norm: {norm}")


def compute_points_radius(sources, targets=None, source_batch=None, target_batch=None, /, batch=None,
                          norm="euclidean"):
    """Largest point norm per point set: a float without batches, a [B] tensor with them (utils.py:46-82)."""
    source_batch, target_batch = _resolve(source_batch, target_batch, batch)
    finish = (lambda r: r.sqrt()) if norm == "euclidean" else (lambda r: r)
    if source_batch is None:
        r = _point_norms(sources, norm).max()
        if targets is not None:
            r = torch.maximum(r, _point_norms(targets, norm).max())
        return finish(r).item()
    B = int(source_batch[-1].item()) + 1
    r = _segment_reduce(_point_norms(sources, norm), source_batch, B, "amax")
    if targets is not None:
        r = torch.maximum(r, _segment_reduce(_point_norms(targets, norm), target_batch, B, "amax"))
    return finish(r)


def scale_points_by_norm(sources, targets=None, source_batch=None, target_batch=None, /, batch=None, factor=1,
                         norm="euclidean"):
    """Scale every point set so that its radius becomes ``factor`` (utils.py:85-99)."""
    source_batch, target_batch = _resolve(source_batch, target_batch, batch)
    radius = compute_points_radius(sources, targets, source_batch, target_batch, norm=norm)
    scale = factor / radius
    sources = sources * (scale if source_batch is None else scale[source_batch, None])
    if targets is not None:
        targets = targets * (scale if target_batch is None else scale[target_batch, None])
    return sources, targets
