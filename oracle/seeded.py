"""oracle/seeded.py -- TEST INFRASTRUCTURE.

Name-seeded parameter recipe (SURVEY.md section 8c "Weights for parity without
shipping checkpoints"): every tensor of a state dict is drawn from a NumPy
PCG64 stream seeded by the CRC32 of its key, with a per-kind scale, so the
reference model (in the build container) and the model under test (anywhere)
can be given bit-identical weights without committing a 200 MB checkpoint.
Default initialisation is avoided on purpose: the reference zero-initialises the
offset / logit Linears (OT:1631-1642), which would hide sampling bugs.
"""
import zlib

import numpy as np
import torch


def _canonical(key):
    """The reference's joint-decoder attention assigns ONE tensor as the `.data` of the
    pre_/''/next_ sampling_offsets biases (MO:1366-1368, 1707-1711), so they alias the same
    storage and `load_state_dict` leaves all frames with whichever was copied last.  Real
    checkpoints therefore hold equal values; the recipe gives every frame of a module the
    same sampling_offsets bias (weights still differ per frame)."""
    if key.endswith('sampling_offsets.bias'):
        head, _, leaf = key[:-len('.bias')].rpartition('.')
        while leaf.startswith('pre_') or leaf.startswith('next_'):
            leaf = leaf[4:] if leaf.startswith('pre_') else leaf[5:]
        return (head + '.' if head else '') + leaf + '.bias'
    return key


def _rng(key, salt=0):
    return np.random.default_rng([zlib.crc32(_canonical(key).encode()), salt])


def seeded_tensor(key, shape, dtype=torch.float32, salt=0):
    """Deterministic tensor for parameter `key` of shape `shape`."""
    shape = tuple(int(s) for s in shape)
    r = _rng(key, salt)
    n = r.standard_normal(shape, dtype=np.float64) if shape else r.standard_normal()
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked' or dtype in (torch.int64, torch.long):
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == 'running_var':
        v = 1.0 + 0.1 * np.abs(n)
    elif leaf == 'running_mean':
        v = 0.1 * n
    elif 'sampling_offsets' in key:
        v = 0.05 * n if leaf == 'weight' else 1.0 * n
    elif 'attention_weights' in key:
        v = 0.05 * n if leaf == 'weight' else 0.5 * n
    elif 'cls_branches' in key:
        v = 0.5 * n if leaf == 'weight' else 0.1 * n - 2.0
    elif 'embedding' in key or 'level_embeds' in key:
        v = n
    elif leaf == 'mask':  # RealNVP buffer (unused at inference)
        v = np.resize(np.array([[0, 1], [1, 0]], dtype=np.float64), shape)
    elif len(shape) == 1 and leaf == 'weight':
        # norm scales; the last BN of a residual block is damped so that 16
        # stacked bottlenecks keep activations O(1)
        v = (0.3 if key.endswith('bn3.weight') else 1.0) * (1.0 + 0.1 * n)
    elif leaf == 'bias' or len(shape) == 1:
        v = 0.05 * n
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        gain = 0.5 if ('refine_kpt_branches' in key or 'kpt_branches' in key) and shape[0] <= 64 \
            else 1.0
        v = gain * n * np.sqrt(2.0 / fan_in)
    return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape)).to(dtype)


def seeded_state_dict(key_shapes, salt=0, like=None):
    """key_shapes: {key: shape}; `like` (optional) gives dtypes per key."""
    out = {}
    for k, shp in key_shapes.items():
        dt = like[k].dtype if like is not None else (
            torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
        out[k] = seeded_tensor(k, shp, dt, salt)
    return out


def seeded_array(name, shape, scale=1.0, salt=0):
    """Deterministic float32 input array (for test inputs, not parameters)."""
    r = _rng('input:' + name, salt)
    return (scale * r.standard_normal(tuple(shape))).astype(np.float32)
