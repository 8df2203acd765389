"""Tensors derived from parameters, computed once per parameter value.

A training iteration evaluates each network several times between two optimizer steps
(the generator in Gmain, Dmain and Greg; the discriminator in Gmain and twice in Dmain),
and most generator weights of a few-shot run never change at all.  Everything a layer
derives from its weights alone - the effective weight with its domain offset, the cast to
the activation dtype, the per-tap packing the convolution kernels read, the squared-weight
sums of the demodulation - is therefore kept while the source tensors are unchanged,
instead of being rebuilt by a dozen small kernels per layer and pass.

`derived(owner, key, sources, fn)` caches `fn()` on the module `owner` under `key`; the entry is
valid while every tensor of `sources` keeps its storage and version counter (in-place writes -
optimizer steps, `load_state_dict`, EMA updates - advance the counter).  Values that need a
gradient graph (a source requires grad while grad mode is on) are never cached.  Tensors
handed out by `derived` are constants: `memo(t, key, fn)` attaches further derived values
(flips, transposes, packings) to them, except while `t` requires grad in grad mode (then `fn()`
is evaluated in the open, as part of the caller's graph).
"""
import torch

enabled = True


def _bypass(sources, values_only=False):
    if not enabled:
        return True
    if not values_only and torch.is_grad_enabled() and any(s.requires_grad for s in sources):
        return True
    # Inside a stream capture every tensor lives in the graph's pool and is rewritten by each replay;
    # a value cached outside would go stale when the parameters change between replays.
    return sources[0].is_cuda and torch.cuda.is_current_stream_capturing()


def constant(t):
    """Mark `t` as a tensor that is never written again, so `memo` may attach derived values to it."""
    if isinstance(t, torch.Tensor) and not hasattr(t, '_sg_memo'):
        t._sg_memo = {}
    return t


def is_constant(t):
    return getattr(t, '_sg_memo', None) is not None


def memo(t, key, fn):
    """`fn()` - a pure function of the tensor `t` - computed once if `t` is a constant."""
    cache = getattr(t, '_sg_memo', None)
    if cache is None or (t.is_cuda and torch.cuda.is_current_stream_capturing()):
        return fn()
    if torch.is_grad_enabled() and t.requires_grad:   # the caller wants a graph through fn: a cached value has none
        return fn()
    hit = cache.get(key)
    if hit is None:
        with torch.no_grad():
            hit = fn()
        cache[key] = constant(hit)
    return hit


def share(alias, base):
    """Let `alias` (a view of the constant `base` made for one autograd graph) use base's memoised values."""
    cache = getattr(base, '_sg_memo', None)
    if cache is not None:
        alias._sg_memo = cache
    return alias


def derived(owner, key, sources, fn, values_only=False):
    """`values_only`: the caller attaches the gradient itself (an autograd function whose forward value this is),
    so the entry is served even while a source requires grad."""
    sources = [s for s in sources if s is not None]
    if _bypass(sources, values_only):
        if values_only:
            with torch.no_grad():
                return fn()
        return fn()
    stamp = tuple((s.data_ptr(), s._version) for s in sources)
    store = owner.__dict__.setdefault('_sg_derived', {})
    hit = store.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    with torch.no_grad():
        value = fn()
    storages = {s.untyped_storage().data_ptr() for s in sources}
    if value.untyped_storage().data_ptr() in storages:   # an alias of a parameter would change under the cache
        value = value.clone()
    store[key] = (stamp, constant(value))
    return value
