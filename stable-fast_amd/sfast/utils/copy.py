"""Structure-preserving copies for the static buffers of graphed callables.

Behaviour follows /root/reference/src/sfast/utils/copy.py:6-59: `tree_copy_` copies replay inputs INTO the captured
buffers, `tree_copy` clones outputs out of them, `can_be_perfectly_copied` says whether a value survives that round trip.
The reference's `shadow_copy` aliases pool memory through a C++ helper (csrc/misc.cpp:25-30); the static buffers here are
ordinary allocations owned by the graphed callable, so there is nothing to alias.

One traversal (`_children`) defines what a container is -- sequences, mappings, dataclass instances -- and the three entry
points are folds over it.
"""
import dataclasses

import torch

_LEAF = object()


def _children(node):
    """(keys, values, rebuild) of a container node, or _LEAF."""
    if isinstance(node, torch.Tensor):
        return _LEAF
    if isinstance(node, tuple) and hasattr(node, "_fields"):  # namedtuple: positional constructor
        return range(len(node)), list(node), lambda vals, t=type(node): t(*vals)
    if isinstance(node, (list, tuple)):
        return range(len(node)), list(node), type(node)
    # dataclass before dict: diffusers' BaseOutput is both, and only its field constructor rebuilds it
    if dataclasses.is_dataclass(node) and not isinstance(node, type):
        names = [f.name for f in dataclasses.fields(node)]
        return names, [getattr(node, n) for n in names], lambda vals, t=type(node), ns=names: t(**dict(zip(ns, vals)))
    if isinstance(node, dict):
        keys = list(node)
        return keys, [node[k] for k in keys], lambda vals, t=type(node), ks=keys: t(zip(ks, vals))
    return _LEAF


def tree_copy_(dest, src):
    """dest[...] <- src[...] leaf by leaf; both trees must have the same shape."""
    kids = _children(dest)
    if kids is _LEAF:
        if isinstance(dest, torch.Tensor):
            dest.copy_(src)
        elif type(dest) is not type(src):
            raise ValueError(f"tree_copy_: leaf type mismatch ({type(dest).__name__} vs {type(src).__name__})")
        return
    other = _children(src)
    if other is _LEAF or list(kids[0]) != list(other[0]):
        raise ValueError("tree_copy_: structure mismatch")
    for d, s in zip(kids[1], other[1]):
        tree_copy_(d, s)


def tree_copy(src, detach=False):
    """Deep copy with cloned tensors (detached when asked); non-tensor leaves are shared."""
    kids = _children(src)
    if kids is _LEAF:
        if isinstance(src, torch.Tensor):
            return src.detach().clone() if detach else src.clone()
        return src
    _, values, rebuild = kids
    return rebuild([tree_copy(v, detach=detach) for v in values])


def can_be_perfectly_copied(obj):
    kids = _children(obj)
    if kids is _LEAF:
        return obj is None or isinstance(obj, (torch.Tensor, float, int, str, bytes))
    return all(can_be_perfectly_copied(v) for v in kids[1])
