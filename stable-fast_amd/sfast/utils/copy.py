"""Tree copy helpers for static graph buffers.

Semantics follow /root/reference/src/sfast/utils/copy.py:6-59 (`tree_copy_` copy-in of replay
inputs, `tree_copy` clone of outputs). The reference's `shadow_copy` aliases pool memory through a
C++ helper (csrc/misc.cpp:25-30); here static buffers are ordinary allocations that the graphed
callable owns, so aliasing is not needed.
"""
import dataclasses

import torch


def tree_copy_(dest, src):
    if isinstance(dest, torch.Tensor):
        dest.copy_(src)
    elif isinstance(dest, (list, tuple)):
        if len(dest) != len(src):
            raise ValueError("tree_copy_: structure mismatch")
        for d, s in zip(dest, src):
            tree_copy_(d, s)
    elif isinstance(dest, dict):
        if len(dest) != len(src):
            raise ValueError("tree_copy_: structure mismatch")
        for k in dest:
            tree_copy_(dest[k], src[k])
    elif dataclasses.is_dataclass(dest) and not isinstance(dest, type):
        for f in dataclasses.fields(dest):
            tree_copy_(getattr(dest, f.name), getattr(src, f.name))
    else:
        if type(dest) is not type(src):
            raise ValueError("tree_copy_: leaf type mismatch")


def tree_copy(src, detach=False):
    if isinstance(src, torch.Tensor):
        return src.detach().clone() if detach else src.clone()
    if isinstance(src, (list, tuple)):
        vals = [tree_copy(x, detach=detach) for x in src]
        if hasattr(src, "_fields"):  # namedtuple
            return type(src)(*vals)
        return type(src)(vals)
    if isinstance(src, dict):
        return type(src)((k, tree_copy(v, detach=detach)) for k, v in src.items())
    if dataclasses.is_dataclass(src) and not isinstance(src, type):
        return type(src)(**{f.name: tree_copy(getattr(src, f.name), detach=detach) for f in dataclasses.fields(src)})
    return src


def can_be_perfectly_copied(obj):
    if obj is None or isinstance(obj, (torch.Tensor, float, int, str, bytes)):
        return True
    if isinstance(obj, (list, tuple)):
        return all(can_be_perfectly_copied(x) for x in obj)
    if isinstance(obj, dict):
        return all(can_be_perfectly_copied(v) for v in obj.values())
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return all(can_be_perfectly_copied(getattr(obj, f.name)) for f in dataclasses.fields(obj))
    return False
