"""hipGraph capture of callables with a per-signature cache ("dynamic shape" by recapture).

Public surface and semantics mirror /root/reference/src/sfast/cuda/graphs.py:
  make_dynamic_graphed_callable (:16-51)   cache key = (training, hash_arg(args), hash_arg(kwargs)),
                                           double-checked locking, `_cached` / `__self__` attributes
  simple_make_graphed_callable  (:54-64)
  make_graphed_callable         (:67-185)  3 warm-up runs on a side stream, static input buffers,
                                           capture on the per-device stream + pool under its lock,
                                           replay wrapper = copy-in -> replay -> clone outputs
  GraphExecutionEnv / get_per_device_graph_execution_env (:188-222)
  hash_arg (:225-241)                      tensors hash to (device, dtype, shape [, value of CPU scalars])

`torch.cuda.CUDAGraph` is hipGraph on ROCm; the kernels of libsfast_hip.so are launched through
ctypes on the capturing stream and are recorded like any other HIP launch (they never allocate,
free or synchronise). Unlike the reference no shadow tensors are needed: static inputs are plain
allocations owned by the graphed callable.
"""
import dataclasses
import functools
import logging
import threading

import torch

from ..utils.copy import tree_copy, tree_copy_

logger = logging.getLogger()

_envs = {}
_envs_lock = threading.Lock()


class GraphExecutionEnv:
    """One capture/replay stream + graph memory pool + lock per device."""

    def __init__(self, *, mempool, device=None, stream=None, lock=None):
        self.mempool = mempool
        if isinstance(device, torch.device):
            assert device.type == "cuda"
            device = device.index
        self.device = torch.cuda.current_device() if device is None else device
        self.stream = torch.cuda.current_stream(self.device) if stream is None else stream
        self.lock = threading.Lock() if lock is None else lock
        # keep the pool alive: an (empty) graph captured into it holds a use-count
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            with torch.cuda.graph(graph, pool=self.mempool, stream=self.stream):
                pass
        self.graph = graph


def get_per_device_graph_execution_env(device=None):
    if isinstance(device, torch.device):
        assert device.type == "cuda"
        device = device.index
    if device is None:
        device = torch.cuda.current_device()
    with _envs_lock:
        env = _envs.get(device)
        if env is None:
            with torch.cuda.device(device):
                mempool = torch.cuda.graphs.graph_pool_handle()
                stream = torch.cuda.Stream()
            env = GraphExecutionEnv(mempool=mempool, device=device, stream=stream, lock=threading.Lock())
            _envs[device] = env
        return env


def hash_arg(arg):
    if isinstance(arg, torch.Tensor):
        dev = arg.device
        val = arg.item() if dev.type == "cpu" and arg.numel() == 1 else None
        return (dev.type, dev.index, arg.dtype, tuple(arg.shape), val)
    if isinstance(arg, (str, int, float, bytes)):
        return arg
    if isinstance(arg, (tuple, list)):
        return tuple(hash_arg(a) for a in arg)
    if isinstance(arg, dict):
        return tuple(sorted(((hash_arg(k), hash_arg(v)) for k, v in arg.items()), key=lambda kv: repr(kv[0])))
    return type(arg)


def get_cuda_device_from_tensors(x):
    if isinstance(x, torch.Tensor):
        return x.device.index if x.device.type == "cuda" else None
    if isinstance(x, (list, tuple)):
        for y in x:
            d = get_cuda_device_from_tensors(y)
            if d is not None:
                return d
        return None
    if isinstance(x, dict):
        return get_cuda_device_from_tensors(list(x.values()))
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        return get_cuda_device_from_tensors([getattr(x, f.name) for f in dataclasses.fields(x)])
    return None


def _owner_module(func):
    if isinstance(func, torch.nn.Module):
        return func
    owner = getattr(func, "__self__", None)
    return owner if isinstance(owner, torch.nn.Module) else None


def make_dynamic_graphed_callable(func):
    lock = threading.Lock()
    cached = {}
    wrapped = func.forward if isinstance(func, torch.nn.Module) else func

    @functools.wraps(wrapped)
    def dynamic_graphed_callable(*args, **kwargs):
        owner = _owner_module(func)
        training = bool(getattr(owner, "training", False)) if owner is not None else False
        key = (training, hash_arg(args), hash_arg(kwargs))
        fn = cached.get(key)
        if fn is None:
            with lock:
                fn = cached.get(key)
                if fn is None:
                    logger.info("Dynamically graphing %s", getattr(func, "__name__", func.__class__.__name__))
                    fn = simple_make_graphed_callable(func, args, kwargs)
                    cached[key] = fn
        return fn(*args, **kwargs)

    owner = _owner_module(func)
    if owner is not None:
        dynamic_graphed_callable.__self__ = owner
    dynamic_graphed_callable._cached = cached
    return dynamic_graphed_callable


def simple_make_graphed_callable(func, example_inputs=None, example_kwarg_inputs=None):
    device = get_cuda_device_from_tensors((example_inputs, example_kwarg_inputs))
    if device is None:
        raise ValueError("simple_make_graphed_callable: no CUDA/ROCm tensor among the example inputs")
    env = get_per_device_graph_execution_env(device)
    return make_graphed_callable(func, example_inputs, example_kwarg_inputs, execution_env=env)


def make_graphed_callable(func, example_inputs=None, example_kwarg_inputs=None, *, execution_env, warmups=3):
    env = execution_env
    example_inputs = tuple() if example_inputs is None else tuple(example_inputs)
    example_kwarg_inputs = {} if example_kwarg_inputs is None else dict(example_kwarg_inputs)
    owner = _owner_module(func)
    training = bool(getattr(owner, "training", False)) if owner is not None else False

    # warm-up off the capture stream: lazy library initialisation must not end up in the capture
    torch.cuda.synchronize(env.device)
    with torch.cuda.device(env.device), torch.cuda.stream(torch.cuda.Stream(device=env.device)):
        for _ in range(warmups):
            func(*tree_copy(example_inputs, detach=True), **tree_copy(example_kwarg_inputs, detach=True))
    torch.cuda.synchronize(env.device)

    static_inputs = tree_copy(example_inputs, detach=True)
    static_kwarg_inputs = tree_copy(example_kwarg_inputs, detach=True)
    graph = torch.cuda.CUDAGraph()
    try:
        with env.lock:
            with torch.cuda.device(env.device), torch.cuda.stream(env.stream):
                with torch.cuda.graph(graph, pool=env.mempool, stream=env.stream):
                    static_outputs = func(*static_inputs, **static_kwarg_inputs)
    except Exception:
        logger.error("Failed to capture hipGraph, please try without it")
        raise

    deps = [func]
    if owner is not None:
        deps.extend(p.data for p in owner.parameters())

    def graphed(*inputs, **kwarg_inputs):
        with env.lock:
            tree_copy_(static_inputs, inputs)
            tree_copy_(static_kwarg_inputs, kwarg_inputs)
            graph.replay()
            return tree_copy(static_outputs)

    graphed._graph = graph
    graphed._deps = deps
    graphed._training = training
    return graphed


# ---- per-module automatic graphing (reference cuda/graphs.py:296-352 + hooks/module_jit_hook.py:9-85) ---------------------------
class AutoGraphCraphCompiler:
    """Compiler object of the reference's module hook (its spelling): decides from the call's inputs / outputs whether a module
    call can be replayed from a hipGraph (only tensors / scalars / strings in nested containers) and captures it."""

    def __init__(self, **kwargs):
        # the reference forwards **kwargs to simple_make_graphed_callable, which accepts none (a latent TypeError at the first capture,
        # cuda/graphs.py:296-340 there): reject them where the mistake is made instead
        if kwargs:
            raise TypeError(f"AutoGraphCraphCompiler: unexpected keyword arguments {sorted(kwargs)} "
                            "(simple_make_graphed_callable takes none)")
        self.kwargs = {}
        self._is_compiling = threading.local()

    def is_compiling(self):
        return getattr(self._is_compiling, "value", False)

    def get_inputs_key(self, func, inputs, kwargs):
        from ..utils.copy import can_be_perfectly_copied
        if not can_be_perfectly_copied((inputs, kwargs)):
            return None
        return (hash_arg(inputs), hash_arg(kwargs))

    def get_outputs_key(self, func, outputs):
        from ..utils.copy import can_be_perfectly_copied
        if not can_be_perfectly_copied(outputs):
            return None
        return hash_arg(outputs)

    def compile(self, func, inputs, kwargs):
        self._is_compiling.value = True
        try:
            graphed = simple_make_graphed_callable(func, inputs, kwargs, **self.kwargs)
            wrapped = func.forward if isinstance(func, torch.nn.Module) else func

            @functools.wraps(wrapped)
            def functionalized(*args, **kw):
                return graphed(*args, **kw)

            owner = _owner_module(func)
            if owner is not None:
                functionalized.__self__ = owner
            return functionalized
        finally:
            self._is_compiling.value = False


class _LazyCompiledForward:
    """Per-module call cache: the first call of an input signature runs eagerly (it reveals whether the outputs can be graphed and
    whether the call mutates its inputs), later calls replay the captured graph; signatures that cannot be graphed stay eager."""
    _CANNOT, _READY = object(), object()

    def __init__(self, module, compiler):
        self.module, self.compiler = module, compiler
        self.forward = module.forward
        self.cache = {}
        self.lock = threading.Lock()
        self.__self__ = module
        self.__name__ = "forward"

    def __call__(self, *args, **kwargs):
        c = self.compiler
        if c.is_compiling():
            return self.forward(*args, **kwargs)
        key = c.get_inputs_key(self.forward, args, kwargs)
        if key is None:
            return self.forward(*args, **kwargs)
        hit = self.cache.get(key)
        if hit is not None and hit is not self._CANNOT and hit is not self._READY:
            return hit(*args, **kwargs)
        with self.lock:
            hit = self.cache.get(key)
            if hit is self._CANNOT:
                return self.forward(*args, **kwargs)
            if hit is self._READY:
                try:
                    hit = self.cache[key] = c.compile(self.forward, args, kwargs)
                except Exception:
                    self.cache[key] = self._CANNOT
                    logger.exception("Failed to graph %s", type(self.module).__name__)
                    raise
                return hit(*args, **kwargs)
            if hit is not None:
                return hit(*args, **kwargs)
            out = self.forward(*args, **kwargs)
            if c.get_outputs_key(self.forward, out) is None:
                self.cache[key] = self._CANNOT
            elif c.get_inputs_key(self.forward, args, kwargs) == key:  # inputs not mutated by the call
                self.cache[key] = c.compile(self.forward, args, kwargs)
            else:
                self.cache[key] = self._READY
            return out


def apply_auto_graph_compiler_to_all_modules(m, filter_func=None, recursive=True, **kwargs):
    """Wrap the forward of `m` (recursive=False) or of every module the filter accepts, walking like the reference's patch_module:
    `filter_func(stack)` sees the [(name, module), ...] path from the root; an accepted CHILD is wrapped and not descended into, an
    accepted root is wrapped and its children are still walked (utils/patch.py:1-19 there)."""
    compiler = AutoGraphCraphCompiler(**kwargs)

    def wrap(mod):
        if not isinstance(mod.forward, _LazyCompiledForward):
            mod.forward = _LazyCompiledForward(mod, compiler)
        return mod

    if not recursive:
        return wrap(m)
    filt = filter_func if filter_func is not None else (lambda stack: True)

    def walk(mod, stack):
        for name, child in mod.named_children():
            stack.append((name, child))
            if filt(stack):
                wrap(child)
            else:
                walk(child, stack)
            stack.pop()

    root = [(None, m)]
    if filt(root):
        wrap(m)
    walk(m, root)
    return m
