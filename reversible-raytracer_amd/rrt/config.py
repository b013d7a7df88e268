"""Device / dtype defaults (the reference's ``theano.config.floatX``,
README.md:11: ``floatX=float32``)."""
import torch

floatX = torch.float32
_device = [None]


def device():
    """Where Scene parameters live: the current ROCm GPU when there is one.
    On a CPU-only host the transform / class plumbing still works (it is plain
    torch), but ``Scene.build()`` raises: the render path is HIP only."""
    if _device[0] is None:
        _device[0] = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device("cpu")
    return _device[0]


def set_device(dev):
    _device[0] = torch.device(dev)


_const_cache = {}


def _const_key(x):
    """Hashable key for small Python constants (numbers, flat tuples/lists of numbers)."""
    if isinstance(x, (int, float)):
        return (float(x),)
    if isinstance(x, (tuple, list)) and len(x) <= 16 and all(isinstance(v, (int, float)) for v in x):
        return tuple(float(v) for v in x)
    return None


def as_tensor_variable(x):
    """``T.as_tensor_variable`` / ``theano.shared`` stand-in: a float32 tensor
    on the default device; tensors are passed through (keeping autograd).

    Small Python constants (``translate((0, 0, 3))``, ``rotate(50, [0., 1., 0.])``)
    are uploaded once and cached per device: the reference bakes them into its
    Theano graph as constants, and a cached device tensor is what lets a whole
    gradient-descent step be captured in a HIP graph (no host-to-device copy
    inside the captured region, see ``GDOptimizer.optimize(graph=True)``)."""
    if torch.is_tensor(x):
        return x if (x.dtype == floatX and x.device == device()) else x.to(device=device(), dtype=floatX)
    key = _const_key(x)
    if key is not None:
        dev = device()
        ck = (str(dev), isinstance(x, (int, float)), key)
        t = _const_cache.get(ck)
        if t is None:
            t = torch.as_tensor(x, dtype=floatX, device=dev)
            _const_cache[ck] = t
        return t
    return torch.as_tensor(x, dtype=floatX, device=device())


def shared(value, requires_grad=True):
    """``theano.shared(np.asarray(...))``: a leaf tensor the optimiser updates."""
    t = torch.as_tensor(value, dtype=floatX).to(device()).clone()
    t.requires_grad_(requires_grad)
    return t
