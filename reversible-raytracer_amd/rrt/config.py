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


def as_tensor_variable(x):
    """``T.as_tensor_variable`` / ``theano.shared`` stand-in: a float32 tensor
    on the default device; tensors are passed through (keeping autograd)."""
    if torch.is_tensor(x):
        return x if (x.dtype == floatX and x.device == device()) else x.to(device=device(), dtype=floatX)
    return torch.as_tensor(x, dtype=floatX, device=device())


def shared(value, requires_grad=True):
    """``theano.shared(np.asarray(...))``: a leaf tensor the optimiser updates."""
    t = torch.as_tensor(value, dtype=floatX).to(device()).clone()
    t.requires_grad_(requires_grad)
    return t
