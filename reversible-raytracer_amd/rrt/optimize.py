"""Optimisers -- reference ``optimize.py`` (GDOptimizer :11-29,
MGDAutoOptimizer :63-84) restated for eager autograd.

The one unavoidable signature change: Theano's ``loss`` is a symbolic graph
that ``theano.function`` re-evaluates on every call; in eager torch the loss
must be a CALLABLE returning a scalar tensor (typically a closure that calls
``scene.build()`` again).  Everything else follows the reference, including
the call pattern of its drivers (``opt.optimize(vars, loss, 0.0008, 0.1)`` then
``train()``: optimize_brightness.py:50-57, match_mirror.py:48-52), which the
reference's own 3-argument signature (optimize.py:19) cannot run.
"""
import torch


class GDOptimizer(object):
    """Gradient Descent optimizer"""

    def __init__(self):
        pass

    def optimize(self, tVars, loss, lr=None, momentum=0, graph=False):
        """Returns ``train(lr=None) -> loss value`` performing one step
        ``v <- v - lr * dloss/dv`` for every v in tVars (optimize.py:23-29; the
        reference ignores ``momentum`` too).  ``lr`` may be bound here (the
        drivers' call pattern) or passed per call (``theano.function([lr], ...)``).
        The returned loss is evaluated BEFORE the update, like Theano's.

        ``graph=True`` (needs a bound ``lr``): the whole step -- the tiny
        transform ops that rebuild the scene, the render kernels, the loss, the
        backward kernels and the in-place update -- is captured once into a HIP
        graph and replayed.  This is the counterpart of ``theano.function``
        compiling the step: the eager step is launch-bound (hundreds of tiny
        torch kernels around ~0.1 ms of render work)."""
        bound_lr = lr
        if graph:
            return self._graphed(tVars, loss, bound_lr)

        def train(lr=None):
            step = bound_lr if lr is None else lr
            if step is None:
                raise TypeError("train() needs a learning rate (none was bound in optimize())")
            value = loss()
            grads = torch.autograd.grad(value, tVars, allow_unused=True)
            with torch.no_grad():
                for var, gvar in zip(tVars, grads):
                    if gvar is not None:
                        var -= step * gvar
            return float(value.detach())

        return train

    @staticmethod
    def _graphed(tVars, loss, lr):
        if lr is None:
            raise TypeError("graph=True needs the learning rate bound in optimize()")
        dev = tVars[0].device
        if dev.type != "cuda":
            raise RuntimeError("graph=True needs the variables on a GPU")

        def step():
            value = loss()
            grads = torch.autograd.grad(value, tVars, allow_unused=True)
            with torch.no_grad():
                for var, gvar in zip(tVars, grads):
                    if gvar is not None:
                        var -= lr * gvar
            return value.detach()

        # warm-up on a side stream (fills the constant caches, lets the allocator settle), undone afterwards
        saved = [v.detach().clone() for v in tVars]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for v, s0 in zip(tVars, saved):
                v.copy_(s0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        with torch.no_grad():                      # the capture does not execute: nothing to undo, but be explicit
            for v, s0 in zip(tVars, saved):
                v.copy_(s0)

        def train(lr_call=None):
            if lr_call is not None and lr_call != lr:
                raise ValueError("the learning rate of a graphed step is fixed at capture time")
            g.replay()
            return float(out)

        train.graph = g
        return train


class MGDAutoOptimizer(object):
    """SGD over an auto-encoder's parameters (optimize.py:63-84; orbit fork
    orbit_experiments/optimize.py:68-97 takes a stereo pair and scales the
    learning rate of 1-D parameters by 0.1)."""

    def __init__(self, ae):
        self.ae = ae

    def optimize(self, train_data, lam=None):
        ae = self.ae

        def opt(lr):
            X = train_data[0]
            cost = ae.cost(*X) if isinstance(X, (tuple, list)) else ae.cost(X)
            grads = torch.autograd.grad(cost, ae.params, allow_unused=True)
            with torch.no_grad():
                for var, gvar in zip(ae.params, grads):
                    if gvar is None:
                        continue
                    scale = 0.1 if (lam is not None and var.dim() == 1) else 1.0
                    var -= scale * lr * gvar
            return float(cost.detach())

        return opt
