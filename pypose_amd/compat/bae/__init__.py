"""Stand-in for the ``bae`` plugin namespace PyPose's sparse LM loads (see pypose_amd/compat/__init__.py).  NOT the
sair-lab/bae package: the five entry points are re-implemented on torch autograd + torch.sparse, so that the reference's
own ``LM(sparse=True)`` code path (pypose/optim/optimizer.py:629-643, 663-664) runs wherever torch does -- on the MI355X
its Lie ops take the HIP kernels once ``pypose_amd.activate.activate(pypose)`` has rebound them."""
__version__ = "0.2.1"
