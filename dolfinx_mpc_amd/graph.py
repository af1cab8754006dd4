"""Steady-state steps as HIP graphs.

A time loop calls ``assemble_matrix`` / ``assemble_vector`` / ``apply_lifting`` with the same forms on the same mesh
thousands of times (python/demos, python/benchmarks/bench_periodic.py in a Newton or time loop).  After the first call
every plan, pattern and device mirror is cached, and what a call costs on the host is argument blocks, dispatch look-ups
and ~10 kernel launches: 172 us per matrix + vector step -- invisible next to 3.5 ms of kernels on one GPU, but within a
factor of three of the 0.45 ms a rank has at 8 GPUs.  ``CapturedStep`` records the launches of one step ONCE (HIP stream
capture through ``torch.cuda.CUDAGraph``: the library's kernels, the side streams they fork onto and the joins) and
replays them with one ``hipGraphLaunch``.

What a replay re-reads and what it does not -- the contract differs from the plain calls:
  * the kernels run again on the SAME device arrays: coordinates (after ``mesh.geometry.x = ...`` the mirror is the same
    tensor: call ``refresh()``), packed coefficients, constants and Dirichlet values are read from their device copies;
  * the host-side value checks of the plain calls (coefficients written through kept views, changed constants / boundary
    values) do NOT run in a replay.  ``refresh()`` runs one plain step (which uploads whatever changed, into the same
    device tensors when shapes are unchanged) -- call it after changing values; structure changes (new forms, another
    constraint) need a new capture.
One process, one GPU (the interface exchange of partitioned meshes goes through torch.distributed and is not captured)."""

from __future__ import annotations

from typing import Callable


class CapturedStep:
    """``step = CapturedStep(lambda: (dm.assemble_matrix(a, mpc, bcs=bcs, A=A), dm.assemble_vector(L, mpc, b=b)))``;
    ``step.replay()`` per iteration, ``step.refresh()`` after changing coefficient / constant / boundary values."""

    def __init__(self, fn: Callable[[], object], warmup: int = 2):
        import torch

        from . import _native
        from .la import wait_assembly

        _native.require_gpu()
        self._fn = fn
        self._wait = wait_assembly
        for _ in range(max(warmup, 1)):  # plans, patterns, device mirrors, allocator growth: outside the capture
            fn()
        wait_assembly()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()
            wait_assembly()  # the side streams forked by the assemblies join the capturing stream

    def replay(self) -> None:
        self.graph.replay()

    def refresh(self) -> None:
        """one plain step: the value checks of the assemblers run and upload what changed"""
        self._fn()
        self._wait()
