"""hipGraph capture of a training step, and the device-resident mask seeds that make it legal.

A launch-bound step (BASELINE config 4 as specified: 16 graphs per GPU, 16 layers, ~35 launches per
layer) spends most of its wall time between kernels.  Captured once into a hipGraph the whole forward +
backward is ONE host call per step.  Kernel arguments are frozen at capture time, so the only per-step
host input of the path — the random attention mask's seed (egt_layers.py:97-103 draws a fresh sample on
every call) — moves into device memory: `egt_block_desc.flags & EGT_BF_SEED_DEVICE`, the kernels XOR the
uint64 at `desc.seed_device` into their seed when they RUN.  `DeviceSeeds.advance()` is itself a captured
launch, so every replay draws a fresh sample, and the sequence of samples is bit-identical to the eager
path's host-side `EGT.next_seed()` sequence (tests/test_graph_gpu.py).
"""
from __future__ import annotations

import torch

_GOLD = 0x9E3779B97F4A7C15      # EGT.next_seed(): seed * _GOLD + calls * _STEP   (mod 2^64)
_STEP = 0xD1B54A32D192ED03
_M64 = 0xFFFFFFFFFFFFFFFF


def _signed(x: int) -> int:
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


class DeviceSeeds:
    """One uint64 word in HBM per EGT module; word i holds what `modules[i].next_seed()` would return for the
    module's current call count.  `advance()` (one launch, capturable) moves every word to the next call."""

    def __init__(self, modules, device):
        self.modules = list(modules)
        if not self.modules:
            raise ValueError("DeviceSeeds needs at least one EGT module")
        vals = [_signed(m.seed * _GOLD + m._calls * _STEP) for m in self.modules]
        self.words = torch.tensor(vals, dtype=torch.int64, device=device)
        self._inc = _signed(_STEP)
        self.steps = 0
        for i, m in enumerate(self.modules):
            m.seed_device = (self, i)

    @classmethod
    def attach(cls, model: torch.nn.Module, device=None):
        """Every EGT module under `model` (in module order) draws its mask seed from device memory from now on."""
        from .layers import EGT
        mods = [m for m in model.modules() if isinstance(m, EGT)]
        if device is None:
            device = next(model.parameters()).device
        return cls(mods, device)

    def ptr(self, i: int) -> int:
        return self.words.data_ptr() + 8 * i

    def advance(self):
        if self.words.is_cuda:         # the library's own one-launch advance (what a C host captures in front of the forward)
            from . import _lib as L
            L.check(L.load().egt_seed_advance(L.ptr(self.words), len(self.modules), _STEP, L.current_stream()))
        else:
            self.words.add_(self._inc)     # int64 wrap-around == arithmetic mod 2^64
        self.steps += 1                # (host-side bookkeeping only; replays advance the device words, not this)

    def values(self):
        """The current words as unsigned Python ints (synchronises; for tests)."""
        return [int(v) & _M64 for v in self.words.tolist()]

    def state_dict(self):
        """the device words (the position of every module's mask stream), for checkpoints: a resumed run continues the
        sequence instead of repeating it"""
        return dict(words=self.values(), steps=self.steps)

    def load_state_dict(self, sd):
        vals = [_signed(int(v)) for v in sd["words"]]
        if len(vals) != len(self.modules):
            raise ValueError("mask-seed state does not match the model's EGT modules")
        self.words.copy_(torch.tensor(vals, dtype=torch.int64))
        self.steps = int(sd.get("steps", 0))

    def detach(self):
        """Back to host-side seeds; the modules' call counts continue after the device words' position."""
        vals = self.values()
        inv = pow(_STEP, -1, 1 << 64)
        for m, v in zip(self.modules, vals):
            m._calls = ((v - m.seed * _GOLD) * inv) & _M64
            m.seed_device = None


class GraphedStep:
    """`fn()` — forward + backward of one step on static input tensors — captured into a hipGraph.

    fn must use the fused path only (a composed-path call with an active random mask raises: its seed is a
    host argument), must not synchronise, and must leave its results (outputs, .grad / the flat gradient
    buffer) in tensors it returns or that the caller holds: replays rewrite the same addresses.
    `seeds.advance()` is captured in front of fn, so replay k draws the sample eager call k would.
    No autograd graph of an earlier eager call on the same leaf tensors may still be alive (e.g. an output that was
    kept without .detach()): its AccumulateGrad nodes stay bound to the stream they were created on, the capture
    would then span a non-capturing stream, and hipStreamEndCapture does not survive that."""

    def __init__(self, fn, seeds: DeviceSeeds | None = None, warmup: int = 2):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU (hipGraph capture)")
        self.fn, self.seeds = fn, seeds
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # lazy one-time work (kernel attributes, workspace sizing) happens un-captured
            for _ in range(warmup):
                if seeds is not None:
                    seeds.advance()
                r = fn()
                del r                      # (a kept output would keep the warm-up's autograd graph alive into the capture)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):    # the warm-up's stream: autograd's per-leaf stream bookkeeping matches
            if seeds is not None:
                seeds.advance()
            self.result = fn()
        self.replays = 0

    def replay(self):
        self.graph.replay()
        self.replays += 1
        return self.result
