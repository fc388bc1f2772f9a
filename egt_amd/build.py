"""Builds the gfx950 C-ABI library in-tree: egt_amd/lib/libegt_amd.so.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libegt_amd.so")
SOURCES = ["egt_capi.hip", "egt_attn.hip", "egt_attn_mfma.hip", "egt_edge.hip", "egt_block.hip", "egt_narrow.hip", "egt_node.hip", "egt_ffn.hip", "egt_masks.hip", "egt_embed.hip", "egt_dp.hip"]
ARCH = "gfx950"
# per-source compiler flags.  egt_ffn.hip: the backward keeps 256 weight-gradient accumulator
# registers per wave; with hipcc's default (AGPR-form MFMA everywhere) the short-lived GEMM
# accumulators compete for the same 256 AccVGPRs and 300+ registers spill; VGPR-form MFMA lets
# the allocator park the long-lived tiles in AccVGPRs instead (0 spills).
EXTRA_FLAGS = {"egt_ffn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # egt_narrow.hip: SLP-packed v_pk_add_f32 cannot take the DPP operand of the dQ reduction
               "egt_narrow.hip": ["-fno-slp-vectorize"]}
if os.environ.get("EGT_BLOCK_FLAGS"):   # experiments: extra hipcc flags for egt_block.hip
    EXTRA_FLAGS["egt_block.hip"] = os.environ["EGT_BLOCK_FLAGS"].split()
if os.environ.get("EGT_NARROW_FLAGS"):  # e.g. -DNRW_ABL=<bits> (timing ablations of the De = 8 kernels)
    EXTRA_FLAGS["egt_narrow.hip"] = EXTRA_FLAGS["egt_narrow.hip"] + os.environ["EGT_NARROW_FLAGS"].split()
if os.environ.get("EGT_ATTN_FLAGS"):    # e.g. -DEGT_ATTN_STAMPS / -DEGT_ATTN_ABL=<bits> (measurement builds of the MFMA inner op)
    EXTRA_FLAGS["egt_attn_mfma.hip"] = os.environ["EGT_ATTN_FLAGS"].split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())   # not the absolute path: the tree may be relocated
            h.update(f.read())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())   # experiment flags change the binary
    return h.hexdigest()


def sources():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(REPO, "include", "egt_amd.h"))
    return srcs, hdrs


def build(force: bool = False, verbose: bool = False) -> str:
    srcs, hdrs = sources()
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, ".stamp")
    dig = _digest(srcs + hdrs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp):
        if open(stamp).read().strip() == dig:
            return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o,
               "-I", CSRC, "-I", os.path.join(REPO, "include"), "-Wno-unused-result"]
        cmd += EXTRA_FLAGS.get(os.path.basename(s), [])
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {s}")
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode(errors="replace"))
        raise RuntimeError("link failed")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


def build_c_host(force: bool = False) -> str:
    """tests/c/graph_replay.cpp -> egt_amd/lib/graph_replay: a C++ host of the C-ABI without torch / Python (stack forward +
    backward captured into a hipGraph; run by tests/test_capi_graph_gpu.py on the GPU box)."""
    lib = build()
    src = os.path.join(REPO, "tests", "c", "graph_replay.cpp")
    exe = os.path.join(LIBDIR, "graph_replay")
    newest = max(os.path.getmtime(x) for x in (src, lib, os.path.join(REPO, "include", "egt_amd.h")))
    if not force and os.path.exists(exe) and os.path.getmtime(exe) >= newest:
        return exe
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", src, "-o", exe, "-I", os.path.join(REPO, "include"),
           "-L", LIBDIR, "-legt_amd", "-Wl,-rpath,$ORIGIN", "-Wno-unused-result"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode(errors="replace"))
        raise RuntimeError("hipcc failed on tests/c/graph_replay.cpp")
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
