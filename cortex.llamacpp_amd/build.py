"""Builds the native library (HIP kernels + C++ host runtime + C-ABI) for gfx950 with hipcc.

    python cortex.llamacpp_amd/build.py            # -> cortex.llamacpp_amd/lib/libmi355_llama.so

hipcc cross-compiles without a GPU.  The .so is kept in-tree (git-ignored) so it travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
# The whole-step kernel (csrc/decode_mega.hip) and the layer engine (csrc/decode_engine.hip) lost their A/Bs against the per-launch path: they are
# experiments, compiled into the library only on request; the product library holds csrc/experiments_absent.cc in their place.
EXPERIMENTS = os.environ.get("MI355_BUILD_EXPERIMENTS", "0") == "1"
SRC = [
    "csrc/mmvq.hip", "csrc/mmvq_fast.hip", "csrc/mmvq_stream.hip", "csrc/mmq.hip", "csrc/mmq_q80.hip", "csrc/act.hip", "csrc/misc.hip", "csrc/mmf.hip", "csrc/attn.hip", "csrc/attn_out.hip", "csrc/attn_prefill.hip", "csrc/clip.hip",
] + (["csrc/decode_engine.hip", "csrc/decode_mega.hip"] if EXPERIMENTS else ["csrc/experiments_absent.cc"]) + [
    "host/gguf.cc", "host/runtime.cc", "host/tp_comm.cc", "host/vocab.cc", "host/sampling.cc", "host/grammar.cc", "host/json_schema.cc", "host/log.cc", "host/server_context.cc", "host/engine.cc",
    "host/hip_backend.cc", "host/tp_split.cc", "host/clip.cc", "host/image_decode.cc", "csrc/c_api.cc",
]
HDRS = ["csrc/dev_common.h", "csrc/kernels.h", "csrc/quant_dev.h", "csrc/mmvq_fast_dev.h", "csrc/mmvq_stream_dev.h", "csrc/attn_decode_dev.h", "host/gguf.h", "host/runtime.h", "host/tp_comm.h", "host/json.h", "host/vocab.h",
        "host/sampling.h", "host/grammar.h", "host/log.h", "host/backend_iface.h", "host/server_context.h", "host/engine.h", "host/hip_backend.h", "host/clip.h", "host/parallel_rows.h", "host/tp_split.h", "host/shm_exchange.h",
        "../include/mi355_llama.h"]
LIB = os.path.join(HERE, "lib", "libmi355_llama.so")
TP_WORKER = os.path.join(HERE, "bin", "mi355_tp_worker")  # a further rank of an engine-formed row split (server/mi355_tp_worker.cc; started by host/tp_split.cc)
SERVER = os.path.join(HERE, "bin", "mi355_server")          # the HTTP host: plain C++, dlopen()s LIB at run time (server/mi355_server.cc)
# -ffp-contract=off: the CPU restatement this backend is checked against does not fuse mul+add
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


# per-file flags.  Kernel-argument preload (gfx950: up to 14 dwords of the FIRST scalar arguments of a kernel arrive in SGPRs with the wave): the stream-bound
# roles of the weight-stream kernel start their stream from them before the argument segment has been read (csrc/mmvq_stream.hip stream_body_fast)
EXTRA = {
    "csrc/mmvq_stream.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
    # (round 6: the fused Q | K | V + attention + attn_output kernel requests the layer input from its first five argument dwords,
    # csrc/attn_out.hip qkv_attn_out_kernel)
    "csrc/attn_out.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=5"],
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hdr_m = max(os.path.getmtime(os.path.join(HERE, h)) for h in HDRS)
    hipcc = _hipcc()
    jobs = []
    for s in SRC:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace("/", "_") + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        rel = os.path.relpath(src, HERE)
        cmd = [hipcc] + FLAGS + EXTRA.get(rel, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(cc, jobs):
            if verbose and warn:
                print(warn)
    objs = [os.path.join(objdir, s.replace("/", "_") + ".o") for s in SRC]
    stamp = os.path.join(objdir, "linked_objects.txt")       # which objects the library on disk was linked from (the experiments switch changes the set)
    want = "\n".join(objs)
    have = open(stamp).read() if os.path.exists(stamp) else ""
    if jobs or not os.path.exists(LIB) or have != want:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-pthread", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
    build_server(force)
    return LIB


def build_server(force: bool = False) -> str:
    src = os.path.join(HERE, "server", "mi355_server.cc")
    deps = [src, os.path.join(HERE, "host", "json.h")]
    if force or not os.path.exists(SERVER) or os.path.getmtime(SERVER) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(SERVER), exist_ok=True)
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-pthread", src, "-o", SERVER, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"server build failed:\n{r.stderr}")
    wsrc = os.path.join(HERE, "server", "mi355_tp_worker.cc")
    if force or not os.path.exists(TP_WORKER) or os.path.getmtime(TP_WORKER) < os.path.getmtime(wsrc):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", wsrc, "-o", TP_WORKER, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"worker build failed:\n{r.stderr}")
    return SERVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
