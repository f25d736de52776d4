"""Build libmdt_hip.so (gfx950) in-tree with hipcc.  `python -m multimodaldiscussiontransformer_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmdt_hip.so")
SOURCES = ["gemm.hip", "gemm_wgrad.hip", "gemm_f8.hip", "layernorm.hip", "attention.hip", "attention_v2.hip", "attention_long.hip", "rowops.hip", "patch_embed.hip", "contrastive.hip", "fp8.hip", "optim.hip", "image.hip", "host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-ffp-contract=off"]
# attention: keep MFMA results in VGPRs — the softmax works on the accumulators in place, and with the default
# AGPR form the compiler spends 10-15 % of the VALU stream on v_accvgpr_read/write copies (gfx950's file is unified)
# gemm_wgrad: the few builtin MFMAs beside the asm ones (the riding bias gradient) must not take AGPRs — all 256 hold the tile
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "attention_v2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         "gemm_wgrad.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def source_hash() -> str:
    """sha256 over every file of csrc/ and include/mdt_hip.h (names and contents, sorted): what the linked library is
    stamped with (mdt_source_hash) and what _lib.py recomputes at import."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp", ".h")))
    files.append(os.path.join(HERE, "..", "include", "mdt_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def hash_inputs_present() -> bool:
    return os.path.isdir(CSRC) and os.path.exists(os.path.join(HERE, "..", "include", "mdt_hip.h"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "attention_common.hpp"), os.path.join(CSRC, "gemm_tiles.hpp"), os.path.join(CSRC, "gemm_epilogue.hpp"),
               os.path.join(HERE, "..", "include", "mdt_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            lang = ["-x", "hip"] if s.endswith(".cpp") else []
            jobs.append([hipcc] + FLAGS + EXTRA.get(s, []) + os.environ.get("MDT_EXTRA_HIPCC_FLAGS", "").split() + lang + ["-c", src, "-o", obj])   # (experiment builds: -D switches)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    # the stamp: a generated translation unit holding the hash of the sources this link was made from
    digest = source_hash()
    stamp_src = os.path.join(objdir, "source_hash.cpp")
    stamp_obj = os.path.join(objdir, "source_hash.o")
    text = f'extern "C" const char* mdt_source_hash(void) {{ return "{digest}"; }}\n'
    if not os.path.exists(stamp_src) or open(stamp_src).read() != text or not os.path.exists(stamp_obj):
        open(stamp_src, "w").write(text)
        run([hipcc, "-x", "c++", "-O1", "-fPIC", "-c", stamp_src, "-o", stamp_obj])     # host-only unit, same compiler as the rest
        jobs.append(None)
    if force or jobs or _stale(OUT, objs + [stamp_obj]):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + [stamp_obj])
    return OUT


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
