"""Build libbgm_hip.so for gfx950 with hipcc (in-tree, next to the package).

    python -m bayesgm_amd.csrc.build [--force] [-D NAME=VALUE ...]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the source snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libbgm_hip.so")
SOURCES = ["causal_api.hip", "causal_event_api.hip", "causal_bx3_api.hip", "causal_prior_api.hip", "aux_kernels.hip", "fit_api.hip", "bgm_api.hip", "egm_api.hip", "bgm_egm_api.hip", "bnn_api.hip", "bnn_sample_api.hip", "bprior_api.hip", "bnf_api.hip", "bnx_api.hip", "bnf_det_api.hip", "gx_api.hip", "gx_bgm_api.hip", "bnn_egm_api.hip", "bnn_egm_gen_chain_a.hip", "bnn_egm_gen_chain_b.hip", "bnn_egm_gen_chain_c.hip", "bnn_egm_gen_chain_d.hip", "bgmb_api.hip", "bgmb_egm_api.hip", "bgmf_api.hip", "bnw_api.hip", "comm_api.hip"]
HEADERS = ["bgm_device.h", "causal_kernels.h", "causal_event_kernels.h", "causal_bx3_kernels.h", "fit_kernels.h", "z_replay.h", "prior_kernels.h", "fit_types.h", "fit_sync.h", "bgm_kernels.h", "bgm_fit_kernels.h", "bgm_state.h", "bgm_host.h", "egm_kernels.h", "egm_chain.h", "egm_chain_gen.h", "egm_chain_bnn.h", "bnn_egm_gen_chain.inc", "fit_chain.h", "bgm_egm_kernels.h", "bnn_kernels.h", "bnn_state.h", "bprior_kernels.h", "bprior_types.h", "bnn_sample_kernels.h", "bnf_kernels.h", "bnx_kernels.h", "bnx_host.h", "bnf_host.h", "bnf_build.h", "bnf_det_host.h", "gx_device.h", "gx_causal_kernels.h", "gw_kernels.h", "gx_fit_kernels.h", "gx_host.h", "gx_bgm_kernels.h", "gx_bgm_host.h", "gx_flipout.h", "bgmf_kernels.h", "bgmfx_kernels.h", "bnw_kernels.h", "bnn_egm_kernels.h", "bgmb_kernels.h", "bgmb_state.h", "bgmb_egm_kernels.h", "comm_host.h", os.path.join("..", "..", "include", "bgm_hip.h")]
FLAGS = os.environ.get("BGM_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
         "-Wno-unused-value", "-Wno-unused-result"]


# per-source flags.  bnf_api.hip: hipcc's SLP vectoriser packs the epilogues' scalar fma / mul pairs into v_pk_fma_f32 / v_pk_mul_f32, which
# cannot carry the |x| modifier of the one-instruction LeakyReLU (an extra v_and per element) and issue no faster next to MFMAs.
SOURCE_FLAGS = {"bnf_api.hip": ["-fno-slp-vectorize"], "bnx_api.hip": ["-fno-slp-vectorize"], "bnf_det_api.hip": ["-fno-slp-vectorize"], "causal_bx3_api.hip": ["-fno-slp-vectorize"]}


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, defines=(), verbose=True, out=None):
    OUT = out or globals()["OUT"]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs = []
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(not _newer(d, OUT) for d in deps):
        return OUT
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)

    def fresh(obj):
        """obj is newer than every file its last compilation read (hipcc -MMD dependency file) and was built with the same flags"""
        dep, tag = obj + ".d", obj + ".flags"
        if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(tag)):
            return False
        if open(tag).read() != " ".join(FLAGS + SOURCE_FLAGS.get(os.path.basename(obj).split(".so.")[-1][:-2], []) + list(defines)):
            return False
        words = open(dep).read().replace("\\\n", " ").split()
        t = os.path.getmtime(obj)
        return all(os.path.exists(w) and os.path.getmtime(w) <= t for w in words[1:] if not w.endswith(":"))

    def cc(src):
        obj = os.path.join(os.path.dirname(OUT) if out else os.path.join(HERE, "build"), os.path.basename(OUT) + "." + os.path.basename(src) + ".o")
        if not force and fresh(obj):
            return obj
        extra = SOURCE_FLAGS.get(os.path.basename(src), [])
        cmd = [hipcc] + FLAGS + extra + ["-D" + d for d in defines] + ["-I", HERE, "-MMD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(obj + ".flags", "w").write(" ".join(FLAGS + extra + list(defines)))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def build_probes(verbose=True):
    """Measurement aids (probes/probe_kernels.hip -> probes/libbgm_probe.so): micro-benchmarks used by scripts/probe_*.py only;
    not part of the product library or its header."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    src = os.path.join(HERE, "probes", "probe_kernels.hip")
    out = os.path.join(HERE, "probes", "libbgm_probe.so")
    if os.path.exists(out) and not _newer(src, out) and not _newer(os.path.join(HERE, "bgm_device.h"), out):
        return out
    cmd = [hipcc] + FLAGS + ["-shared", "-I", HERE, src, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--probes" in sys.argv[1:]:
        print(build_probes())
        sys.exit(0)
    defs = []
    argv = sys.argv[1:]
    force = "--force" in argv
    out = None
    for i, a in enumerate(argv):
        if a == "-D":
            defs.append(argv[i + 1])
        elif a.startswith("-D") and len(a) > 2:
            defs.append(a[2:])
        elif a == "-o":
            out = os.path.abspath(argv[i + 1])
    print(build(force=force, defines=defs, out=out))
