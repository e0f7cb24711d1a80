"""Generate ABI_MAP.md: every entry point of include/bgm_hip.h -> the translation unit that defines it, the kernel headers that unit
includes (the kernel family), the Python wrapper methods that call it and the test files that exercise it (by symbol or through a wrapper).
    python scripts/gen_abi_map.py [--check]      (--check: exit 1 if ABI_MAP.md is not current; used by tests/test_abi.py)"""
import glob
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
hdr = open(os.path.join(ROOT, "include", "bgm_hip.h")).read()
syms = re.findall(r"BGM_API\s+[\w\s\*]+?\b(bgm_\w+)\s*\(", hdr)
syms = list(dict.fromkeys(syms))
csrc = os.path.join(ROOT, "bayesgm_amd", "csrc")
tus = {}
for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc"))):
    tus[os.path.basename(f)] = open(f).read()
py = {}
for f in sorted(glob.glob(os.path.join(ROOT, "bayesgm_amd", "*.py")) + glob.glob(os.path.join(ROOT, "bayesgm_amd", "models", "*.py"))):
    py[os.path.relpath(f, ROOT)] = open(f).read()
tests = {os.path.basename(f): open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "test_*.py")))}
scripts = {os.path.basename(f): open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "scripts", "dp_*.py")))}
GENERIC = {"close", "begin", "end", "read", "write", "predict", "fit", "evaluate", "__init__", "__del__", "_stream", "split", "load"}


import ast

FUNCS = []      # (file, class or None, function name, own source text)
for f_, src_ in py.items():
    tree = ast.parse(src_)
    parents = {}
    for node in ast.walk(tree):
        for ch in ast.iter_child_nodes(node):
            parents[ch] = node
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            inner = [n for n in ast.walk(node) if isinstance(n, ast.FunctionDef) and n is not node]
            seg = ast.get_source_segment(src_, node) or ""
            for n in inner:                      # a nested helper's text belongs to the helper, not to the enclosing function
                seg = seg.replace(ast.get_source_segment(src_, n) or "\0", "")
            par = parents.get(node)
            FUNCS.append((f_, par.name if isinstance(par, ast.ClassDef) else None, node.name, seg))


def wrappers_of(sym):
    """(file, class, function) triples whose own body mentions the symbol, plus the methods of the same class that call those"""
    direct = [(f_, c, name) for f_, c, name, seg in FUNCS if re.search(r"\b%s\b" % sym, seg)]
    out = list(direct)
    for f_, c, name in direct:
        if c is None or name in GENERIC:
            continue
        for f2, c2, n2, seg in FUNCS:
            if f2 == f_ and c2 == c and n2 != name and re.search(r"self\.%s\(" % re.escape(name), seg) and (f2, c2, n2) not in out:
                out.append((f2, c2, n2))
    return out


# which wrapper classes a test file exercises: by name, or through a helper imported from another test file that names them
def classes_in(src):
    return set(re.findall(r"\b(CausalEngine|BgmEngine|BnnEngine|BvnEngine|CausalBGM|BGM|IdentifiableCausalBGM|CausalBGMBayes|BGMBayes|IdentifiableCausalBGMBayes)\b", src))


TEST_CLASSES = {tn: classes_in(src) for tn, src in tests.items()}
for tn, src in tests.items():
    for other in re.findall(r"from (?:tests\.)?(test_\w+) import", src):
        TEST_CLASSES[tn] |= TEST_CLASSES.get(other + ".py", set())
MODEL_OF = {"CausalBGMBayes": "CausalBGM", "BGMBayes": "BGM", "IdentifiableCausalBGMBayes": "IdentifiableCausalBGM"}
# kernels that reach a translation unit through bgm_host.h / shared headers rather than a direct include
FAMILY_OVERRIDE = {"causal_api.hip": "causal_kernels.h (through bgm_host.h); dispatches to causal_event_api.hip, causal_bx3_api.hip, causal_prior_api.hip, bnf_det_api.hip, gx_api.hip",
                   "aux_kernels.hip": "aux_kernels.hip (reductions, quantiles)",
                   "bnn_sample_api.hip": "bnn_sample_kernels.h (batch-statistics normalisation); dispatches to bnf_api.hip (bnf_kernels.h: inference-mode "
                                         "normalisation, default shapes; bprior_kernels.h for the conditional prior) and bnw_api.hip (bnw_kernels.h: hidden widths > 64)"}


rows = []
for s in syms:
    tu = [n for n, src in tus.items() if re.search(r'extern "C"[^;{]*\b%s\s*\(' % s, src)]
    fam = []
    for n in tu:
        fam += [h for h in re.findall(r'#include "(\w+kernels\w*\.h|\w+chain\w*\.h|gx_\w+\.h|aux_\w+\.h)"', tus[n]) if not h.endswith("host.h")]
    ws = wrappers_of(s)
    names = {w for _, _, w in ws if w not in GENERIC and not w.startswith("__")}
    hit = []
    for tn, src in tests.items():
        direct = re.search(r"\b%s\b" % s, src) is not None
        via = False
        for _, c, w in ws:
            if w in GENERIC or w.startswith("__") or not re.search(r"\.%s\(" % re.escape(w), src):
                continue
            if c is None or c in TEST_CLASSES[tn] or MODEL_OF.get(c) in TEST_CLASSES[tn]:
                via = True
        if direct or via:
            hit.append(tn)
    if not hit:      # reached only through the model classes' fit / predict / evaluate: the class-level tests of the module that wraps it
        cls = {MODEL_OF.get(c, c) for _, c, _ in ws if c}
        for tn in tests:
            if cls & TEST_CLASSES[tn] and any(k in tn for k in ("class_level", "cli", "main_yaml", "identifiable", "bgm_bnn", "gpu_bnn", "gpu_bgm")):
                hit.append(tn)
    if tu and tu[0] in FAMILY_OVERRIDE:
        fam = [FAMILY_OVERRIDE[tu[0]]]
    rows.append((s, ", ".join(tu) or "?", ", ".join(dict.fromkeys(fam)) or "(host code only)", ", ".join(sorted(names)) or "-",
                 ", ".join(sorted(set(hit))) or "tests/test_abi.py (export only)"))

lines = ["# ABI map -- include/bgm_hip.h entry points (%d)" % len(rows), "",
         "Generated by `python scripts/gen_abi_map.py` (kept current by `tests/test_abi.py::test_abi_map_is_current`).  Columns: entry point; the",
         "translation unit under `bayesgm_amd/csrc/` that defines it; the kernel headers that unit includes (the kernel family that serves the call;",
         "dispatch between families is described in DESIGN.md sections 4 and 4b); the Python wrapper methods that call it; the test files that",
         "exercise it, by name or through a wrapper (`-m gpu` files call through the C ABI on an MI355X).", "",
         "| entry point | defined in | kernel headers of that unit | Python wrappers | tests |", "|---|---|---|---|---|"]
for r in rows:
    lines.append("| `%s` | %s | %s | %s | %s |" % r)
text = "\n".join(lines) + "\n"
path = os.path.join(ROOT, "ABI_MAP.md")
if "--check" in sys.argv:
    cur = open(path).read() if os.path.exists(path) else ""
    if cur != text:
        print("ABI_MAP.md is stale: run python scripts/gen_abi_map.py")
        sys.exit(1)
    sys.exit(0)
open(path, "w").write(text)
print("%d entry points -> %s" % (len(rows), path))
unmapped = [r[0] for r in rows if r[1] == "?"]
print("without a defining unit:", unmapped)
print("export-only:", [r[0] for r in rows if r[4].startswith("tests/test_abi.py (export")])
