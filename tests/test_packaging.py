"""Installable drop-in names (missing #3): the reference installs the console scripts `bayesgm` and `causalBGM`
(src/setup.py:29-33) and its R wrapper imports `bayesgm.models` / `bayesgm.datasets` (r-package/bayesgm/R/python-config.R:86,106).
setup.py declares the same scripts on bayesgm_amd.cli and ships an alias package `bayesgm` (compat/bayesgm) whose names ARE the
bayesgm_amd objects.  Nothing is pip-installed here: the entry points are resolved by hand, the alias is imported with compat/ on the
path, and a wheel is built offline with the installed setuptools and inspected."""
import importlib
import os
import shutil
import subprocess
import sys
import zipfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")


def _setup():
    import runpy
    return runpy.run_path(os.path.join(ROOT, "setup.py"), run_name="setup_constants")


def _py(code, *argv):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([COMPAT, ROOT]))
    return subprocess.run([sys.executable] + (["-c", code] if code else []) + list(argv), cwd="/tmp", env=env, capture_output=True, text=True,
                          timeout=300)


def test_console_scripts_resolve_to_the_cli():
    cfg = _setup()
    scripts = cfg["CONSOLE_SCRIPTS"]
    assert set(scripts) == {"bayesgm", "causalBGM"}                       # setup.py:29-33
    for name, target in scripts.items():
        mod, fn = target.split(":")
        assert callable(getattr(importlib.import_module(mod), fn)), name
    for p in cfg["PACKAGES"]:
        top = p.split(".")[0]
        path = os.path.join(ROOT, cfg["PACKAGE_DIR"][top], *p.split(".")[1:])
        assert os.path.isfile(os.path.join(path, "__init__.py")), p


def test_alias_names_are_the_same_objects():
    r = _py("import sys, bayesgm\n"
            "assert 'torch' not in sys.modules and 'bayesgm_amd.models' not in sys.modules      # lazy, as the reference's __init__\n"
            "import bayesgm.models, bayesgm.datasets, bayesgm.utils, bayesgm_amd.models, bayesgm_amd.datasets, bayesgm_amd.utils\n"
            "assert bayesgm.models.CausalBGM is bayesgm_amd.models.CausalBGM and bayesgm.models.BGM is bayesgm_amd.models.BGM\n"
            "assert bayesgm.models.IdentifiableCausalBGM is bayesgm_amd.models.IdentifiableCausalBGM\n"
            "assert bayesgm.CausalBGM is bayesgm_amd.models.CausalBGM and bayesgm.Sim_Hirano_Imbens_sampler is bayesgm_amd.datasets.Sim_Hirano_Imbens_sampler\n"
            "from bayesgm.models.causalbgm import CausalBGM as C2; from bayesgm.models.bgm import BGM as B2\n"
            "assert C2 is bayesgm_amd.models.CausalBGM and B2 is bayesgm_amd.models.BGM\n"
            "assert bayesgm.datasets.Sim_Hirano_Imbens_sampler is bayesgm_amd.datasets.Sim_Hirano_Imbens_sampler\n"
            "assert bayesgm.utils.get_ADRF is bayesgm_amd.utils.get_ADRF and bayesgm.utils.save_data is bayesgm_amd.utils.save_data\n"
            "from bayesgm.cli.cli import main, main_causalbgm; import bayesgm_amd.cli as c\n"
            "assert main is c.main and main_causalbgm is c.main_causalbgm and bayesgm.__version__ == '1.0.2'\n"
            "try:\n    bayesgm.models.MNISTBGM\n    raise SystemExit('no error')\nexcept AttributeError as e:\n    assert 'outside the hot path' in str(e)\n"
            "print('alias ok')")
    assert r.returncode == 0 and "alias ok" in r.stdout, r.stderr[-2000:]


def test_python_dash_m_bayesgm_is_the_cli():
    r = _py(None, "-m", "bayesgm", "causalbgm", "--help")
    assert r.returncode == 0, r.stderr[-2000:]
    for flag in ("-output_dir", "-input", "-z_dims", "-binary_treatment"):
        assert flag in r.stdout, flag
    r = _py("from bayesgm.cli import main_causalbgm; main_causalbgm(['--help'])")
    assert r.returncode == 0 and "causalBGM" in r.stdout


def test_wheel_carries_both_packages_scripts_and_the_library(tmp_path):
    """an offline wheel build with the installed setuptools (no pip, no network): what `pip install .` would put on a user's path"""
    if not os.path.exists(os.path.join(ROOT, "bayesgm_amd", "libbgm_hip.so")):
        pytest.skip("library not built yet")
    src = tmp_path / "src"
    shutil.copytree(ROOT, src, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "tests", "build", "*.o", "__pycache__", "scripts",
                                                              "oracle", "*.json", "*.md"))
    (src / "README.md").write_text("bayesgm-amd\n")
    out = tmp_path / "dist"
    out.mkdir()
    r = subprocess.run([sys.executable, "-c", "from setuptools import build_meta as b; print(b.build_wheel(%r))" % str(out)], cwd=src,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    whl = [f for f in os.listdir(out) if f.endswith(".whl")]
    assert len(whl) == 1
    with zipfile.ZipFile(out / whl[0]) as z:
        names = set(z.namelist())
        ep = [n for n in names if n.endswith("entry_points.txt")]
        text = z.read(ep[0]).decode()
    assert "bayesgm = bayesgm_amd.cli:main" in text and "causalBGM = bayesgm_amd.cli:main_causalbgm" in text
    for need in ("bayesgm/__init__.py", "bayesgm/models/__init__.py", "bayesgm/datasets/__init__.py", "bayesgm/cli/cli.py", "bayesgm/__main__.py",
                 "bayesgm_amd/libbgm_hip.so", "bayesgm_amd/models/causalbgm.py", "bayesgm_amd/_lib.py", "bayesgm_amd/csrc/build.py",
                 "bayesgm_amd/csrc/causal_api.hip", "bayesgm_amd/csrc/host/host_rng.c"):
        assert need in names, need
