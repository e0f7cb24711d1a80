"""Packaging of the MI355X-native hot path so that the reference's callers run unchanged (reference: src/setup.py:3-35).

    python -m bayesgm_amd.csrc.build && pip install --no-build-isolation .

installs the package `bayesgm_amd` (with the prebuilt gfx950 library and its HIP sources as package data), the ALIAS package `bayesgm`
(compat/bayesgm: re-exports of the bayesgm_amd objects, no code) and the reference's two console scripts, `bayesgm` and `causalBGM`.
Metadata lives here rather than in pyproject.toml's [project] table because the image's setuptools (59.6) predates PEP 621."""
import setuptools

PACKAGES = ["bayesgm_amd", "bayesgm_amd.models", "bayesgm_amd.csrc", "bayesgm", "bayesgm.models", "bayesgm.datasets", "bayesgm.utils", "bayesgm.cli"]
PACKAGE_DIR = {"bayesgm_amd": "bayesgm_amd", "bayesgm": "compat/bayesgm"}
CONSOLE_SCRIPTS = {"bayesgm": "bayesgm_amd.cli:main", "causalBGM": "bayesgm_amd.cli:main_causalbgm"}      # setup.py:29-33 of the reference

if __name__ == "__main__":
    setuptools.setup(
        name="bayesgm-amd",
        version="0.5.0",
        description="MI355X-native (HIP / gfx950) implementation of bayesgm's BGM / CausalBGM fit -> posterior sampling -> effects hot path "
                    "behind the reference's Python surface",
        packages=PACKAGES,
        package_dir=PACKAGE_DIR,
        package_data={"bayesgm_amd": ["libbgm_hip.so", "libbgm_hostrng.so"], "bayesgm_amd.csrc": ["*.hip", "*.h", "*.inc", "host/*"]},
        include_package_data=False,
        install_requires=["numpy", "torch", "pyyaml", "scikit-learn", "pandas"],
        python_requires=">=3.9",
        entry_points={"console_scripts": ["%s = %s" % kv for kv in CONSOLE_SCRIPTS.items()]},
        zip_safe=False,
    )
