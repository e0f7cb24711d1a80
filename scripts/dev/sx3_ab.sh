python -m pytest tests/test_gpu_bgm_x3.py -q -x 2>&1 | tail -4
for w in 6 8 12; do echo "SX3 W=$w"; BGM_SX3_WAVES=$w BGM_PROBE_PRECISION=f16x3 python scripts/probe_bgm_wide.py 2e5 4 2>&1 | grep -v amdgpu; done
echo "heads only"; BGM_X3_HEADS_ONLY=1 BGM_PROBE_PRECISION=f16x3 python scripts/probe_bgm_wide.py 2e5 4 2>&1 | grep -v amdgpu
