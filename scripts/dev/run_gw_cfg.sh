export BGM_FORCE_GX=1
for cfg in "1 3" "1 4" "2 2" "3 1"; do set -- $cfg; echo "== db $1 occ $2"; BGM_GW_DB=$1 BGM_GW_OCC=$2 python scripts/probe_gx.py 2>&1 | grep "gx  " | grep -v "encoder\|256, 256"; done
