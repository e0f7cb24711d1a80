export BGM_HIP_LIB=$PWD/bayesgm_amd/csrc/build/var/lib_gxnew.so BGM_FORCE_GX=1
for cfg in "2 4" "4 1" "8 1" "6 2" "4 2"; do set -- $cfg; echo "== occ $1 db $2"; BGM_GX_OCC=$1 BGM_GX_DB=$2 python scripts/probe_gx.py 2>&1 | grep "gx  " | grep -v encoder; done
