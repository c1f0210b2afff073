for lib in libvsrbac.so libvsrbac_d8.so; do
 for cfg in "0 0" "256 0" "256 81" "162 81" "324 81" "512 81" "512 32" "768 0"; do set -- $cfg
  echo "$lib budget=$1 fan=$2: $(VSRBAC_LIB=$PWD/vectorsearch-rbac_amd/lib/$lib VSR_BLOCK_BUDGET=$1 VSR_FUSED_FAN=$2 timeout -k 10 200 python3 tools/single_query_probe.py 2>&1 | tail -1 | cut -c1-60)"
 done
done
