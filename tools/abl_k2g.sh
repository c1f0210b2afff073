#!/bin/bash
# Development tool: times the 768-d GEMM launch (tools/kbench.py, 2M x 768, 1000 unfiltered queries) on the product library and
# on the variant libraries (libvsrbac_<name>.so) found in vectorsearch-rbac_amd/lib.
out=$1; mkdir -p "$out"
for lib in vectorsearch-rbac_amd/lib/libvsrbac.so vectorsearch-rbac_amd/lib/libvsrbac_*.so; do
  name=$(basename $lib .so)
  VSRBAC_LIB=$PWD/$lib timeout -k 10 240 python3 tools/kbench.py --rows 2000000 --dim 768 --gauss --metric cosine --cases full1000 --steps 5 > "$out/$name.json" 2> "$out/$name.err" || echo "$name failed"
  echo "$name $(tail -1 $out/$name.json | cut -c1-400)"
done
