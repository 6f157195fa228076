#!/bin/bash
# lists every kernel of the product build whose code object reports spilled registers / scratch (hipcc does not warn)
#   bash tools/spill_scan.sh [obj dir]
OBJ=${1:-dash-infer_amd/lib/obj}
TMP=$(mktemp -d)
for o in "$OBJ"/*.o; do
  b=$(basename "$o")
  cp "$o" "$TMP/$b" && (cd "$TMP" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading "$b" >/dev/null 2>&1)
  f=$(ls "$TMP" | grep "^$b\..*gfx950" | head -1)
  [ -z "$f" ] && continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/$f" | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|sgpr_spill" | paste - - - - - |
    awk -v B="$b" '{n=$2; ps=$4; ss=$6; vc=$8; vs=$10; if (ps+0 > 0 || vs+0 > 0) printf "%s  %s  scratch %s B, sgpr spills %s, vgprs %s, vgpr spills %s\n", B, n, ps, ss, vc, vs}'
done
rm -rf "$TMP"
