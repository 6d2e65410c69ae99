#!/bin/bash
# tools/kres.sh <source.hip> [name regex] [extra hipcc flags]: VGPR / SGPR / LDS / scratch / occupancy of every kernel of one source
# (hipcc -S --cuda-device-only; the resource comments the compiler leaves after each kernel in the ISA listing)
src=$1; flt=${2:-.}; shift; shift
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o $tmp/k.s $src 2>/dev/null || { echo "compile failed"; exit 1; }
awk -v flt="$flt" '
/^_Z[A-Za-z0-9_]*:/ { name=$1; sub(/:$/, "", name) }
/^; NumVgprs:/ { v=$3 } /^; NumAgprs:/ { a=$3 } /^; ScratchSize:/ { sc=$3 } /^; LDSByteSize:/ { lds=$3 } /^; TotalNumSgprs:/ { s=$3 }
/^; Occupancy:/ { if (name ~ flt) { cmd = "echo " name " | c++filt"; cmd | getline dn; close(cmd); gsub(/\(anonymous namespace\)::/, "", dn); sub(/\(.*/, "", dn);
                  printf "%-64s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s occ %s\n", substr(dn,1,64), v, a, s, sc, lds, $3 } }
' $tmp/k.s
rm -rf $tmp
