#!/bin/bash
# usage: bash tools/isa_waits.sh <file.hip> <kernel name substring> [lines]
# Compiles one source for gfx950 and prints, for the first kernel whose mangled name contains the substring, the order of its
# memory instructions, waits, barriers, branches and MFMA runs: the quickest way to see a load that is waited for with
# vmcnt(0) behind stores, or a prefetch the compiler serialised.
set -e
SRC=$1; PAT=$2; N=${3:-80}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/isa_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I$ROOT/include -I$ROOT/panoswintransformerobjectdetection_amd/csrc -S -o $OUT --cuda-device-only $SRC 2>/dev/null
L=$(grep -n "^_Z[A-Za-z0-9_]*$PAT[A-Za-z0-9_]*:" $OUT | head -1 | cut -d: -f1)
[ -z "$L" ] && { echo "no kernel matching $PAT"; grep -o "^_Z[A-Za-z0-9_]*:" $OUT | head -40; exit 1; }
sed -n "$L,\$p" $OUT | awk '/s_endpgm/{print; exit} {print}' > $OUT.k
sed -n "${L}p" $OUT | cut -c1-120
grep "s_waitcnt vmcnt\|buffer_load\|buffer_store\|global_load\|global_store\|s_cbranch\|s_barrier\|v_mfma\|scratch_" $OUT.k | awk '{print $1, ($1=="s_waitcnt"?$2:"")}' | uniq -c | head -$N
sed -n "$L,\$p" $OUT | grep -m4 "; NumVgprs:\|; ScratchSize:\|; Occupancy:\|; LDSByteSize:"
rm -f $OUT $OUT.k
