#!/bin/bash
# Post-build check (pytorch-motion-planner_amd/csrc/Makefile): no gfx950 code object of libnfopp_hip.so may contain a packed
# fp32 instruction whose LOW lane is fed from a register pair's HIGH dword (v_pk_{fma,mul,add}_f32 ... op_sel:[..1..]).
# That form returned wrong low lanes on MI355X right behind an LDS read of the pair (tools/micro/
# pk_opsel_lds.hip); hipcc produces it by folding a splat into op_sel -- from its SLP pass (hence -fno-slp-vectorize) and
# from hand-written packed code alike.  op_sel_hi forms (HIGH lane <- low dword) are the harmless direction and pass.
set -e
LIB=$1
OBJDUMP=${OBJDUMP:-/opt/rocm/lib/llvm/bin/llvm-objdump}
TMP=$(mktemp -d)
trap 'rm -rf $TMP' EXIT
cp "$LIB" $TMP/lib.so
[ -x "$OBJDUMP" ] || { echo "check_packed_opsel: $OBJDUMP not found -- the check cannot run, failing closed" >&2; exit 2; }
(cd $TMP && $OBJDUMP --offloading lib.so > $TMP/extract.log 2>&1) || { echo "check_packed_opsel: code-object extraction failed:" >&2; tail -5 $TMP/extract.log >&2; exit 2; }
n=0
objects=0
for co in $TMP/lib.so.*gfx950*; do
  [ -f "$co" ] || continue
  objects=$((objects + 1))
  $OBJDUMP -d "$co" > $TMP/dis.txt || { echo "check_packed_opsel: disassembly of $(basename $co) failed" >&2; exit 2; }
  [ "$(grep -c "v_mfma\|s_endpgm" $TMP/dis.txt)" -gt 0 ] || { echo "check_packed_opsel: $(basename $co) disassembled to no instructions" >&2; exit 2; }
  grep -E "v_pk_(fma|mul|add)_f32" $TMP/dis.txt | grep -E "op_sel:\[[01,]*1[01,]*\]" > $TMP/hits.txt || true
  if [ -s $TMP/hits.txt ]; then
    echo "check_packed_opsel: $(wc -l < $TMP/hits.txt) packed fp32 instruction(s) with a low lane fed from a high dword in $(basename $co):" >&2
    head -5 $TMP/hits.txt >&2
    n=1
  fi
done
# fail closed: a library without a single gfx950 code object was not checked at all
[ "$objects" -gt 0 ] || { echo "check_packed_opsel: no gfx950 code object found in $LIB -- nothing was checked" >&2; exit 2; }
exit $n
