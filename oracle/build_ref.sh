#!/bin/sh
# Builds oracle/_ref/librootba_ref.so: the reference's OWN hot-path sources, compiled where they lie
# under $REF (default /root/reference), against the third-party stand-ins of oracle/ref_shims/.
# TEST INFRASTRUCTURE ONLY (see the header of oracle/ref_driver.cpp for exactly what this pins).
# Nothing is copied out of the reference; outputs go to oracle/_ref/ only (git-ignored, travels to the
# GPU box with the snapshot). Does nothing, successfully, when the reference tree is absent.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${REF:-/root/reference}
OUT="$HERE/_ref"
if [ ! -d "$REF/src/rootba" ]; then
  echo "build_ref.sh: $REF/src/rootba not found - keeping whatever is in $OUT" >&2
  exit 0
fi
mkdir -p "$OUT/obj"
CXX=${CXX:-g++}
# the reference's release flags are -O3 -march=native -DNDEBUG (CMakeLists.txt); assertions are kept ON
# here (the stand-ins check sizes), and the ISA is the portable x86-64-v3 of oracle/Makefile
FLAGS="-std=c++17 -O2 -march=x86-64-v3 -fPIC -DROOTBA_INSTANTIATIONS_FLOAT -DROOTBA_INSTANTIATIONS_DOUBLE \
 -I$HERE/ref_shims -I$REF/src -Wno-deprecated-declarations"
SRCS="bal/bal_bundle_adjustment_helper.cpp bal/residual_info.cpp bal/solver_options.cpp bal/bal_problem.cpp \
 qr/landmark_block.cpp qr/impl/landmark_block_dynamic.cpp \
 solver/linearizor.cpp solver/linearizor_base.cpp solver/linearizor_qr.cpp solver/linearizor_sc.cpp \
 solver/linearizor_power_sc.cpp solver/bal_bundle_adjustment.cpp"
OBJS=""
pids=""
for s in $SRCS; do
  o="$OUT/obj/$(echo "$s" | tr '/' '_' | sed 's/\.cpp$/.o/')"
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$REF/src/rootba/$s" -nt "$o" ] || [ -n "$(find "$HERE/ref_shims" -newer "$o" -type f | head -1)" ]; then
    $CXX $FLAGS -c "$REF/src/rootba/$s" -o "$o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait "$p"; done
$CXX $FLAGS -I"$HERE/../integration" -c "$HERE/ref_driver.cpp" -o "$OUT/obj/ref_driver.o"
$CXX -shared -o "$OUT/librootba_ref.so" $OBJS "$OUT/obj/ref_driver.o"
echo "built $OUT/librootba_ref.so"

# The same reference objects + the reference-side BINDING of the HIP library (integration/): the factory is wrapped
# (integration/linearizor_factory_hip.cpp), the rba_* entry points stay undefined and are resolved at load time
# from whichever provider the test loads first with RTLD_GLOBAL: rootba_amd/librootba_hip.so on a GPU box, or the
# oracle-backed test double librootba_hip_mock.so on a machine without one (tests/test_reference_loop_on_hip.py).
ROOT=$(cd "$HERE/.." && pwd)
W=_ZN6rootba10LinearizorI
X=E6createERNS_10BalProblemI
Y=EERKNS_13SolverOptionsEPNS_13SolverSummaryE
$CXX $FLAGS -I"$ROOT/integration" -I"$ROOT/include" -c "$ROOT/integration/linearizor_factory_hip.cpp" \
  -o "$OUT/obj/linearizor_factory_hip.o" -Wno-return-type-c-linkage
$CXX -shared -o "$OUT/librootba_ref_binding.so" $OBJS "$OUT/obj/ref_driver.o" "$OUT/obj/linearizor_factory_hip.o" \
  -Wl,--wrap=${W}d${X}d${Y} -Wl,--wrap=${W}f${X}f${Y}
$CXX -std=c++17 -O2 -march=x86-64-v3 -fPIC -fopenmp -fvisibility-inlines-hidden -shared -Wl,-Bsymbolic \
  "$HERE/mock_rootba_hip.cpp" -o "$OUT/librootba_hip_mock.so"
echo "built $OUT/librootba_ref_binding.so and $OUT/librootba_hip_mock.so"
